// snac_traj.hip -- trajectory memory (snac_traj_alloc / snac_traj_free / snac_traj_describe of include/snac_hip.h).
//
// On MI355X the physical address space behaves as slices of 32 GiB: write streams that stay inside one slice top out at ~5.7 TB/s,
// the same streams spread over two or more slices reach ~7.1 (tools/wr_blocks.hip, profiles/; the split may be as coarse as 128 MB
// pieces taking turns).  A tensor from hipMalloc is one contiguous run of at most 16 GB -- inside one slice unless it happens to
// straddle a boundary, which is all the "fast and slow regions" of the address map ever were.  The virtual-memory API lets ONE
// contiguous virtual range be backed by 32 MB handles from different slices taking turns; nothing about the tensor changes for its
// users.  Which slice a handle lies in cannot be asked, so it is MEASURED, and since round 4 the measurement is of the thing itself:
//   * the probe kernel writes the headline rollout's own store shape (emit_tile in snac_hip.hip: a wave's 64-env tile = 26 112
//     contiguous bytes per step in two halves, 16 bytes per lane, 1 KiB per store instruction; 1024 waves = one step's 26.7 MB);
//   * a block is assembled from WINDOWS of 1 GiB -- the 16 chunks of a group from the reference's slice and the 16 chunks of a
//     group from another slice taking turns -- and every window is timed in exactly that arrangement BEFORE it is used: a pair that
//     does not run at the fast level is taken apart (whichever of the two also fails with another partner is dropped);
//   * the finished block is timed again under the same pattern, window by window over ALL of it and once as a whole; a block with a
//     slow window is rebuilt from a fresh, larger pool.  Round 3 timed the first GiB only, chunks assigned clearest cases first:
//     in the driver's run three of six such blocks ran at the single-slice level and were handed out as "measured".
// What was measured travels with the block (snac_traj_describe): levels, every window's time, rebuilds, probe launches.
// The fallback when a measurement is not to be had: handles created back to back -- run 0, a gap that brings the distance to
// 32 GiB, run 1, a gap, run 2 --, virtual chunk j mapped to run j % 3, the gaps released; consecutive handles follow each other in
// physical memory only on an allocator that has seen no releases, so that layout is a lottery (5.7-7.1 TB/s, tools/wr_vmm.hip).
// This is the one place where the library allocates and keeps state of its own: g_traj maps every live block to the handles that
// back it (what snac_traj_free needs to unmap it) and to its description, behind g_traj_mu.  Every block is checked before it is
// handed out: a pattern written by one kernel, read back by another and -- one word per chunk -- by a copy (traj_verify).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "snac_common.h"

using snac_detail::fail;
using snac_detail::fail_hip;
using snac_detail::g_err;

namespace {
struct TrajBlock {
    size_t total, chunk;
    int device, layout;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    snac_traj_info info;
};
std::mutex g_traj_mu;
std::unordered_map<void*, TrajBlock> g_traj;
std::atomic<unsigned long long> g_reserved_dead{0};     // address space of ranges that are no longer mapped (never handed out again)
// tests/native/traj_host_test.cpp compiles this file with gcc against a fake HIP layer and shrinks the whole geometry by 2^13 (4 KB
// chunks, "1 GiB" = 128 KB) so that every path -- pools, spacers, windows, rebuilds, every failure -- runs on real memory in milliseconds
#ifndef SNAC_TRAJ_TEST_SHIFT
#define SNAC_TRAJ_TEST_SHIFT 0
#endif
#ifndef SNAC_TRAJ_AUX_BLOCKS
#define SNAC_TRAJ_AUX_BLOCKS 4096
#endif
constexpr size_t gib(size_t n) { return (n << 30) >> SNAC_TRAJ_TEST_SHIFT; }
constexpr int TRAJ_CHUNK_LOG2 = 25 - SNAC_TRAJ_TEST_SHIFT;
constexpr size_t TRAJ_CHUNK = (size_t)1 << TRAJ_CHUNK_LOG2;   // one physical handle per 32 MB: 480 handles for the headline's 16 GB
constexpr size_t TRAJ_SLICE = gib(32);   // distance between the starts of consecutive runs
constexpr int TRAJ_RUNS = 3;
constexpr size_t TRAJ_SPLIT_MIN = gib(1);     // smaller blocks are not worth the probe: one run
constexpr size_t TRAJ_POOL_DEFAULT = gib(160);  // what the pool (groups + spacers) may hold beyond the block itself
constexpr size_t TRAJ_MARGIN = gib(4);        // device memory the pool never touches

// Unmap chunk by chunk (each call undoes exactly one hipMemMap), release the physical handles -- and KEEP the address range
// reserved: a range that was handed out again right after an unmap has been seen to serve stale translations (round 2: a fresh
// block at a recycled address read back zeros through a copy after a kernel had filled it; tools/vmm_stale.hip reproduces it,
// profiles/r03_vmm_stale.txt).  A reservation costs no memory, and a stale pointer into a freed block faults instead of hitting
// someone else's data.  The price is address space: every block of 1 GiB or more leaves its own range and the ranges its pool was
// probed in behind (counted in g_reserved_dead, snac_traj_reserved_bytes(); 47 bits of address space last for > 1000 headline-sized
// blocks per process).
void traj_release(char* va, size_t mapped, size_t chunk, std::vector<hipMemGenericAllocationHandle_t>& hs) {
    for (size_t off = 0; off < mapped; off += chunk) (void)hipMemUnmap(va + off, chunk);
    for (auto h : hs) (void)hipMemRelease(h);
    (void)hipGetLastError();
    g_reserved_dead += mapped;
}

// ---- the check every block passes before it is handed out --------------------------------------------------------------------
__device__ __forceinline__ uint64_t traj_word(uint64_t i, uint64_t salt) {
    uint64_t x = (i + salt) * 0x9E3779B97F4A7C15ull;
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    return x;
}
__global__ __launch_bounds__(256) void k_traj_fill(uint64_t* p, size_t words, uint64_t salt) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) p[i] = traj_word(i, salt);
}
__global__ __launch_bounds__(256) void k_traj_check(const uint64_t* p, size_t words, uint64_t salt, unsigned long long* bad) {
    unsigned long long n = 0;
    // the other way round: the last word first, so that no lane meets the lines its own fill left in a cache
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x)
        n += p[words - 1 - i] != traj_word(words - 1 - i, salt);
    if (n) atomicAdd(bad, n);
}
// 0: every word of the block reads back what was written, through a kernel and (first word of every chunk) through a copy
int traj_verify(char* va, size_t total, size_t chunk, hipStream_t stream) {
    const size_t words = total / 8, nchunks = total / chunk;
    const uint64_t salt = (uint64_t)(uintptr_t)va ^ 0x5AC5AC5ull;
    unsigned long long* bad = nullptr;
    hipError_t e = hipMalloc((void**)&bad, sizeof(*bad));
    if (e != hipSuccess) return fail_hip(e, "hipMalloc (block check)");
    std::vector<uint64_t> firsts(nchunks, 0);
    unsigned long long hbad = 0;
    (void)hipMemsetAsync(bad, 0, sizeof(*bad), stream);
    hipLaunchKernelGGL(k_traj_fill, dim3(SNAC_TRAJ_AUX_BLOCKS), dim3(256), 0, stream, (uint64_t*)va, words, salt);
    hipLaunchKernelGGL(k_traj_check, dim3(SNAC_TRAJ_AUX_BLOCKS), dim3(256), 0, stream, (const uint64_t*)va, words, salt, bad);
    e = hipMemcpyAsync(&hbad, bad, sizeof(hbad), hipMemcpyDeviceToHost, stream);
    for (size_t c = 0; c < nchunks && e == hipSuccess; ++c) e = hipMemcpyAsync(&firsts[c], va + c * chunk, 8, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(bad);
    if (e != hipSuccess) return fail_hip(e, "block check");
    size_t cbad = 0;
    for (size_t c = 0; c < nchunks; ++c) {
        uint64_t x = (c * (chunk / 8) + salt) * 0x9E3779B97F4A7C15ull;
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        cbad += firsts[c] != x;
    }
    if (hbad || cbad) {
        std::snprintf(g_err, sizeof(g_err), "trajectory block at %p failed its check: %llu words differ through a kernel, %zu of %zu chunks through a copy",
                      (void*)va, hbad, cbad, nchunks);
        return SNAC_ERR_HIP;
    }
    return SNAC_OK;
}

// ---- the probed layout -------------------------------------------------------------------------------------------------------
// Where a handle lands is the driver's business, so the block is built from what a measurement says: 32 MB handles are created
// back to back and taken in groups of 16 (512 MB, each mapped into a range of its own).  Two groups are timed TOGETHER -- the
// rollout's store shape over the two groups' chunks taking turns, 1 GiB per probe, ~0.15 ms.  Pairs in different slices run at
// ~7 TB/s, pairs in the same slice at ~5.7.  Step 1 sorts the pool against one reference group into "near" (its slice) and "far"
// (another); step 2 assembles the block's windows from (near, far) pairs, each timed as the pair it will be; step 3 times the
// mapped block.  The pool starts at the block's own size + 8 GiB and grows by 8 GiB while one of the two kinds is short, up to
// `pool_cap` beyond the block -- never more than half of what is free and never into the last 4 GiB.  No contrast (a pool inside
// one slice, a driver that scatters handles below the group size) or no memory for a pool: the caller falls back to the fixed
// three-run layout.
constexpr size_t TRAJ_GROUP = 16;                                // chunks per probed group
constexpr size_t TRAJ_WINDOW = 2 * TRAJ_GROUP;                   // chunks per window of the block: a near group and a far group in turn
constexpr size_t TRAJ_GROW = 16;                                 // groups per pool extension (8 GiB)
constexpr int PROBE_WAVES = 1024;                                // N = 65 536 envs in tiles of 64
constexpr int PROBE_HALF = 32 * 51 * 8;                          // emit_tile<double>: 13 056 bytes per half tile
constexpr size_t PROBE_TILE = 2 * (size_t)PROBE_HALF;            // 26 112 bytes per wave and step

// logical chunk c of the probed bytes lies in a (c even) or b (c odd) at chunk index c >> 1; or, for a finished block
// (b == a + chunk, pair_log2 = chunk_log2 + 1), simply at a + c * chunk.  reps: passes over the same bytes in one launch (warm-up).
__global__ __launch_bounds__(256) void k_traj_probe(char* a, char* b, int chunk_log2, int pair_log2, int chunks_total, int reps) {
    const int lane = threadIdx.x & 63, wave = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6), waves = (int)gridDim.x * 4;
    const size_t total = (size_t)chunks_total << chunk_log2, mask = ((size_t)1 << chunk_log2) - 1;
    const int steps = (int)(total / ((size_t)waves * PROBE_TILE));
    for (int r = 0; r < reps; ++r)
        for (int t = 0; t < steps; ++t) {
            const size_t L0 = ((size_t)t * waves + wave) * PROBE_TILE + (size_t)lane * 16;
            const uint4 v = make_uint4((unsigned)t, (unsigned)wave, (unsigned)lane, (unsigned)r);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 13; ++i) {
                    if (i * 1024 + lane * 16 < PROBE_HALF) {         // the 13th store of a half covers 768 of its 1024 bytes
                        const size_t L = L0 + (size_t)(h * PROBE_HALF + i * 1024), c = L >> chunk_log2;
                        *(uint4*)(((c & 1) ? b : a) + ((c >> 1) << pair_log2) + (L & mask)) = v;
                    }
                }
        }
}

struct TrajPool {
    size_t chunk, gbytes;
    hipMemAllocationProp prop;
    hipMemAccessDesc acc;
    std::vector<hipMemGenericAllocationHandle_t> h;              // handles in creation order: group g = h[g * TRAJ_GROUP ...]
    std::vector<char*> gva;                                      // where group g is mapped
    std::vector<std::pair<char*, size_t>> ranges;                // the reservations the groups live in (kept reserved, see traj_release)
    std::vector<hipMemGenericAllocationHandle_t> spacers;        // unmapped handles that only hold physical memory (see extend())
    size_t spacer_bytes = 0;
    size_t groups() const { return gva.size(); }
    bool space(size_t bytes) {
        hipMemGenericAllocationHandle_t x;
        if (hipMemCreate(&x, bytes, &prop, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
        spacers.push_back(x); spacer_bytes += bytes;
        return true;
    }
    // `n` more groups: handles created back to back, mapped into one new range; returns how many groups were added
    size_t grow(size_t n, size_t gran) {
        if (!n) return 0;
        std::vector<hipMemGenericAllocationHandle_t> fresh;
        for (size_t i = 0; i < n * TRAJ_GROUP; ++i) {
            hipMemGenericAllocationHandle_t x;
            if (hipMemCreate(&x, chunk, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
            fresh.push_back(x);
        }
        size_t got = fresh.size() / TRAJ_GROUP;
        while (fresh.size() > got * TRAJ_GROUP) { (void)hipMemRelease(fresh.back()); fresh.pop_back(); }
        if (!got) return 0;
        char* va = nullptr;
        if (hipMemAddressReserve((void**)&va, got * gbytes, gran, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); for (auto x : fresh) (void)hipMemRelease(x); return 0; }
        size_t mapped = 0;
        bool ok = true;
        for (size_t i = 0; i < fresh.size() && ok; ++i) {
            if (hipMemMap(va + i * chunk, chunk, 0, fresh[i], 0) != hipSuccess) { (void)hipGetLastError(); ok = false; } else mapped += chunk;
        }
        if (ok && hipMemSetAccess(va, mapped, &acc, 1) != hipSuccess) { (void)hipGetLastError(); ok = false; }
        if (!ok) { traj_release(va, mapped, chunk, fresh); g_reserved_dead += got * gbytes - mapped; return 0; }   // (the whole range stays reserved: round 6, tests/native/traj_host_test.cpp)
        ranges.emplace_back(va, mapped);
        for (size_t g = 0; g < got; ++g) gva.push_back(va + g * gbytes);
        h.insert(h.end(), fresh.begin(), fresh.end());
        return got;
    }
    // unmap everything; release the handles not marked in `keep` (keep == nullptr: all of them)
    void drop(const std::vector<char>* keep) {
        for (auto& r : ranges) {
            for (size_t off = 0; off < r.second; off += chunk) (void)hipMemUnmap(r.first + off, chunk);
            g_reserved_dead += r.second;
        }
        for (size_t i = 0; i < h.size(); ++i) if (!keep || !(*keep)[i]) (void)hipMemRelease(h[i]);
        for (auto x : spacers) (void)hipMemRelease(x);
        (void)hipGetLastError();
        ranges.clear(); gva.clear(); spacers.clear(); spacer_bytes = 0;
    }
};

float median_of(std::vector<float> v) {
    if (v.empty()) return 0.f;
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

// 0: *out holds the block (described by *info; info->windows_slow > 0: a window of the finished block did not run at the fast
// level); 1: not applicable (no room for a pool; nothing allocated: use the fixed layout); 2: measured, but this pool does not hold
// two classes in sufficient number (nothing allocated; a second pool may: releasing one reshuffles the driver's free lists);
// < 0: error code.
// extra_groups: pool groups beyond the usual start (a rebuild starts larger)
int traj_alloc_probed(size_t bytes, int device, const hipMemAllocationProp& prop, size_t gran, size_t pool_cap, size_t extra_groups,
                      hipStream_t stream, void** out, snac_traj_info* info) {
    const size_t chunk = TRAJ_CHUNK;
    if (chunk % gran) return 1;
    const size_t k = (bytes + chunk - 1) / chunk, kg = (k + TRAJ_GROUP - 1) / TRAJ_GROUP;
    const size_t W = (k + TRAJ_WINDOW - 1) / TRAJ_WINDOW;        // windows of the block: W near groups and W far groups
    const bool debug = std::getenv("SNAC_TRAJ_DEBUG") != nullptr;
    // what the pool may take beyond the block: the caller's cap, half of what would be free next to the block, nothing of the last 4 GiB
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return 1; }
    if (free_b < bytes + TRAJ_MARGIN) return 1;
    size_t extra = std::min(pool_cap, (free_b - bytes) / 2);
    extra = std::min(extra, free_b - bytes - TRAJ_MARGIN);
    const size_t max_groups = kg + extra / (TRAJ_GROUP * chunk);
    if (max_groups < kg + 8) return 1;                           // not enough memory for a pool worth probing
    const size_t limit_bytes = max_groups * TRAJ_GROUP * chunk;   // groups and spacers together
    TrajPool pool;
    pool.chunk = chunk; pool.gbytes = TRAJ_GROUP * chunk; pool.prop = prop;
    std::memset(&pool.acc, 0, sizeof(pool.acc));
    pool.acc.location = prop.location; pool.acc.flags = hipMemAccessFlagsProtReadWrite;
    auto give_up = [&]() { pool.drop(nullptr); return 1; };
    auto give_up_measured = [&]() { pool.drop(nullptr); return 2; };   // probed, and no block to be had from this pool
    pool.grow(std::min(max_groups, kg + TRAJ_GROW + extra_groups), gran);
    if (pool.groups() < kg + 8) return give_up();
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool ok = hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
    int launches = 0;
    // one launch of the store pattern over `chunks` logical chunks of (pa, pb), timed on the stream: ms
    auto launch_probe = [&](char* pa, char* pb, int pair_log2, int chunks, int reps, bool timed) -> float {
        if (timed) (void)hipEventRecord(e0, stream);
        hipLaunchKernelGGL(k_traj_probe, dim3(PROBE_WAVES / 4), dim3(256), 0, stream, pa, pb, TRAJ_CHUNK_LOG2, pair_log2, chunks, reps);
        ++launches;
        if (!timed) return 0.f;
        (void)hipEventRecord(e1, stream);
        if (hipEventSynchronize(e1) != hipSuccess) { ok = false; return 0.f; }
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) ok = false;
        return ms;
    };
    // a pair of groups as a window of the block would hold them: the faster of two launches, in us
    auto probe = [&](size_t ga, size_t gb) -> float {
        const float t1 = launch_probe(pool.gva[ga], pool.gva[gb], TRAJ_CHUNK_LOG2, (int)TRAJ_WINDOW, 1, true);
        const float t2 = launch_probe(pool.gva[ga], pool.gva[gb], TRAJ_CHUNK_LOG2, (int)TRAJ_WINDOW, 1, true);
        return 1000.f * (t1 < t2 ? t1 : t2);
    };
    // the same for sorting the pool against a reference: ONE launch where the class is clear (noise only ever adds time, and every pair
    // the block is made of is timed again as a pair), a second one between the levels -- half of the probe launches of a block
    auto probe_class = [&](size_t ga, size_t gb, float self_us) -> float {
        const float t1 = 1000.f * launch_probe(pool.gva[ga], pool.gva[gb], TRAJ_CHUNK_LOG2, (int)TRAJ_WINDOW, 1, true);
        if (t1 <= 1.05f * self_us || t1 >= 1.18f * self_us) return t1;
        const float t2 = 1000.f * launch_probe(pool.gva[ga], pool.gva[gb], TRAJ_CHUNK_LOG2, (int)TRAJ_WINDOW, 1, true);
        return t1 < t2 ? t1 : t2;
    };
    std::vector<size_t> order_a, order_b;                        // groups of class A (the reference's slice) / class B, surest first
    float fast_level = 0.f, slow_level = 0.f, t_self = 0.f;
    bool found = false;
    const auto tp0 = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp0).count(); };
    if (ok) {
        launch_probe(pool.gva[0], pool.gva[1], TRAJ_CHUNK_LOG2, (int)TRAJ_WINDOW, 80, false);   // ~12 ms of the probe itself: the clocks are up before anything is timed
        if (hipGetLastError() != hipSuccess) ok = false;
        std::vector<float> t;                                    // us of group g against the current reference (0: not measured)
        std::vector<char> aside;                                 // groups of earlier references' slices: class B material
        size_t ref = 0, self_of = (size_t)-1;
        int blur_tries = 0, switches = 0, stagnant = 0;
        size_t jump = 0;                                         // bytes of spacer in front of the next extension
        // More memory for the pool.  Handles come from wherever the driver's allocator is at, and on a box whose memory has seen
        // other tenants 80 GiB in a row can lie in ONE slice (profiles/r04_traj_alloc.txt): when an extension brought nothing of the
        // class that is short, the next one first takes a SPACER -- one unmapped handle of 16, 32, 64 GiB, released with the pool --
        // so that the 8 GiB behind it come from somewhere else.
        auto extend = [&]() {
            const size_t held = pool.groups() * pool.gbytes + pool.spacer_bytes;
            if (held + 4 * pool.gbytes > limit_bytes) return false;
            size_t room = limit_bytes - held;
            if (jump) {
                size_t sp = std::min(jump, room > TRAJ_GROW * pool.gbytes ? room - TRAJ_GROW * pool.gbytes : (size_t)0);
                sp &= ~(chunk - 1);
                while (sp >= gib(2) && !pool.space(sp)) sp = (sp / 2) & ~(chunk - 1);
                room = limit_bytes - (pool.groups() * pool.gbytes + pool.spacer_bytes);
            }
            const size_t got = pool.grow(std::min(TRAJ_GROW, room / pool.gbytes), gran);
            if (debug) std::fprintf(stderr, "snac_traj_alloc: [%.0f ms] pool + %zu groups behind %.0f GiB of spacers: %zu groups\n", since(), got, (double)pool.spacer_bytes / 1073741824.0, pool.groups());
            return got > 0;
        };
        for (int round = 0; round < 48 && ok && !found; ++round) {
            t.resize(pool.groups(), 0.f); aside.resize(pool.groups(), 0);
            if (self_of != ref) { t_self = probe(ref, ref); self_of = ref; }   // the reference paired with ITSELF: the fast level, measured
            float lo = 1e30f, hi = 0.f;
            std::vector<size_t> part;                            // the partners measured against this reference
            for (size_t g = 0; g < pool.groups() && ok; ++g) {
                if (g == ref || aside[g]) continue;
                if (t[g] == 0.f) t[g] = probe_class(ref, g, t_self);
                part.push_back(g);
                lo = t[g] < lo ? t[g] : lo; hi = t[g] > hi ? t[g] : hi;
            }
            if (debug) {
                std::fprintf(stderr, "snac_traj_alloc: [%.0f ms] probe round %d, reference group %zu (with itself %.0f us), %zu groups, %.0f .. %.0f us:", since(), round, ref, t_self, pool.groups(), lo, hi);
                for (size_t g = 0; g < pool.groups(); ++g) std::fprintf(stderr, " %.0f", t[g]);
                std::fprintf(stderr, "\n");
            }
            if (!ok || part.empty()) break;
            // The scale comes from the reference group paired with ITSELF, which runs at the fast level (the same rows written twice),
            // as a partner in another slice does (1.02 .. 1.05 x); partners in the reference's own slice take 1.18 .. 1.35 x.  A
            // reference that straddles two stretches of physical memory blurs every time measured against it -- most partners land
            // BETWEEN the two levels -- and nothing can be judged from it: its median partner is a clean group almost surely (a
            // straddler's slowest and fastest partners are straddlers themselves).
            size_t n_mid = 0;
            for (size_t g : part) n_mid += (t[g] > 1.07f * t_self && t[g] < 1.16f * t_self) ? 1 : 0;
            if (blur_tries < 3 && part.size() >= 8 && n_mid * 20 > part.size() * 7) {
                ++blur_tries;
                std::vector<size_t> by(part);
                std::sort(by.begin(), by.end(), [&](size_t x, size_t y) { return t[x] < t[y]; });
                if (debug) std::fprintf(stderr, "snac_traj_alloc: reference group %zu is blurred (%zu of %zu partners between the levels): taking its median partner %zu\n", ref, n_mid, part.size(), by[by.size() / 2]);
                ref = by[by.size() / 2];
                std::fill(t.begin(), t.end(), 0.f);
                continue;
            }
            std::vector<size_t> near{ref}, far;                  // near: the reference's slice; far: another one; the clear cases only count
            size_t near_clean = 1, far_clean = 0;
            for (size_t g : part) {
                (t[g] >= 1.10f * t_self ? near : far).push_back(g);
                near_clean += t[g] >= 1.16f * t_self ? 1 : 0;
                far_clean += t[g] <= 1.07f * t_self ? 1 : 0;
            }
            size_t n_aside = 0;
            for (size_t g = 0; g < pool.groups(); ++g) n_aside += aside[g] ? 1 : 0;
            if (near_clean >= W && far_clean + n_aside >= W) {
                // the clearest cases first: the slowest partners are surest to share the reference's slice, the fastest surest not to
                // (in-between times are groups that straddle two regions)
                std::sort(near.begin() + 1, near.end(), [&](size_t x, size_t y) { return t[x] > t[y]; });
                std::sort(far.begin(), far.end(), [&](size_t x, size_t y) { return t[x] < t[y]; });
                std::vector<float> tf, tn;
                for (size_t g : far) if (t[g] <= 1.07f * t_self) tf.push_back(t[g]);
                for (size_t g : near) if (g != ref && t[g] >= 1.16f * t_self) tn.push_back(t[g]);
                fast_level = tf.empty() ? 1.04f * t_self : std::min(median_of(tf), 1.05f * t_self);
                slow_level = median_of(tn);
                // class B: the clear far partners, then the groups of earlier references' slices (not this one's), then the doubtful
                size_t nclear = 0;
                while (nclear < far.size() && t[far[nclear]] <= 1.07f * t_self) ++nclear;
                for (size_t g : near) order_a.push_back(g);
                for (size_t i = 0; i < nclear; ++i) order_b.push_back(far[i]);
                for (size_t g = 0; g < pool.groups(); ++g) if (aside[g]) order_b.push_back(g);
                for (size_t i = nclear; i < far.size(); ++i) order_b.push_back(far[i]);
                found = true;
                break;
            }
            if (near_clean < W && far_clean >= W && switches < 3) {
                // the reference's slice has too small a share of this pool: set it aside (class B material) and judge from a group of the
                // class that is plentiful -- whatever turns up later in ANY other slice then counts for B
                ++switches;
                for (size_t g : near) if (t[g] >= 1.16f * t_self || g == ref) aside[g] = 1;
                ref = far[0];
                for (size_t g : far) if (t[g] < t[ref]) ref = g;
                std::fill(t.begin(), t.end(), 0.f);
                stagnant = 0; jump = 0;
                continue;
            }
            // one class is short: more memory, measured against the same reference.  What the extension brought decides how the next
            // one is made: nothing of the short class -> the next one jumps (16, 32, 64 GiB of spacer); a pool without ANY group of the
            // other class jumps at once
            const size_t have = std::min(near_clean, far_clean + n_aside), before = pool.groups();
            if (have <= 1 && !jump) jump = gib(16);
            if (!extend()) break;
            t.resize(pool.groups(), 0.f); aside.resize(pool.groups(), 0);
            size_t nc = near_clean, fc = far_clean;
            for (size_t g = before; g < pool.groups() && ok; ++g) {
                t[g] = probe_class(ref, g, t_self);
                nc += t[g] >= 1.16f * t_self ? 1 : 0;
                fc += t[g] <= 1.07f * t_self ? 1 : 0;
            }
            const size_t now = std::min(nc, fc + n_aside);
            stagnant = now >= have + std::min((size_t)3, W - have) ? 0 : stagnant + 1;
            jump = stagnant ? std::min(gib(16) << std::min(stagnant - 1, 2), gib(64)) : 0;
        }
    }
    // ---- step 2: the block's windows, each a (near, far) pair timed as the pair it will be
    struct Pair { size_t a, b; float us; };
    std::vector<Pair> pairs;
    const float accept = 1.05f * fast_level;                     // far partners scatter by +-2 %, near ones start 13 % above
    if (ok && found) {
        std::vector<size_t> qa(order_a.begin(), order_a.end()), qb(order_b.begin(), order_b.end());
        std::vector<Pair> poor;                                  // pairs that missed `accept`: the least bad ones fill in when the pool runs dry
        size_t ia = 0, ib = 0;
        while (pairs.size() < W && ia < qa.size() && ib < qb.size() && ok) {
            const size_t a = qa[ia], b = qb[ib];
            const float tab = probe(a, b);
            if (tab <= accept) { pairs.push_back(Pair{a, b, tab}); ++ia; ++ib; continue; }
            // one of the two is not what its class says (a group that straddles two stretches): a second partner tells which
            if (ib + 1 < qb.size()) {
                const size_t b2 = qb[ib + 1];
                const float tab2 = probe(a, b2);
                if (debug) std::fprintf(stderr, "snac_traj_alloc: pair (%zu, %zu) %.0f us > %.0f; (%zu, %zu) %.0f us\n", a, b, tab, accept, a, b2, tab2);
                if (tab2 <= accept) {                            // b was the odd one: drop it, a goes with b2
                    pairs.push_back(Pair{a, b2, tab2});
                    poor.push_back(Pair{a, b, tab});
                    ++ia; ib += 2;
                    continue;
                }
                poor.push_back(Pair{a, tab2 < tab ? b2 : b, tab2 < tab ? tab2 : tab});
            } else {
                poor.push_back(Pair{a, b, tab});
            }
            ++ia;                                                // a was the odd one: drop it, b stays for the next near group
        }
        if (pairs.size() < W) {
            // the pool ran out of clean pairs: the least bad ones make up the rest (the block says so: windows_slow)
            std::sort(poor.begin(), poor.end(), [](const Pair& x, const Pair& y) { return x.us < y.us; });
            std::vector<char> used(pool.groups(), 0);
            for (auto& p : pairs) used[p.a] = used[p.b] = 1;
            for (auto& p : poor) {
                if (pairs.size() >= W) break;
                if (used[p.a] || used[p.b]) continue;
                used[p.a] = used[p.b] = 1;
                pairs.push_back(p);
            }
        }
        if (pairs.size() < W) found = false;
        if (debug && found) {
            std::fprintf(stderr, "snac_traj_alloc: fast level %.0f us per GiB, slow %.0f, accept %.0f; %zu windows:", fast_level, slow_level, accept, pairs.size());
            for (auto& p : pairs) std::fprintf(stderr, " (%zu,%zu) %.0f", p.a, p.b, p.us);
            std::fprintf(stderr, "\n");
        }
    }
    if (ok) (void)hipStreamSynchronize(stream);
    if (!ok || !found) {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        if (debug) std::fprintf(stderr, "snac_traj_alloc: [%.0f ms] no block from this pool (%zu groups, %.0f GiB of spacers)\n", since(), pool.groups(), (double)pool.spacer_bytes / 1073741824.0);
        return ok ? give_up_measured() : give_up();
    }
    // ---- step 3: the block -- chunk j lies in window j / 32, there in the near group (even) or the far group (odd)
    std::vector<char> used(pool.h.size(), 0);
    std::vector<hipMemGenericAllocationHandle_t> hs(k);
    for (size_t j = 0; j < k; ++j) {
        const Pair& p = pairs[j / TRAJ_WINDOW];
        const size_t q = j % TRAJ_WINDOW, idx = ((q & 1) ? p.b : p.a) * TRAJ_GROUP + (q >> 1);
        hs[j] = pool.h[idx]; used[idx] = 1;
    }
    const size_t pool_groups = pool.groups();
    pool.drop(&used);                                            // the probe ranges stay reserved, unused
    char* va = nullptr;
    const size_t total = k * chunk;
    auto done_events = [&]() { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); };
    hipError_t e = hipMemAddressReserve((void**)&va, total, gran, nullptr, 0);
    if (e != hipSuccess) { done_events(); traj_release(nullptr, 0, chunk, hs); return fail_hip(e, "hipMemAddressReserve"); }
    size_t m2 = 0;
    for (size_t j = 0; j < k; ++j) {
        e = hipMemMap(va + j * chunk, chunk, 0, hs[j], 0);
        if (e != hipSuccess) { done_events(); traj_release(va, m2, chunk, hs); g_reserved_dead += total - m2; return fail_hip(e, "hipMemMap"); }
        m2 += chunk;
    }
    e = hipMemSetAccess(va, total, &pool.acc, 1);
    if (e != hipSuccess) { done_events(); traj_release(va, m2, chunk, hs); return fail_hip(e, "hipMemSetAccess"); }
    // the finished block under the same pattern: EVERY window (the faster of two launches each), then all of it in one launch
    std::memset(info, 0, sizeof(*info));
    const float accept_final = 1.08f * fast_level;
    float wmax = 0.f, wsum = 0.f;
    int wslow = 0, wn = 0;
    launch_probe(va, va + chunk, TRAJ_CHUNK_LOG2 + 1, (int)std::min(k, TRAJ_WINDOW), 8, false);
    for (size_t w = 0; w * TRAJ_WINDOW < k && ok; ++w) {
        const size_t nch = std::min(TRAJ_WINDOW, k - w * TRAJ_WINDOW);
        if (nch < 4) break;                                      // a stub of a last window: too short to time
        char* const wa = va + w * TRAJ_WINDOW * chunk;
        const float t1 = launch_probe(wa, wa + chunk, TRAJ_CHUNK_LOG2 + 1, (int)nch, 1, true), t2 = launch_probe(wa, wa + chunk, TRAJ_CHUNK_LOG2 + 1, (int)nch, 1, true);
        const float us = 1000.f * (t1 < t2 ? t1 : t2) * (float)TRAJ_WINDOW / (float)nch;    // per GiB
        if (wn < SNAC_TRAJ_INFO_WINDOWS) info->window_us[wn] = us;
        ++wn;
        wmax = us > wmax ? us : wmax; wsum += us;
        wslow += us > accept_final ? 1 : 0;
    }
    float whole = 0.f;
    if (ok) {
        const float t1 = launch_probe(va, va + chunk, TRAJ_CHUNK_LOG2 + 1, (int)k, 1, true), t2 = launch_probe(va, va + chunk, TRAJ_CHUNK_LOG2 + 1, (int)k, 1, true);
        whole = 1000.f * (t1 < t2 ? t1 : t2) * (float)TRAJ_WINDOW / (float)k;
    }
    done_events();
    if (!ok) { traj_release(va, m2, chunk, hs); return fail(SNAC_ERR_HIP, "probe of the finished block failed"); }
    info->layout = SNAC_TRAJ_MEASURED;
    info->bytes = total;
    info->pool_groups = (int32_t)pool_groups;
    info->probe_launches = launches;
    info->windows = wn;
    info->windows_slow = wslow;
    info->self_us_per_gib = t_self;
    info->fast_us_per_gib = fast_level;
    info->slow_us_per_gib = slow_level;
    info->window_max_us_per_gib = wmax;
    info->window_mean_us_per_gib = wn ? wsum / (float)wn : 0.f;
    info->block_us_per_gib = whole;
    if (debug) {
        std::fprintf(stderr, "snac_traj_alloc: finished block %.0f us per GiB as a whole, windows mean %.0f max %.0f (accepted up to %.0f)%s:", whole,
                     info->window_mean_us_per_gib, wmax, accept_final, wslow ? " -- SLOW" : "");
        for (int w = 0; w < wn && w < SNAC_TRAJ_INFO_WINDOWS; ++w) std::fprintf(stderr, " %.0f", info->window_us[w]);
        std::fprintf(stderr, "\n");
    }
    {
        std::lock_guard<std::mutex> lk(g_traj_mu);
        g_traj[va] = TrajBlock{total, chunk, device, SNAC_TRAJ_MEASURED, std::move(hs), *info};
    }
    *out = va;
    return 0;
}

int traj_alloc_fixed(size_t bytes, int device, const hipMemAllocationProp& prop, size_t gran, void** out) {
    const size_t chunk = bytes >= TRAJ_CHUNK ? ((TRAJ_CHUNK + gran - 1) / gran) * gran : ((bytes + gran - 1) / gran) * gran;
    const size_t k = (bytes + chunk - 1) / chunk, total = k * chunk;
    const int runs = (bytes >= TRAJ_SPLIT_MIN && k >= (size_t)TRAJ_RUNS) ? TRAJ_RUNS : 1;   // run r: chunks r, r + runs, r + 2 runs, ...
    char* va = nullptr;
    hipError_t e = hipMemAddressReserve((void**)&va, total, gran, nullptr, 0);
    if (e != hipSuccess) return fail_hip(e, "hipMemAddressReserve");
    std::vector<hipMemGenericAllocationHandle_t> hs(k), gap;
    std::vector<size_t> made;                                      // chunk indices whose handles exist, in creation order
    made.reserve(k);
    auto drop_gap = [&]() { for (auto g : gap) (void)hipMemRelease(g); gap.clear(); };
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = 0; }
    // the gaps are transient too: never more than half of what is free next to the block, never the last 4 GiB
    size_t gap_budget = free_b > bytes + TRAJ_MARGIN ? std::min((free_b - bytes) / 2, free_b - bytes - TRAJ_MARGIN) / chunk : 0;
    for (int r = 0; r < runs; ++r) {
        size_t in_run = 0;
        for (size_t c = (size_t)r; c < k; c += (size_t)runs) {
            hipError_t ce = hipMemCreate(&hs[c], chunk, &prop, 0);
            if (ce != hipSuccess && !gap.empty()) {                 // the gaps hold what this run needs: give them back, try once more
                drop_gap();
                (void)hipGetLastError();
                ce = hipMemCreate(&hs[c], chunk, &prop, 0);
            }
            if (ce != hipSuccess) {
                std::vector<hipMemGenericAllocationHandle_t> have;
                for (size_t q : made) have.push_back(hs[q]);
                drop_gap();
                traj_release(va, 0, chunk, have);
                g_reserved_dead += total;
                return fail_hip(ce, "hipMemCreate (out of device memory?)");
            }
            made.push_back(c);
            ++in_run;
        }
        if (r + 1 < runs) {
            // the gap: as many handles as bring the next run's start 32 GiB behind this one's; best effort (a full device gets less)
            const size_t run_bytes = in_run * chunk;
            size_t want = run_bytes < TRAJ_SLICE ? (TRAJ_SLICE - run_bytes) / chunk : 0;
            want = std::min(want, gap_budget > gap.size() ? gap_budget - gap.size() : (size_t)0);
            for (size_t g = 0; g < want; ++g) {
                hipMemGenericAllocationHandle_t x;
                if (hipMemCreate(&x, chunk, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
                gap.push_back(x);
            }
        }
    }
    drop_gap();
    size_t mapped = 0;
    for (size_t j = 0; j < k; ++j) {
        e = hipMemMap(va + j * chunk, chunk, 0, hs[j], 0);
        if (e != hipSuccess) {
            traj_release(va, mapped, chunk, hs);
            g_reserved_dead += total - mapped;
            return fail_hip(e, "hipMemMap");
        }
        mapped += chunk;
    }
    hipMemAccessDesc acc;
    std::memset(&acc, 0, sizeof(acc));
    acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    e = hipMemSetAccess(va, total, &acc, 1);
    if (e != hipSuccess) { traj_release(va, mapped, chunk, hs); return fail_hip(e, "hipMemSetAccess"); }
    snac_traj_info info;
    std::memset(&info, 0, sizeof(info));
    info.layout = runs > 1 ? SNAC_TRAJ_THREE_RUNS : SNAC_TRAJ_ONE_RUN;
    info.bytes = total;
    {
        std::lock_guard<std::mutex> lk(g_traj_mu);
        g_traj[va] = TrajBlock{total, chunk, device, info.layout, std::move(hs), info};
    }
    *out = va;
    return SNAC_OK;
}

// take a block out of the registry and give its memory back (the caller has made sure nothing uses it any more)
int traj_unregister_and_release(void* ptr) {
    TrajBlock b;
    {
        std::lock_guard<std::mutex> lk(g_traj_mu);
        auto it = g_traj.find(ptr);
        if (it == g_traj.end()) return fail(SNAC_ERR_ARG, "not a block of snac_traj_alloc");
        b = std::move(it->second);
        g_traj.erase(it);
    }
    traj_release((char*)ptr, b.total, b.chunk, b.handles);
    return SNAC_OK;
}
}  // namespace

extern "C" {

int snac_traj_alloc_ex(size_t bytes, int device, size_t pool_cap_bytes, void* stream, void** out) {
    if (!out) return fail(SNAC_ERR_ARG, "null out");
    *out = nullptr;
    if (bytes == 0) return fail(SNAC_ERR_ARG, "bytes must be positive");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess) return fail_hip(e, "hipGetDeviceCount");
    if (device < 0 || device >= ndev) return fail(SNAC_ERR_ARG, "device out of range");
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (cur != device) (void)hipSetDevice(device);
    struct Restore { int cur, dev; ~Restore() { if (cur >= 0 && cur != dev) (void)hipSetDevice(cur); } } restore{cur, device};
    hipStream_t s = (hipStream_t)stream;
    hipMemAllocationProp prop;
    std::memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = device;
    size_t gran = 0;
    e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
    if (e != hipSuccess) return fail_hip(e, "hipMemGetAllocationGranularity");
    if (gran < (((size_t)2 << 20) >> SNAC_TRAJ_TEST_SHIFT)) gran = ((size_t)2 << 20) >> SNAC_TRAJ_TEST_SHIFT;   // whole 2 MB pages whatever the minimum is
    const size_t cap = pool_cap_bytes ? pool_cap_bytes : TRAJ_POOL_DEFAULT;
    const char* off = std::getenv("SNAC_TRAJ_PROBE");
    const bool measured = bytes >= TRAJ_SPLIT_MIN && !(off && off[0] == '0');
    const auto t0 = std::chrono::steady_clock::now();
    auto build_ms = [&]() { return (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    auto stamp = [&](void* p, int rebuilds) {
        std::lock_guard<std::mutex> lk(g_traj_mu);
        auto it = g_traj.find(p);
        if (it != g_traj.end()) { it->second.info.rebuilds = rebuilds; it->second.info.build_ms = build_ms(); }
    };
    // the measured layout first; a block with a slow window is built again from a larger pool (twice at most; the last one stands, and
    // says so in its description)
    constexpr int ATTEMPTS = 3;
    for (int attempt = 0; attempt < ATTEMPTS && measured; ++attempt) {
        snac_traj_info info;
        const int rc = traj_alloc_probed(bytes, device, prop, gran, cap, (size_t)attempt * TRAJ_GROW, s, out, &info);
        if (rc < 0) return rc;
        if (rc == 1) break;                                      // not to be had: the fixed layout
        if (rc == 2) continue;                                   // this pool did not do: another one, larger from the start
        const int vr = traj_verify((char*)*out, ((bytes + TRAJ_CHUNK - 1) / TRAJ_CHUNK) * TRAJ_CHUNK, TRAJ_CHUNK, s);
        if (vr == SNAC_OK && (info.windows_slow == 0 || attempt + 1 == ATTEMPTS)) { stamp(*out, attempt); return SNAC_OK; }
        (void)traj_unregister_and_release(*out);                 // (the stream was synchronised by the check)
        *out = nullptr;
        if (vr != SNAC_OK) return vr;                            // a block that does not read back what was written: report, never retry silently
    }
    const int rc = traj_alloc_fixed(bytes, device, prop, gran, out);
    if (rc != SNAC_OK) return rc;
    size_t total = 0, chunk = 0;
    {
        std::lock_guard<std::mutex> lk(g_traj_mu);
        const TrajBlock& b = g_traj[*out];
        total = b.total; chunk = b.chunk;
    }
    const int vr = traj_verify((char*)*out, total, chunk, s);
    if (vr != SNAC_OK) { (void)traj_unregister_and_release(*out); *out = nullptr; }
    else stamp(*out, 0);
    return vr;
}

int snac_traj_alloc(size_t bytes, int device, void** out) { return snac_traj_alloc_ex(bytes, device, 0, nullptr, out); }

int snac_traj_layout(const void* ptr) {
    std::lock_guard<std::mutex> lk(g_traj_mu);
    auto it = g_traj.find(const_cast<void*>(ptr));
    return it == g_traj.end() ? fail(SNAC_ERR_ARG, "not a block of snac_traj_alloc") : it->second.layout;
}

int snac_traj_describe(const void* ptr, snac_traj_info* out) {
    if (!out) return fail(SNAC_ERR_ARG, "null out");
    std::lock_guard<std::mutex> lk(g_traj_mu);
    auto it = g_traj.find(const_cast<void*>(ptr));
    if (it == g_traj.end()) return fail(SNAC_ERR_ARG, "not a block of snac_traj_alloc");
    *out = it->second.info;
    return SNAC_OK;
}

uint64_t snac_traj_reserved_bytes(void) { return (uint64_t)g_reserved_dead.load(); }

int snac_traj_free(void* ptr) {
    if (!ptr) return SNAC_OK;
    int dev = -1;
    {
        std::lock_guard<std::mutex> lk(g_traj_mu);
        auto it = g_traj.find(ptr);
        if (it == g_traj.end()) return fail(SNAC_ERR_ARG, "not a block of snac_traj_alloc");
        dev = it->second.device;
    }
    int cur = -1;                                                // nothing may still be writing into the block: its device goes idle
    (void)hipGetDevice(&cur);
    if (cur != dev) (void)hipSetDevice(dev);
    (void)hipDeviceSynchronize();
    if (cur >= 0 && cur != dev) (void)hipSetDevice(cur);
    return traj_unregister_and_release(ptr);
}

}  // extern "C"
