// traj_host_test.cpp -- the host logic of trajectory memory (snac_amd/csrc/snac_traj.hip, the very file libsnac_hip.so compiles: pools,
// 32 MB handles, spacers, windows, rebuilds, the block registry, reserved ranges) against tests/native/fakehip, with the geometry
// shrunk by 2^13 (SNAC_TRAJ_TEST_SHIFT: 4 KB chunks, "1 GiB" = 128 KB) so that every path runs on real memory in milliseconds.  Built by
// tests/test_native_host_logic.py with gcc -fsanitize=address,undefined.  Test infrastructure.
// The fake device has HBM in slices (writes confined to one slice are slow, spread over two fast): the cost model below prices the
// probe kernel by the simulated physical addresses of the chunks it would write.
#define SNAC_TRAJ_TEST_SHIFT 13
#define SNAC_TRAJ_AUX_BLOCKS 8
#include "../../snac_amd/csrc/snac_traj.hip"

#include <cmath>
#include <cstdarg>
#include <map>

namespace snac_detail { thread_local char g_err[256] = ""; }
extern "C" const char* snac_last_error(void) { return snac_detail::g_err; }

static int g_failed = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAILED line %d: %s   (last error: %s)\n", __LINE__, #c, snac_last_error()); ++g_failed; } } while (0)

static constexpr size_t GIB = (size_t)1 << 17;                        // one scaled GiB
static constexpr size_t SLICE = 32 * GIB;
enum { PHYSICS_SLICES, PHYSICS_FLAT, PHYSICS_SLOW_BLOCK };
static int g_physics = PHYSICS_SLICES;
static uint32_t g_noise = 12345u;

// microseconds of one launch: the probe is priced by where its chunks lie, everything else costs a microsecond
static double cost(const fakehip::Launch& l) {
    if (std::strcmp(l.name, "k_traj_probe") != 0 || l.ptrs.size() < 2 || l.ints.size() < 4) return 1.0;
    char* const a = (char*)l.ptrs[0];
    char* const b = (char*)l.ptrs[1];
    const int chunk_log2 = (int)l.ints[0], pair_log2 = (int)l.ints[1], chunks = (int)l.ints[2], reps = (int)l.ints[3];
    if (a == b && g_physics != PHYSICS_FLAT) {                       // a group paired with ITSELF: the same chunks written twice run at the fast level
        g_noise = g_noise * 1664525u + 1013904223u;
        return 147.0 * (1.0 + 0.01 * ((double)(g_noise >> 16) / 65536.0 - 0.5)) * (double)chunks / 32.0 * (double)reps;
    }
    std::map<uint64_t, int> per_slice;
    for (int c = 0; c < chunks; ++c) {
        char* const at = ((c & 1) ? b : a) + ((size_t)(c >> 1) << pair_log2);
        const std::vector<uint64_t> ph = fakehip::phys_at(at, (size_t)1 << chunk_log2);
        if (ph.empty()) return 1e9;                                  // a probe of unmapped memory: on the GPU a fault
        per_slice[ph[0] / SLICE] += 1;
    }
    int top = 0;
    for (auto& kv : per_slice) top = std::max(top, kv.second);
    const double share = (double)top / (double)chunks;              // 1.0: all in one slice; 0.5: two slices in turn
    double per_gib = g_physics == PHYSICS_FLAT ? 150.0 : (share <= 0.55 ? 150.0 : (share >= 0.95 ? 190.0 : 150.0 + 40.0 * (share - 0.55) / 0.40));
    if (g_physics == PHYSICS_SLOW_BLOCK && pair_log2 == chunk_log2 + 1) per_gib = 185.0;   // the FINISHED block never reaches the fast level
    g_noise = g_noise * 1664525u + 1013904223u;
    per_gib *= 1.0 + 0.01 * ((double)(g_noise >> 16) / 65536.0 - 0.5);
    return per_gib * (double)chunks / 32.0 * (double)reps;
}

static bool same(const fakehip::Counts& a, const fakehip::Counts& b) {
    return a.handles == b.handles && a.mappings == b.mappings && a.mallocs == b.mallocs && a.events == b.events && a.streams == b.streams;
}
static void fill_and_check(void* p, size_t bytes, uint64_t salt) {
    uint64_t* w = (uint64_t*)p;
    const size_t n = bytes / 8;
    for (size_t i = 0; i < n; i += 509) w[i] = i * 0x9E3779B97F4A7C15ull + salt;
    w[n - 1] = salt;
    bool ok = w[n - 1] == salt;
    for (size_t i = 0; i < n - 1; i += 509) ok = ok && w[i] == i * 0x9E3779B97F4A7C15ull + salt;
    CHECK(ok);
}

static void case_raw_abi() {
    const fakehip::Counts base = fakehip::counts();
    void* p = (void*)1;
    CHECK(snac_traj_alloc(0, 0, &p) == SNAC_ERR_ARG && p == nullptr);
    CHECK(snac_traj_alloc(4096, 7, &p) == SNAC_ERR_ARG && std::strstr(snac_last_error(), "device"));
    CHECK(snac_traj_alloc(4096, -1, &p) == SNAC_ERR_ARG);
    CHECK(snac_traj_alloc(4096, 0, nullptr) == SNAC_ERR_ARG);
    CHECK(snac_traj_alloc(3 * 4096 + 17, 0, &p) == SNAC_OK && p);          // below the split size: one run
    CHECK(snac_traj_layout(p) == SNAC_TRAJ_ONE_RUN);
    fill_and_check(p, 3 * 4096 + 17, 5);
    snac_traj_info info;
    CHECK(snac_traj_describe(p, &info) == SNAC_OK && info.layout == SNAC_TRAJ_ONE_RUN && info.bytes >= 3 * 4096 + 17);
    CHECK(snac_traj_describe(p, nullptr) == SNAC_ERR_ARG && snac_traj_describe((char*)p + 8, &info) == SNAC_ERR_ARG);
    int x = 0;
    CHECK(snac_traj_free(&x) == SNAC_ERR_ARG && std::strstr(snac_last_error(), "snac_traj_alloc"));   // a foreign pointer
    CHECK(snac_traj_free(p) == SNAC_OK);
    CHECK(snac_traj_free(p) == SNAC_ERR_ARG);                        // already gone
    CHECK(snac_traj_free(nullptr) == SNAC_OK);
    CHECK(snac_traj_layout(p) == SNAC_ERR_ARG);
    CHECK(same(fakehip::counts(), base));
}

static void case_measured_blocks_and_turns() {
    const fakehip::Counts base = fakehip::counts();
    g_physics = PHYSICS_SLICES;
    // the headline's 16 "GiB" on a device with slices: a measured block, every window at the fast level
    void* p = nullptr;
    const size_t big = 16 * GIB - 4096 * 3;
    CHECK(snac_traj_alloc(big, 0, &p) == SNAC_OK && p);
    snac_traj_info info;
    CHECK(snac_traj_describe(p, &info) == SNAC_OK);
    CHECK(info.layout == SNAC_TRAJ_MEASURED && info.windows == 16 && info.windows_slow == 0 && info.rebuilds == 0);
    CHECK(info.fast_us_per_gib > 140.f && info.fast_us_per_gib < 160.f && info.slow_us_per_gib > 180.f && info.block_us_per_gib < 1.05f * info.fast_us_per_gib);
    CHECK(info.probe_launches > 20 && info.pool_groups >= 32 + 8);
    fill_and_check(p, big, 77);
    // ... and its chunks really take turns between two slices
    {
        const std::vector<uint64_t> ph = fakehip::phys_at(p, 64 * 4096);
        CHECK(ph.size() == 64);
        int differ = 0;
        for (size_t i = 0; i + 1 < ph.size(); ++i) differ += ph[i] / SLICE != ph[i + 1] / SLICE;
        CHECK(differ >= 60);
    }
    const unsigned long long dead0 = snac_traj_reserved_bytes();
    CHECK(snac_traj_free(p) == SNAC_OK);
    CHECK(snac_traj_reserved_bytes() >= dead0 + big);                // the range stays reserved (never handed out again)
    CHECK(same(fakehip::counts(), base));
    // 60 allocate / release turns of 1 .. 3 "GiB" with a long-lived neighbour that fragments the physical space
    void* keep = nullptr;
    CHECK(snac_traj_alloc(5 * GIB, 0, &keep) == SNAC_OK);
    int measured = 0;
    for (int i = 0; i < 60; ++i) {
        const size_t bytes = GIB + (size_t)(i % 5) * (GIB / 2) + 4096 * (size_t)(i % 3);
        void* q = nullptr;
        if (snac_traj_alloc_ex(bytes, 0, i % 2 ? 0 : 24 * GIB, nullptr, &q) != SNAC_OK || !q) { CHECK(!"alloc in a turn"); break; }
        const int lay = snac_traj_layout(q);
        CHECK(lay == SNAC_TRAJ_MEASURED || lay == SNAC_TRAJ_THREE_RUNS);
        measured += lay == SNAC_TRAJ_MEASURED;
        fill_and_check(q, bytes, (uint64_t)i);
        if (i % 7 == 3) {                                            // the neighbour comes and goes
            CHECK(snac_traj_free(keep) == SNAC_OK);
            CHECK(snac_traj_alloc((size_t)(3 + i % 4) * GIB, 0, &keep) == SNAC_OK);
        }
        CHECK(snac_traj_free(q) == SNAC_OK);
    }
    std::printf("  60 turns: %d measured blocks, %d on the fixed layout (a pool capped at 24 GiB cannot leave the neighbour's slice)\n", measured, 60 - measured);
    CHECK(measured >= 30);
    CHECK(snac_traj_free(keep) == SNAC_OK);
    const fakehip::Counts end = fakehip::counts();
    CHECK(same(end, base));
}

static void case_fallbacks_and_rebuilds() {
    const fakehip::Counts base = fakehip::counts();
    snac_traj_info info;
    void* p = nullptr;
    // no contrast anywhere (a device without slices): the fixed three-run layout, nothing of the pool left behind
    g_physics = PHYSICS_FLAT;
    CHECK(snac_traj_alloc(4 * GIB, 0, &p) == SNAC_OK && snac_traj_layout(p) == SNAC_TRAJ_THREE_RUNS);
    fill_and_check(p, 4 * GIB, 1);
    CHECK(snac_traj_free(p) == SNAC_OK && same(fakehip::counts(), base));
    // a pool cap that leaves nothing worth probing: the fixed layout at once (no probe launch)
    g_physics = PHYSICS_SLICES;
    const uint64_t l0 = fakehip::launches();
    CHECK(snac_traj_alloc_ex(2 * GIB, 0, GIB, nullptr, &p) == SNAC_OK && snac_traj_layout(p) == SNAC_TRAJ_THREE_RUNS);
    CHECK(fakehip::launches() - l0 <= 2);                            // the block's own check (fill + read back) only
    CHECK(snac_traj_free(p) == SNAC_OK);
    // a finished block that never reaches the fast level: rebuilt twice from larger pools, the third stands and says so
    g_physics = PHYSICS_SLOW_BLOCK;
    CHECK(snac_traj_alloc(2 * GIB, 0, &p) == SNAC_OK && snac_traj_describe(p, &info) == SNAC_OK);
    CHECK(info.layout == SNAC_TRAJ_MEASURED && info.rebuilds == 2 && info.windows_slow == info.windows && info.windows == 2);
    fill_and_check(p, 2 * GIB, 9);
    CHECK(snac_traj_free(p) == SNAC_OK);
    g_physics = PHYSICS_SLICES;
    // a device with little memory left: no room for a pool -> fixed layout; no room for the block -> an error and nothing leaked
    fakehip::set_device_memory(2 * GIB + 5 * GIB);
    CHECK(snac_traj_alloc(2 * GIB, 0, &p) == SNAC_OK && snac_traj_layout(p) == SNAC_TRAJ_THREE_RUNS);
    CHECK(snac_traj_free(p) == SNAC_OK);
    fakehip::set_device_memory(2 * GIB - 4096);
    p = (void*)1;
    CHECK(snac_traj_alloc(2 * GIB, 0, &p) == SNAC_ERR_HIP && p == nullptr && std::strstr(snac_last_error(), "hipMemCreate"));
    fakehip::set_device_memory(288 * GIB);
    CHECK(same(fakehip::counts(), base));
    // another device than the current one: the block is built there and the current device is restored
    fakehip::set_devices(2);
    int cur = -1;
    CHECK(snac_traj_alloc(GIB / 2, 1, &p) == SNAC_OK && hipGetDevice(&cur) == hipSuccess && cur == 0);
    CHECK(snac_traj_free(p) == SNAC_OK && hipGetDevice(&cur) == hipSuccess && cur == 0);
    fakehip::set_devices(1);
    CHECK(same(fakehip::counts(), base));
}

// every fallible call of an allocation failing in turn: the allocation either succeeds all the same (a fallback took over: the block
// is then whole) or returns an error with *out == NULL -- and in both cases nothing is left behind
static void case_every_call_fails_once(size_t bytes, size_t cap, long stride) {
    const fakehip::Counts base = fakehip::counts();
    g_physics = PHYSICS_SLICES;
    long k = 0, errors = 0, survived = 0, total = 0;
    for (;; k += (k < 120 ? 1 : stride)) {
        fakehip::fail_nth(k);
        void* p = (void*)1;
        const int rc = snac_traj_alloc_ex(bytes, 0, cap, nullptr, &p);
        const bool fired = fakehip::failure_fired();
        total = fakehip::fallible_calls();
        fakehip::fail_nth(-1);
        if (rc == SNAC_OK) {
            CHECK(p != nullptr && p != (void*)1);
            fill_and_check(p, bytes, (uint64_t)k);
            CHECK(snac_traj_free(p) == SNAC_OK);
            survived += fired;
        } else {
            CHECK(fired && p == nullptr && snac_last_error()[0] != 0);
            ++errors;
        }
        if (!same(fakehip::counts(), base)) { CHECK(!"something left behind"); std::printf("  at call index %ld of %ld (rc %d)\n", k, total, rc); break; }
        if (!fired) break;                                           // the index lies beyond the allocation's last call: done
    }
    std::printf("  %zu bytes: %ld fallible calls per allocation; failures injected up to index %ld: %ld ended in an error, %ld in a whole block\n",
                bytes, total, k, errors, survived);
    CHECK(errors > 0 && survived > 0);
}

int main() {
    fakehip::set_cost_model(cost);
    fakehip::set_device_memory(288 * GIB);
    case_raw_abi();
    case_measured_blocks_and_turns();
    case_fallbacks_and_rebuilds();
    case_every_call_fails_once(GIB / 4, 0, 1);                       // one run
    case_every_call_fails_once(2 * GIB + 4096, 0, 17);               // the measured layout, its pool and its fallbacks
    // every range ever reserved is still reserved (none is handed out twice: tools/vmm_stale.hip) and the library's own account of them
    // is exact: no block is alive now, so all of it is "dead"
    std::printf("  address space left reserved: %zu bytes in %ld ranges; snac_traj_reserved_bytes() = %llu\n", fakehip::counts().reserved_bytes,
                fakehip::counts().reservations, (unsigned long long)snac_traj_reserved_bytes());
    CHECK(fakehip::counts().reserved_bytes == snac_traj_reserved_bytes());
    std::printf(g_failed ? "traj_host_test: %d check(s) FAILED\n" : "traj_host_test: all checks passed\n", g_failed);
    return g_failed ? 1 : 0;
}
