// wr_vmm.hip -- can the ONE contiguous [T][N][51] tensor get the 7.1 TB/s of tools/wr_blocks.hip's split tensors?  The virtual
// range stays contiguous; its physical backing is dealt over runs of memory a 32 GiB slice apart with the virtual-memory API:
// hipMemCreate handles of 32 MB created one after the other (they follow each other in physical memory on an idle device), gap
// handles between the runs (released afterwards), virtual chunk j mapped to run j % runs.  This is the FALLBACK layout of
// snac_traj_alloc (snac_amd/csrc/snac_hip.hip): the runs show that the order of creation says little about where a handle lies once
// the allocator has seen releases -- which is why snac_traj_alloc measures instead.  Patterns: R = the rollout's rows, W = wave-major.
// build: hipcc -O3 --offload-arch=gfx950 -o wr_vmm wr_vmm.hip ; run: ./wr_vmm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s (%d) at line %d\n", hipGetErrorString(e_), (int)e_, __LINE__); exit(1); } } while (0)
constexpr int E = 64, D = 51, N = 65536, T = 600, WAVES = N / E;
constexpr size_t TILE = (size_t)E * D * 8;
__global__ __launch_bounds__(256) void k_r(char* out) {
    const int lane = threadIdx.x & 63; const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= WAVES) return;
    for (int t = 0; t < T; ++t) { double* base = (double*)(out + ((size_t)t * WAVES + wave) * TILE);
#pragma unroll 8
        for (int e = 0; e < E; ++e) if (lane < D) base[e * D + lane] = (double)(t + lane + e); }
}
__global__ __launch_bounds__(256) void k_w(char* out) {
    const int lane = threadIdx.x & 63; const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= WAVES) return;
    for (int t = 0; t < T; ++t) { double* base = (double*)(out + ((size_t)wave * T + t) * TILE);
#pragma unroll 8
        for (int e = 0; e < E; ++e) if (lane < D) base[e * D + lane] = (double)(t + lane + e); }
}
__global__ void k_check(const double* p, size_t n, unsigned long long* bad) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) if (p[i] == -7.0) atomicAdd(bad, 1ull);
}
template <typename F> float best_of(F launch) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); float best = 1e30f;
    for (int it = 0; it < 7; ++it) { CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (it >= 2 && ms < best) best = ms; }
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return best;
}
void report(const char* what, char* p, size_t bytes) {
    float r = best_of([&] { hipLaunchKernelGGL(k_r, dim3(256), dim3(256), 0, 0, p); });
    float w = best_of([&] { hipLaunchKernelGGL(k_w, dim3(256), dim3(256), 0, 0, p); });
    printf("%-72s R %5.2f  W %5.2f TB/s   (R %.3f ms)\n", what, bytes / r / 1e9, bytes / w / 1e9, r);
    fflush(stdout);
}
// one contiguous virtual range of `bytes`, backed by `runs` runs of 32 MB handles; consecutive runs start `dist` bytes apart in creation
// order (gap handles in between, released afterwards); virtual chunk j -> run j % runs.  dist = 0: the runs follow each other directly.
struct Block { char* va; size_t total, chunk; std::vector<hipMemGenericAllocationHandle_t> h; };
Block make(size_t bytes, int runs, size_t dist, int dev) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
    Block b; b.chunk = (size_t)32 << 20;
    const size_t k = ((bytes + b.chunk - 1) / b.chunk + runs - 1) / runs * runs, per = k / runs;
    b.total = k * b.chunk; b.h.resize(k);
    std::vector<hipMemGenericAllocationHandle_t> gap;
    for (int r = 0; r < runs; ++r) {
        for (size_t i = 0; i < per; ++i) CK(hipMemCreate(&b.h[i * runs + r], b.chunk, &prop, 0));
        if (r + 1 < runs && dist > per * b.chunk)
            for (size_t g = 0; g < (dist - per * b.chunk) / b.chunk; ++g) { hipMemGenericAllocationHandle_t x; CK(hipMemCreate(&x, b.chunk, &prop, 0)); gap.push_back(x); }
    }
    for (auto x : gap) CK(hipMemRelease(x));
    CK(hipMemAddressReserve((void**)&b.va, b.total, 0, nullptr, 0));
    for (size_t j = 0; j < k; ++j) CK(hipMemMap(b.va + j * b.chunk, b.chunk, 0, b.h[j], 0));
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(b.va, b.total, &acc, 1));
    return b;
}
void drop(Block& b) {
    CK(hipDeviceSynchronize());
    for (size_t off = 0; off < b.total; off += b.chunk) CK(hipMemUnmap(b.va + off, b.chunk));
    for (auto x : b.h) CK(hipMemRelease(x));
}
int main() {
    int dev = 0; CK(hipSetDevice(dev));
    const size_t bytes = (size_t)T * WAVES * TILE;
    const size_t G = (size_t)1 << 30;
    printf("target %.2f GB: R = the rollout's rows in the reference order [T][N][51], W = wave-major; store-only, TB/s\n", bytes / 1e9);
    for (int round = 0; round < 4; ++round) {
        printf("-- round %d%s\n", round, round ? " (the allocator has seen allocations and releases by now)" : " (fresh process)");
        { char* p; CK(hipMalloc((void**)&p, bytes)); report("hipMalloc (one contiguous run)", p, bytes); CK(hipFree(p)); }
        struct V { const char* what; int runs; size_t dist; } vs[] = {
            {"virtual memory, one run of 32 MB handles", 1, 0},
            {"virtual memory, two runs back to back (7.5 GiB apart), chunks taking turns", 2, 0},
            {"virtual memory, two runs 16 GiB apart, chunks taking turns", 2, 16 * G},
            {"virtual memory, two runs 32 GiB apart, chunks taking turns", 2, 32 * G},
            {"virtual memory, two runs 64 GiB apart, chunks taking turns", 2, 64 * G},
            {"virtual memory, three runs 32 GiB apart, chunks taking turns", 3, 32 * G},
            {"virtual memory, four runs 16 GiB apart, chunks taking turns", 4, 16 * G},
            {"virtual memory, four runs 24 GiB apart, chunks taking turns", 4, 24 * G},
        };
        for (auto& v : vs) { Block b = make(bytes, v.runs, v.dist, dev); report(v.what, b.va, bytes); drop(b); }
    }
    return 0;
}
