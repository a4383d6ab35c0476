// wr_vmm.hip -- can the ONE contiguous [T][N][51] tensor get the 7.1 TB/s of tools/wr_blocks.hip's split tensors?  The virtual
// range stays contiguous; its physical backing is dealt over two (or four) far-apart pieces with the virtual-memory API:
// hipMemCreate handles of `chunk` bytes created one after the other (assumed to follow each other in physical memory), virtual
// chunk j mapped to handle (j % B) * (k / B) + j / B.  Patterns: R = the rollout's rows (plain addresses), W = wave-major.
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
int main() {
    int dev = 0; CK(hipSetDevice(dev));
    const size_t bytes = (size_t)T * WAVES * TILE;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
    size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    size_t gmin = 0; CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
    printf("allocation granularity: minimum %zu, recommended %zu bytes\n", gmin, gran);
    { char* p; CK(hipMalloc((void**)&p, bytes)); report("hipMalloc", p, bytes); CK(hipFree(p)); }
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    for (size_t chunk_mb : {2048, 256, 32, 2}) {
        const size_t chunk = chunk_mb << 20;
        if (chunk % gran) { printf("chunk %zu MB is not a multiple of the granularity\n", chunk_mb); continue; }
        const size_t k = (bytes + chunk - 1) / chunk;                 // chunks of the tensor
        // handles: the tensor's k chunks plus a gap of `gap` chunks between the halves, released after mapping
        for (int B : {1, 2, 4}) {
            const size_t kk = ((k + B - 1) / B) * B, per = kk / B;
            const size_t gap = B == 1 ? 0 : ((size_t)8 << 30) / chunk;      // >= 8 GiB of other memory between the pieces
            std::vector<hipMemGenericAllocationHandle_t> h(kk), pad;
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            size_t made = 0;
            for (int b = 0; b < B; ++b) {
                for (size_t i = 0; i < per; ++i) CK(hipMemCreate(&h[made++], chunk, &prop, 0));
                if (b + 1 < B) for (size_t i = 0; i < gap; ++i) { hipMemGenericAllocationHandle_t g; CK(hipMemCreate(&g, chunk, &prop, 0)); pad.push_back(g); }
            }
            char* va = nullptr; CK(hipMemAddressReserve((void**)&va, kk * chunk, 0, nullptr, 0));
            for (size_t j = 0; j < kk; ++j) CK(hipMemMap(va + j * chunk, chunk, 0, h[(j % B) * per + j / B], 0));
            CK(hipMemSetAccess(va, kk * chunk, &acc, 1));
            for (auto g : pad) CK(hipMemRelease(g));
            char what[160];
            snprintf(what, sizeof what, "virtual memory: %zu chunks of %zu MB dealt over %d piece(s), >= 8 GiB between", kk, chunk_mb, B);
            report(what, va, bytes);
            CK(hipMemUnmap(va, kk * chunk));
            for (auto x : h) CK(hipMemRelease(x));
            CK(hipMemAddressFree(va, kk * chunk));
            CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
            if (chunk_mb == 2 && B == 2) break;                         // the 2 MB case: one layout is enough (7650 handles)
        }
    }
    return 0;
}
