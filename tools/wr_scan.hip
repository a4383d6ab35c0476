// wr_scan.hip -- where in memory is a 16 GB streaming target fast?  One 112 GB allocation, the wave-major store-only pattern of
// tools/wr_frontier.hip (W: the most allocation-sensitive one, 5.7 vs 7.1 TB/s) and the rollout's own pattern (B) over a 16 GB
// window at offsets 0, 2, 4, ... GB; and I, the whole-granule 512-env workgroups on the reference layout.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int E = 64, D = 51;
__global__ __launch_bounds__(256) void k_w(double* out, int N, int T) {
    const int lane = threadIdx.x & 63; const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave * E >= N) return;
    for (int t = 0; t < T; ++t) { double* base = out + ((size_t)wave * T + t) * E * D;
#pragma unroll 8
        for (int e = 0; e < E; ++e) if (lane < D) base[e * D + lane] = (double)(t + lane + e); }
}
__global__ __launch_bounds__(256) void k_b(double* out, int N, int T) {
    const int lane = threadIdx.x & 63; const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int env0 = wave * E; if (env0 >= N) return;
    for (int t = 0; t < T; ++t) { double* base = out + ((size_t)t * N + env0) * D;
#pragma unroll 8
        for (int e = 0; e < E; ++e) if (lane < D) base[e * D + lane] = (double)(t + lane + e); }
}
// I of wr_frontier.hip: 512-env workgroups writing whole 4 KB granules of [T][N][51] in address order
typedef double v2s __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(512) void k_i(double* out, int N, int T) {
    const int env0 = blockIdx.x * 512; if (env0 >= N) return;
    for (int t = 0; t < T; ++t) { v2s* base = (v2s*)(out + ((size_t)t * N + env0) * D);
        for (int q = threadIdx.x; q < 512 * D / 2; q += 512) base[q] = (v2s){(double)(t + q), (double)(t - q)}; }
}
template <typename F> float best_of(F launch) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); float best = 1e30f;
    for (int it = 0; it < 5; ++it) { CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (it >= 2 && ms < best) best = ms; }
    return best;
}
int main(int argc, char** argv) {
    const int N = 65536, T = 600; const size_t bytes = (size_t)T * N * D * 8;
    const size_t total = (size_t)(argc > 1 ? atoi(argv[1]) : 112) << 30;
    char* big; CK(hipMalloc((void**)&big, total));
    printf("one allocation of %zu GiB at %p; 16.04 GB window at each offset: W wave-major / B rollout rows, TB/s\n", total >> 30, (void*)big);
    for (size_t off = 0; off + bytes <= total; off += (size_t)2 << 30) {
        double* out = (double*)(big + off);
        float w = best_of([&] { hipLaunchKernelGGL(k_w, dim3(256), dim3(256), 0, 0, out, N, T); });
        float b = best_of([&] { hipLaunchKernelGGL(k_b, dim3(256), dim3(256), 0, 0, out, N, T); });
        float i = best_of([&] { hipLaunchKernelGGL(k_i, dim3(N / 512), dim3(512), 0, 0, out, N, T); });
        printf("offset %3zu GiB: W %5.2f   B %5.2f   I %5.2f\n", off >> 30, bytes / w / 1e9, bytes / b / 1e9, bytes / i / 1e9);
        fflush(stdout);
    }
    return 0;
}
