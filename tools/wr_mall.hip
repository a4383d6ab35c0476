// wr_mall.hip -- can the 256 MB Infinity Cache absorb a rollout's observation rows?  (a) The rollout's persistent store pattern
// (a wave owns 64 envs, 64 row stores of 408 B per tick) over 600 ticks, but into a RING of R ticks (R x 26.7 MB) instead of the
// full [600][N][51] tensor: does the rate rise when the ring fits the cache?  (b) a relay: one-shot workgroups copy 4 KB chunks
// from a ring (read) to the full tensor (written in address order): the rate of that copy.
// build: hipcc -O3 --offload-arch=gfx950 -o wr_mall wr_mall.hip ; run: ./wr_mall
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int E = 64, D = 51;

__global__ __launch_bounds__(256) void k_ring(double* out, int N, int T, int R) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int env0 = wave * E;
    if (env0 >= N) return;
    for (int t = 0; t < T; ++t) {
        double* base = out + ((size_t)(t % R) * N + env0) * D;
#pragma unroll 8
        for (int e = 0; e < E; ++e)
            if (lane < D) base[e * D + lane] = (double)(t + lane + e);
    }
}

typedef double v2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_copy(const v2* src, v2* dst, size_t src_quads) {   // one 4 KB chunk per workgroup
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    dst[i] = src[i % src_quads];
}

template <typename F> float best_of(F launch) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); float best = 1e30f;
    for (int it = 0; it < 6; ++it) { CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (it >= 2 && ms < best) best = ms; }
    return best;
}

int main() {
    const int N = 65536, T = 600;
    const size_t tick = (size_t)N * D * 8, bytes = tick * T;
    double* full; CK(hipMalloc(&full, bytes));
    const int blocks = N / E / 4;
    for (int R : {600, 64, 16, 8, 4, 2, 1}) {
        float ms = best_of([&] { hipLaunchKernelGGL(k_ring, dim3(blocks), dim3(256), 0, 0, full, N, T, R); });
        printf("(a) persistent rows into a ring of %3d ticks (%7.1f MB): %6.3f ms  %5.2f TB/s of stores\n", R, R * tick / 1e6, ms, bytes / ms / 1e9);
    }
    double* ring; CK(hipMalloc(&ring, tick * 16));
    CK(hipMemset(ring, 0, tick * 16));
    for (int R : {16, 4, 1}) {
        const size_t quads = bytes / 16;
        float ms = best_of([&] { hipLaunchKernelGGL(k_copy, dim3((unsigned)(quads / 256)), dim3(256), 0, 0, (const v2*)ring, (v2*)full, R * tick / 16); });
        printf("(b) one-shot 4 KB copy, source ring of %2d ticks (%6.1f MB) -> full tensor: %6.3f ms  %5.2f TB/s written\n", R, R * tick / 1e6, ms, bytes / ms / 1e9);
    }
    return 0;
}
