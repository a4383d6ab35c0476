// wr_bench.hip -- store-pattern micro-benchmark for the rollout kernel's observation stream (MI355X).
// Output tensor [T][N][51] float64.  Each "owner" (a wave) holds E consecutive envs and, for t = 0..T-1, writes
// their E*408 contiguous bytes.  Variants differ in how a wave issues those bytes.
//   mode 0: one env per wave (E=1), 51 lanes x 8 B per store            (round-1 kernel shape)
//   mode 1: E=64 envs per wave, 64 stores of 51 lanes x 8 B per step    (per-env loop)
//   mode 2: E=64 envs per wave, 26 stores of 64 lanes x 16 B per step   (flat 1 KiB stores)
//   mode 3: E=16 envs per wave, 6528 B = 6.4 stores of 64 lanes x 16 B
//   mode 4: plain streaming fill of the whole tensor, 16 B per lane (ceiling)
// build: hipcc -O3 --offload-arch=gfx950 -o wr_bench wr_bench.hip ; run: ./wr_bench [N] [T]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE, int WPB>
__global__ __launch_bounds__(WPB * 64) void k(double* out, int N, int T) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * WPB + (threadIdx.x >> 6);
    if (MODE == 0) {
        const int env = wave;
        if (env >= N) return;
        for (int t = 0; t < T; ++t) {
            double v = (double)(t + lane);
            if (lane < 51) out[((size_t)t * N + env) * 51 + lane] = v;
        }
    } else if (MODE == 1) {
        const int env0 = wave * 64;
        if (env0 >= N) return;
        for (int t = 0; t < T; ++t) {
            double* base = out + ((size_t)t * N + env0) * 51;
#pragma unroll 8
            for (int e = 0; e < 64; ++e) {
                double v = (double)(t + lane + e);
                if (lane < 51) base[e * 51 + lane] = v;
            }
        }
    } else if (MODE == 2 || MODE == 3) {
        constexpr int E = MODE == 2 ? 64 : 16;
        const int env0 = wave * E;
        if (env0 >= N) return;
        constexpr int NQ = E * 51 / 2;  // 16-byte quads per step
        for (int t = 0; t < T; ++t) {
            double2* base = (double2*)(out + ((size_t)t * N + env0) * 51);
            for (int q = lane; q < NQ; q += 64) {
                double2 v = make_double2((double)(t + q), (double)(t - q));
                base[q] = v;
            }
        }
    } else {
        const size_t total = (size_t)T * N * 51 / 2;
        double2* o = (double2*)out;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
            o[i] = make_double2((double)i, 1.0);
    }
}

template <int MODE, int WPB>
void run(const char* name, double* out, int N, int T, int envs_per_wave) {
    int waves = MODE == 4 ? 256 * 32 : (N + envs_per_wave - 1) / envs_per_wave;
    int blocks = (waves + WPB - 1) / WPB;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int it = 0; it < 4; ++it) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k<MODE, WPB>), dim3(blocks), dim3(WPB * 64), 0, 0, out, N, T);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (it > 0 && ms < best) best = ms;
    }
    double bytes = (double)T * N * 408.0;
    printf("%-44s WPB=%2d blocks=%6d  %.3f ms  %.2f TB/s\n", name, WPB, blocks, best, bytes / best / 1e9);
}

int main(int argc, char** argv) {
    int N = argc > 1 ? atoi(argv[1]) : 65536, T = argc > 2 ? atoi(argv[2]) : 600;
    double* out;
    CK(hipMalloc(&out, (size_t)T * N * 408));
    printf("N=%d T=%d bytes=%.2f GB\n", N, T, (double)T * N * 408 / 1e9);
    run<4, 4>("4: streaming fill 16B/lane", out, N, T, 1);
    run<0, 4>("0: wave/env, 51x8B", out, N, T, 1);
    run<0, 16>("0: wave/env, 51x8B", out, N, T, 1);
    run<1, 1>("1: 64 env/wave, 64 stores 51x8B", out, N, T, 64);
    run<1, 4>("1: 64 env/wave, 64 stores 51x8B", out, N, T, 64);
    run<2, 1>("2: 64 env/wave, flat 64x16B", out, N, T, 64);
    run<2, 4>("2: 64 env/wave, flat 64x16B", out, N, T, 64);
    run<3, 1>("3: 16 env/wave, flat 64x16B", out, N, T, 16);
    run<3, 4>("3: 16 env/wave, flat 64x16B", out, N, T, 16);
    return 0;
}
