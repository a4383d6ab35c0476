// wr_shape3d.hip -- which store shape reaches the write ceiling for the 3D rollout's output stream on MI355X?
// Output [T][N][51] float64 with N = 16384, T = 1000 (BASELINE config 5): only N / 8 = 2048 waves exist when a wave owns
// a tile of 8 envs (the 3D height maps cap the tile at 8), so each wave-tick writes 8 x 408 B = 3264 contiguous bytes.
// Variants (all write every byte exactly once):
//   rows      8 row stores of 51 lanes x 8 B per wave-tick            (the rollout kernel's shape)
//   drain     rows + s_waitcnt vmcnt(0) every tick                    (what a global load in the loop forces)
//   sync      rows + __syncthreads() every tick                       (a block's 4 / 8 tiles advance in lockstep)
//   xcd       rows, blockIdx remapped so that an XCD owns a contiguous eighth of the env range
//   flat      the tile's 3264 B as 16 B per lane (3.2 stores)         (needs a transpose through LDS in the real kernel)
//   bflat     the BLOCK's WPB x 3264 B as 16 B per lane, waves interleaved by KiB, one barrier per tick
//   work      rows + a dependent VALU chain of `work` fma per tick before the stores (the stepping work of the real kernel);
//             workdrain: the same with vmcnt(0) after the chain (a load in the loop); worklds: the chain goes through LDS
//   rd        rows + the per-tick reward (8 x 4 B) and done (8 x 1 B) stores of the wave's 8 envs, [T][N] layouts
//   rdbatch   rows + the same small pieces, but written every 16 ticks (16 x 32 B + 16 x 8 B per wave)
//   rdblock   rows + reward / done staged in LDS per BLOCK for 16 ticks, then written as whole 128 B / 32 B runs by one wave
// build: hipcc -O3 --offload-arch=gfx950 -o wr_shape3d wr_shape3d.hip ; run: ./wr_shape3d [N] [T]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

enum { ROWS, DRAIN, SYNC, XCD, XCDSYNC, FLAT, BFLAT, WORK, WORKDRAIN, WORKLDS, RD, RDBATCH, RDBLOCK };

__device__ int g_work = 0;

__device__ float* g_rew;
__device__ unsigned char* g_done;

template <int MODE, int WPB, int E>
__global__ __launch_bounds__(WPB * 64) void k(double* out, int N, int T, int work, float* rew, unsigned char* done) {
    __shared__ float srew[16][WPB * E];
    __shared__ unsigned char sdone[16][WPB * E];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __shared__ float sh[WPB * 64 * 40];                        // 10 KB per wave, like the 3D height maps
    float acc = (float)lane;
    if (MODE == WORKLDS) for (int i = 0; i < 40; ++i) sh[threadIdx.x * 40 + i] = 0.f;
    int blk = blockIdx.x;
    if (MODE == XCD || MODE == XCDSYNC) { const int nb = gridDim.x; blk = (blk % 8) * (nb / 8) + blk / 8; }
    const int env0 = (blk * WPB + w) * E;
    if (env0 >= N) return;
    for (int t = 0; t < T; ++t) {
        if (MODE == FLAT) {
            double2* base = (double2*)(out + ((size_t)t * N + env0) * 51);
            for (int q = lane; q < E * 51 / 2; q += 64) base[q] = make_double2((double)(t + q), (double)(t - q));
        } else if (MODE == BFLAT) {
            double2* base = (double2*)(out + ((size_t)t * N + (size_t)blk * WPB * E) * 51);
            for (int q = w * 64 + lane; q < WPB * E * 51 / 2; q += WPB * 64) base[q] = make_double2((double)(t + q), (double)(t - q));
            __syncthreads();
        } else {
            double* base = out + ((size_t)t * N + env0) * 51;
            if (MODE == WORK || MODE == WORKDRAIN || MODE == WORKLDS) {
                for (int i = 0; i < work; ++i) {
                    acc = __builtin_fmaf(acc, 1.0000001f, 0.5f);
                    if (MODE == WORKLDS && (i & 15) == 0) { sh[threadIdx.x * 40 + (i >> 4) % 40] = acc; acc += sh[(threadIdx.x ^ 1) * 40 + (i >> 4) % 40]; }
                }
                if (MODE == WORKDRAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                double v = (double)(t + lane + e) + (double)acc;
                if (lane < 51) base[e * 51 + lane] = v;
            }
            if (MODE == DRAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (MODE == SYNC || MODE == XCDSYNC) __syncthreads();
            if (MODE == RD) {
                if (lane < E) { rew[(size_t)t * N + env0 + lane] = (float)(t + lane); done[(size_t)t * N + env0 + lane] = (unsigned char)(t & 1); }
            }
            if (MODE == RDBATCH || MODE == RDBLOCK) {
                if (lane < E) { srew[t & 15][w * E + lane] = (float)(t + lane); sdone[t & 15][w * E + lane] = (unsigned char)(t & 1); }
                if ((t & 15) == 15) {
                    const int t0 = t - 15;
                    if (MODE == RDBATCH) {                      // each wave flushes its own 16 x 8 values
                        for (int q = lane; q < 16 * E; q += 64) {
                            const int tt = q / E, e = q % E;
                            rew[(size_t)(t0 + tt) * N + env0 + e] = srew[tt][w * E + e];
                            done[(size_t)(t0 + tt) * N + env0 + e] = sdone[tt][w * E + e];
                        }
                    } else {
                        __syncthreads();
                        if (w == 0) {                           // one wave flushes the block's 16 x (WPB * E) values as whole runs
                            const int benv = blk * WPB * E;
                            for (int q = lane; q < 16 * WPB * E; q += 64) {
                                const int tt = q / (WPB * E), e = q % (WPB * E);
                                rew[(size_t)(t0 + tt) * N + benv + e] = srew[tt][e];
                            }
                            for (int q = lane; q < 16 * WPB * E / 4; q += 64) {
                                const int tt = q / (WPB * E / 4), e4 = q % (WPB * E / 4);
                                ((unsigned*)(done + (size_t)(t0 + tt) * N + benv))[e4] = ((const unsigned*)sdone[tt])[e4];
                            }
                        }
                        __syncthreads();
                    }
                }
            }
        }
    }
}

float* h_rew;
unsigned char* h_done;

template <int MODE, int WPB, int E>
void run(const char* name, double* out, int N, int T, int work = 0) {
    const int waves = N / E, blocks = waves / WPB;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int it = 0; it < 5; ++it) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k<MODE, WPB, E>), dim3(blocks), dim3(WPB * 64), 0, 0, out, N, T, work, h_rew, h_done);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (it > 0 && ms < best) best = ms;
    }
    double bytes = (double)T * N * 408.0;
    printf("%-58s E=%2d WPB=%2d blocks=%5d work=%4d  %.3f ms  %.2f TB/s\n", name, E, WPB, blocks, work, best, bytes / best / 1e9);
}

int main(int argc, char** argv) {
    int N = argc > 1 ? atoi(argv[1]) : 16384, T = argc > 2 ? atoi(argv[2]) : 1000;
    double* out;
    CK(hipMalloc(&out, (size_t)T * N * 408));
    CK(hipMemset(out, 0, (size_t)T * N * 408));
    printf("N=%d T=%d bytes=%.2f GB\n", N, T, (double)T * N * 408 / 1e9);
    CK(hipMalloc(&h_rew, (size_t)T * N * 4)); CK(hipMalloc(&h_done, (size_t)T * N));
    run<ROWS, 4, 8>("rows", out, N, T);
    run<RD, 4, 8>("rows + reward/done per tick", out, N, T);
    run<RDBATCH, 4, 8>("rows + reward/done every 16 ticks per wave", out, N, T);
    run<RDBLOCK, 4, 8>("rows + reward/done every 16 ticks per block, whole runs", out, N, T);
    run<RDBLOCK, 8, 8>("rows + reward/done every 16 ticks per block, whole runs", out, N, T);
    for (int work : {0, 100, 200, 300, 400, 600}) run<WORK, 4, 8>("rows after a VALU chain", out, N, T, work);
    for (int work : {100, 200, 300, 400}) run<WORKDRAIN, 4, 8>("rows after a VALU chain + vmcnt(0)", out, N, T, work);
    for (int work : {100, 200, 300, 400}) run<WORKLDS, 4, 8>("rows after a VALU + LDS chain", out, N, T, work);
    run<ROWS, 1, 8>("rows", out, N, T);
    run<ROWS, 8, 8>("rows", out, N, T);
    run<DRAIN, 4, 8>("rows + vmcnt(0) per tick", out, N, T);
    run<SYNC, 4, 8>("rows + barrier per tick", out, N, T);
    run<SYNC, 8, 8>("rows + barrier per tick", out, N, T);
    run<SYNC, 16, 8>("rows + barrier per tick", out, N, T);
    run<XCD, 4, 8>("rows, XCD owns a contiguous env range", out, N, T);
    run<XCDSYNC, 4, 8>("rows, XCD range + barrier", out, N, T);
    run<XCDSYNC, 8, 8>("rows, XCD range + barrier", out, N, T);
    run<FLAT, 4, 8>("flat 16 B/lane per tile", out, N, T);
    run<FLAT, 1, 8>("flat 16 B/lane per tile", out, N, T);
    run<BFLAT, 4, 8>("flat 16 B/lane per block + barrier", out, N, T);
    run<BFLAT, 8, 8>("flat 16 B/lane per block + barrier", out, N, T);
    run<ROWS, 4, 16>("rows", out, N, T);
    run<ROWS, 1, 16>("rows", out, N, T);
    run<SYNC, 4, 16>("rows + barrier per tick", out, N, T);
    run<FLAT, 4, 16>("flat 16 B/lane per tile", out, N, T);
    run<ROWS, 4, 4>("rows", out, N, T);
    run<SYNC, 8, 4>("rows + barrier per tick", out, N, T);
    run<BFLAT, 8, 4>("flat 16 B/lane per block + barrier", out, N, T);
    return 0;
}
