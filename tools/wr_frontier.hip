// wr_frontier.hip -- why does the rollout's observation stream top out at 5.4-5.95 TB/s (and depend on where the tensor lies)
// when torch's fill of the same 16 GB tensor runs at 6.75 TB/s wherever it lies?  Store-only kernels over [T][N][51] float64:
//   A  linear, fine-grained: one workgroup per (tick, 4 tiles), dispatched in address order (what a framework fill does)
//   B  the rollout's pattern: a wave owns 64 envs and loops over the ticks, 64 row stores of 51 x 8 B per tick
//   C  the same loop with flat stores (64 lanes x 16 B, whole aligned lines)
//   D  B with a bounded lead: a wave may run at most LEAD ticks ahead of the average tick of the waves that have started
//      (two global counters, updated every 8 ticks; the slowest started wave is never held, so it cannot deadlock)
//   E  C with the same bounded lead
// Every variant runs on several separately allocated tensors (the rollout's time depends on the allocation).
// build: hipcc -O3 --offload-arch=gfx950 -o wr_frontier wr_frontier.hip ; run: ./wr_frontier [N] [T] [allocations]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int E = 64, D = 51;

__global__ __launch_bounds__(256) void k_linear(double* out, size_t quads) {   // A
    double2* o = (double2*)out;
    const size_t per = (size_t)4 * E * D / 2;                   // one block: 4 tiles of one tick = 104 448 B
    const size_t b0 = per * blockIdx.x;
    for (size_t i = threadIdx.x; i < per && b0 + i < quads; i += 256) o[b0 + i] = make_double2((double)i, 1.0);
}

template <bool FLAT, int LEAD>
__global__ __launch_bounds__(256) void k_loop(double* out, int N, int T, unsigned long long* ctr) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int env0 = wave * E;
    if (env0 >= N) return;
    if (LEAD > 0 && lane == 0) atomicAdd(&ctr[0], 1ull);        // started waves
    for (int t = 0; t < T; ++t) {
        if (LEAD > 0 && (t & 7) == 0) {
            if (lane == 0 && t > 0) atomicAdd(&ctr[1], 8ull);     // ticks finished by all started waves
            // hold while this wave is more than LEAD ticks ahead of the average started wave
            for (int spin = 0; spin < 100000; ++spin) {
                const unsigned long long started = __hip_atomic_load(&ctr[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long ticks = __hip_atomic_load(&ctr[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned long long)t * started <= ticks + (unsigned long long)LEAD * started) break;
                __builtin_amdgcn_s_sleep(8);
            }
        }
        if (FLAT) {
            double2* base = (double2*)(out + ((size_t)t * N + env0) * D);
            for (int q = lane; q < E * D / 2; q += 64) base[q] = make_double2((double)(t + q), (double)(t - q));
        } else {
            double* base = out + ((size_t)t * N + env0) * D;
#pragma unroll 8
            for (int e = 0; e < E; ++e)
                if (lane < D) base[e * D + lane] = (double)(t + lane + e);
        }
    }
    if (LEAD > 0 && lane == 0) atomicAdd(&ctr[1], (unsigned long long)(T - ((T - 1) & ~7)));   // the last partial group
}

// F  strided ownership: a wave owns 64 / G groups of G consecutive envs (G x 408 B = whole 128-byte lines for G = 16, 32, 64); group
//    g of wave w starts at env g * (waves * G) + w * G, so that at any moment the waves together write ONE contiguous region of
//    waves x G x 408 B (6.7 MB for G = 16) that sweeps through the tick's 26.7 MB -- the write frontier of a fine-grained fill
template <int G>
__global__ __launch_bounds__(256) void k_strided(double* out, int N, int T) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int waves = N / E;
    if (wave >= waves) return;
    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int g = 0; g < E / G; ++g) {
            double* base = out + ((size_t)t * N + (size_t)g * waves * G + (size_t)wave * G) * D;
#pragma unroll 8
            for (int e = 0; e < G; ++e)
                if (lane < D) base[e * D + lane] = (double)(t + lane + e);
        }
    }
}

// G  B with a tick skew: wave w writes tick (t + (w % S) * K) mod T at its iteration t -- the address pattern of waves that were
//    started K ticks apart in S groups (no state here, so the skew costs no idle time): at any moment the waves write in S
//    regions K x 26.7 MB apart instead of one.  SKEWBLK: the skew is taken from the workgroup index (the 4 waves of a block together)
template <bool FLAT, bool SKEWBLK>
__global__ __launch_bounds__(256) void k_skew(double* out, int N, int T, int S, int K) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int env0 = wave * E;
    if (env0 >= N) return;
    const int skew = ((SKEWBLK ? (int)blockIdx.x : wave) % S) * K;
    for (int t0 = 0; t0 < T; ++t0) {
        const int t = (t0 + skew) % T;
        if (FLAT) {
            double2* base = (double2*)(out + ((size_t)t * N + env0) * D);
            for (int q = lane; q < E * D / 2; q += 64) base[q] = make_double2((double)(t + q), (double)(t - q));
        } else {
            double* base = out + ((size_t)t * N + env0) * D;
#pragma unroll 8
            for (int e = 0; e < E; ++e)
                if (lane < D) base[e * D + lane] = (double)(t + lane + e);
        }
    }
}

// H  B with the workgroups renumbered so that an XCD (workgroups are dealt round-robin to the 8 XCDs) owns a contiguous range of
//    envs: XCD x gets blocks [x * chunk, (x + 1) * chunk).  PIECES > 1: the XCD's range is cut into PIECES contiguous pieces
//    interleaved with the other XCDs' (piece size = chunk / PIECES blocks) -- how long must an XCD-exclusive address run be?
template <bool FLAT>
__global__ __launch_bounds__(256) void k_xcd(double* out, int N, int T, int pieces) {
    const int lane = threadIdx.x & 63;
    const int nb = (int)gridDim.x, chunk = nb >> 3;
    const int x = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;             // XCD, position inside the XCD's share
    const int psz = chunk / pieces;                                       // blocks per piece
    const int blk = ((j / psz) * 8 + x) * psz + (j % psz);
    const int wave = blk * 4 + (threadIdx.x >> 6);
    const int env0 = wave * E;
    if (env0 >= N) return;
    for (int t = 0; t < T; ++t) {
        if (FLAT) {
            double2* base = (double2*)(out + ((size_t)t * N + env0) * D);
            for (int q = lane; q < E * D / 2; q += 64) base[q] = make_double2((double)(t + q), (double)(t - q));
        } else {
            double* base = out + ((size_t)t * N + env0) * D;
#pragma unroll 8
            for (int e = 0; e < E; ++e)
                if (lane < D) base[e * D + lane] = (double)(t + lane + e);
        }
    }
}

// I  whole 4 KB granules: a workgroup of 8 waves owns 512 consecutive envs = 208 896 B per tick = exactly 51 granules of 4 KB
//    (408 B rows: lcm(408, 4096) = 512 rows), and writes them in address order, pass p = 8 KB: wave w the 1 KB piece
//    [p * 8192 + w * 1024, + 1024).  Every 4 KB-aligned granule is written whole, by one workgroup, at one time.  BAR: a
//    __syncthreads() per tick (what a real kernel would need between its transition and the block-wide store phase).
template <bool BAR>
__global__ __launch_bounds__(512) void k_granule(double* out, int N, int T) {
    const int env0 = blockIdx.x * 512;
    if (env0 >= N) return;
    const int tid = threadIdx.x;
    for (int t = 0; t < T; ++t) {
        double2* base = (double2*)(out + ((size_t)t * N + env0) * D);
        for (int q = tid; q < 512 * D / 2; q += 512) base[q] = make_double2((double)(t + q), (double)(t - q));
        if (BAR) __syncthreads();
    }
}

// M  B with each wave starting its 64 rows at a different row (rotation by (wave * ROT) % 64, wrapping inside the tile): at any
//    moment the waves then write at different offsets inside their tiles instead of all at row j (addresses w * 51 * 512 + j * 408:
//    the same low 9 address bits in every wave)
__global__ __launch_bounds__(256) void k_rot(double* out, int N, int T, int rot) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int env0 = wave * E;
    if (env0 >= N) return;
    const int r0 = (wave * rot) & 63;
    for (int t = 0; t < T; ++t) {
        double* base = out + ((size_t)t * N + env0) * D;
#pragma unroll 8
        for (int e = 0; e < E; ++e) {
            const int row = (e + r0) & 63;
            if (lane < D) base[row * D + lane] = (double)(t + lane + e);
        }
    }
}

// W  B's stores into a WAVE-MAJOR tensor [N / 64][T][64][51] (not the reference's layout): every wave streams through its own
//    15.7 MB, a 2 MB page serves 80 of its ticks -- if this is much faster than B, B pays for touching a new page per wave and tick
__global__ __launch_bounds__(256) void k_wavemajor(double* out, int N, int T) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave * E >= N) return;
    for (int t = 0; t < T; ++t) {
        double* base = out + ((size_t)wave * T + t) * E * D;
#pragma unroll 8
        for (int e = 0; e < E; ++e)
            if (lane < D) base[e * D + lane] = (double)(t + lane + e);
    }
}

template <typename F>
float best_of(F launch, int reps = 5) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int it = 0; it < reps + 2; ++it) {
        CK(hipEventRecord(a));
        launch();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (it >= 2 && ms < best) best = ms;
    }
    return best;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 65536, T = argc > 2 ? atoi(argv[2]) : 600, A = argc > 3 ? atoi(argv[3]) : 6;
    const size_t bytes = (size_t)T * N * D * 8;
    unsigned long long* ctr;
    CK(hipMalloc(&ctr, 16));
    printf("N=%d T=%d  %.2f GB per tensor, %d allocations; ms (TB/s)\n", N, T, bytes / 1e9, A);
    printf("%-6s %-18s %-18s %-18s %-18s %-18s %-18s\n", "alloc", "A linear fine", "B loop rows", "C loop flat", "D rows lead 8", "D rows lead 2", "E flat lead 2");
    double* bufs[32];
    const int waves = (N + E - 1) / E, blocks = (waves + 3) / 4;
    for (int a = 0; a < A && a < 32; ++a) {
        CK(hipMalloc(&bufs[a], bytes));
        double* out = bufs[a];
        const size_t quads = bytes / 16;
        const int lin_blocks = (int)((quads + (size_t)4 * E * D / 2 - 1) / ((size_t)4 * E * D / 2));
        float r[6];
        r[0] = best_of([&] { hipLaunchKernelGGL(k_linear, dim3(lin_blocks), dim3(256), 0, 0, out, quads); });
        r[1] = best_of([&] { hipLaunchKernelGGL((k_loop<false, 0>), dim3(blocks), dim3(256), 0, 0, out, N, T, ctr); });
        r[2] = best_of([&] { hipLaunchKernelGGL((k_loop<true, 0>), dim3(blocks), dim3(256), 0, 0, out, N, T, ctr); });
        r[3] = r[4] = r[5] = 0.f;   // D / E (bounded lead through two global counters): 2.5 TB/s, the atomics serialise -- not run any more
        float f16 = best_of([&] { hipLaunchKernelGGL((k_strided<16>), dim3(blocks), dim3(256), 0, 0, out, N, T); });
        float f32 = best_of([&] { hipLaunchKernelGGL((k_strided<32>), dim3(blocks), dim3(256), 0, 0, out, N, T); });
        float f8 = best_of([&] { hipLaunchKernelGGL((k_strided<8>), dim3(blocks), dim3(256), 0, 0, out, N, T); });
        {
            float g = best_of([&] { hipLaunchKernelGGL(k_wavemajor, dim3(blocks), dim3(256), 0, 0, out, N, T); });
            printf("%-6d W wave-major layout [N/64][T][64][51]: %5.3f (%4.2f)\n", a, g, bytes / g / 1e9);
        }
        printf("%-6d M rows rotated per wave:", a);
        for (int R : {0, 1, 7, 13, 16, 21, 32}) {
            float g = best_of([&] { hipLaunchKernelGGL(k_rot, dim3(blocks), dim3(256), 0, 0, out, N, T, R); });
            printf(" rot=%d %5.3f (%4.2f)", R, g, bytes / g / 1e9);
        }
        printf("\n");
        {
            const int gb = (N + 511) / 512;
            float g0 = best_of([&] { hipLaunchKernelGGL((k_granule<false>), dim3(gb), dim3(512), 0, 0, out, N, T); });
            float g1 = best_of([&] { hipLaunchKernelGGL((k_granule<true>), dim3(gb), dim3(512), 0, 0, out, N, T); });
            printf("%-6d I whole granules (512-env workgroups, %d blocks): %5.3f (%4.2f)   with a barrier per tick %5.3f (%4.2f)\n", a, gb, g0, bytes / g0 / 1e9, g1, bytes / g1 / 1e9);
        }
        printf("%-6d H XCD-contiguous (rows), pieces per XCD:", a);
        for (int P : {1, 2, 4, 8, 16, 32}) {
            float g = best_of([&] { hipLaunchKernelGGL((k_xcd<false>), dim3(blocks), dim3(256), 0, 0, out, N, T, P); });
            printf(" P=%d %5.3f (%4.2f)", P, g, bytes / g / 1e9);
        }
        {
            float g = best_of([&] { hipLaunchKernelGGL((k_xcd<true>), dim3(blocks), dim3(256), 0, 0, out, N, T, 1); });
            printf("   flat P=1 %5.3f (%4.2f)", g, bytes / g / 1e9);
        }
        printf("\n");
        printf("%-6d G skew (rows, per wave) K=1:", a);
        for (int S : {1, 2, 4, 8, 16, 32, 64}) {
            float g = best_of([&] { hipLaunchKernelGGL((k_skew<false, false>), dim3(blocks), dim3(256), 0, 0, out, N, T, S, 1); });
            printf(" S=%d %5.3f (%4.2f)", S, g, bytes / g / 1e9);
        }
        printf("\n       per block K=1:");
        for (int S : {4, 8, 16, 64}) {
            float g = best_of([&] { hipLaunchKernelGGL((k_skew<false, true>), dim3(blocks), dim3(256), 0, 0, out, N, T, S, 1); });
            printf(" S=%d %5.3f (%4.2f)", S, g, bytes / g / 1e9);
        }
        printf("   per wave K=4:");
        for (int S : {4, 8, 16}) {
            float g = best_of([&] { hipLaunchKernelGGL((k_skew<false, false>), dim3(blocks), dim3(256), 0, 0, out, N, T, S, 4); });
            printf(" S=%d %5.3f (%4.2f)", S, g, bytes / g / 1e9);
        }
        printf("   flat per wave K=1:");
        for (int S : {8, 64}) {
            float g = best_of([&] { hipLaunchKernelGGL((k_skew<true, false>), dim3(blocks), dim3(256), 0, 0, out, N, T, S, 1); });
            printf(" S=%d %5.3f (%4.2f)", S, g, bytes / g / 1e9);
        }
        printf("\n");
        printf("%-6d F strided G=8 %6.3f (%4.2f)  G=16 %6.3f (%4.2f)  G=32 %6.3f (%4.2f) |", a, f8, bytes / f8 / 1e9, f16, bytes / f16 / 1e9, f32, bytes / f32 / 1e9);
        for (int i = 0; i < 3; ++i) printf(" %6.3f (%4.2f)     ", r[i], bytes / r[i] / 1e9);
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
