// wr_blocks.hip -- what are the "fast regions" of tools/wr_scan.hip?  Its profile (profiles/r02e_wr_patterns.txt) is a triangle of
// period 32 GiB: a 16 GB streaming target is fastest when a 32 GiB boundary of the address map cuts it in half, slowest when it lies
// inside one 32 GiB block.  Hypothesis: concurrent write streams want to be spread over several 32 GiB blocks.  Test, in one slab:
//   1. find the phase: wave-major store-only pattern W over a 16 GB window at offsets 0, 2, ... 32 GiB -> the peak offset p,
//      boundary at p + window / 2;
//   2. W(B): the 1024 wave streams dealt round-robin over B blocks (each block holds a contiguous 16 GB / B piece, none of them
//      across a boundary), B = 1, 2, 3, 4, 6;
//   3. R(B): the rollout's rows (the reference layout [T][N][51] cut into B sub-tensors [T][N / B][51], one per block): tile w of tick t
//      goes to block w % B.
// build: hipcc -O3 --offload-arch=gfx950 -o wr_blocks wr_blocks.hip ; run: ./wr_blocks [slab GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int E = 64, D = 51, N = 65536, T = 600, WAVES = N / E;
constexpr size_t TILE = (size_t)E * D * 8;                 // 26 112 B: one wave's rows of one tick
constexpr size_t BLOCK = (size_t)32 << 30;

// wave w streams through its own T x TILE bytes; stream w lies in block w % B at piece offset (w / B) * T * TILE
__global__ __launch_bounds__(256) void k_w(char* slab, size_t first, int B, size_t stride) {
    const int lane = threadIdx.x & 63; const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= WAVES) return;
    char* p = slab + first + (size_t)(wave % B) * stride + (size_t)(wave / B) * T * TILE;
    for (int t = 0; t < T; ++t) { double* base = (double*)(p + (size_t)t * TILE);
#pragma unroll 8
        for (int e = 0; e < E; ++e) if (lane < D) base[e * D + lane] = (double)(t + lane + e); }
}
// rows: tile w of tick t -> block w % B, sub-tensor [T][N / B][51]: offset (t * (WAVES / B) + w / B) * TILE (B divides WAVES or the
// last sub-tensor is ragged: per-block stride is ceil(WAVES / B) tiles)
__global__ __launch_bounds__(256) void k_r(char* slab, size_t first, int B, size_t stride) {
    const int lane = threadIdx.x & 63; const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= WAVES) return;
    const size_t per = (WAVES + B - 1) / B;
    char* p = slab + first + (size_t)(wave % B) * stride + (size_t)(wave / B) * TILE;
    for (int t = 0; t < T; ++t) { double* base = (double*)(p + (size_t)t * per * TILE);
#pragma unroll 8
        for (int e = 0; e < E; ++e) if (lane < D) base[e * D + lane] = (double)(t + lane + e); }
}
// what a virtual-memory mapping could do for the ONE contiguous [T][N][51] tensor: its logical 2^cl-byte chunks dealt round-robin over B
// physical pieces `stride` apart (chunk c -> piece c % B, slot c / B); every lane computes the address of its own 8 bytes
__global__ __launch_bounds__(256) void k_c(char* slab, size_t first, int B, size_t stride, int cl) {
    const int lane = threadIdx.x & 63; const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= WAVES) return;
    const size_t mask = ((size_t)1 << cl) - 1;
    const int lb = B == 1 ? 0 : (B == 2 ? 1 : 2);            // B in {1, 2, 4}
    for (int t = 0; t < T; ++t) { const size_t L0 = ((size_t)t * WAVES + wave) * TILE + (size_t)lane * 8;
#pragma unroll 8
        for (int e = 0; e < E; ++e) if (lane < D) { const size_t L = L0 + (size_t)e * (D * 8); const size_t c = L >> cl;
            *(double*)(slab + first + (c & (size_t)(B - 1)) * stride + ((c >> lb) << cl) + (L & mask)) = (double)(t + lane + e); } }
}
// which address bit has to differ between neighbouring chunks?  The logical tensor (R: the reference order, W: wave-major) in 2 MB
// chunks, padded to 8192 chunks (16 GiB); physical chunk = logical chunk with index bits 0 and b swapped: logically adjacent chunks
// end up 2 MB << b apart, everything else stays.  b = 0: the contiguous tensor.
template <bool WAVE_MAJOR>
__global__ __launch_bounds__(256) void k_p(char* slab, size_t first, int b) {
    const int lane = threadIdx.x & 63; const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= WAVES) return;
    for (int t = 0; t < T; ++t) {
        const size_t L0 = (WAVE_MAJOR ? ((size_t)wave * T + t) : ((size_t)t * WAVES + wave)) * TILE + (size_t)lane * 8;
#pragma unroll 8
        for (int e = 0; e < E; ++e) if (lane < D) { const size_t L = L0 + (size_t)e * (D * 8); size_t c = L >> 21;
            const size_t x = ((c >> b) ^ c) & 1; c ^= x | (x << b);
            *(double*)(slab + first + (c << 21) + (L & (((size_t)1 << 21) - 1))) = (double)(t + lane + e); } }
}
template <typename F> float best_of(F launch) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); float best = 1e30f;
    for (int it = 0; it < 6; ++it) { CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (it >= 2 && ms < best) best = ms; }
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return best;
}
int main(int argc, char** argv) {
    const size_t bytes = (size_t)T * WAVES * TILE;
    const size_t gib = argc > 1 ? atoi(argv[1]) : 232;
    const size_t total = gib << 30;
    char* slab; CK(hipMalloc((void**)&slab, total));
    printf("slab of %zu GiB at %p; target %.2f GB\n", gib, (void*)slab, bytes / 1e9);
    // 1. phase
    double bestv = 0; size_t bestoff = 0;
    for (size_t off = 0; off <= ((size_t)34 << 30) && off + bytes <= total; off += (size_t)2 << 30) {
        float w = best_of([&] { hipLaunchKernelGGL(k_w, dim3(256), dim3(256), 0, 0, slab, off, 1, BLOCK); });
        float r = best_of([&] { hipLaunchKernelGGL(k_r, dim3(256), dim3(256), 0, 0, slab, off, 1, BLOCK); });
        const double v = bytes / w / 1e9;
        printf("offset %2zu GiB: W %5.2f  R %5.2f TB/s\n", off >> 30, v, bytes / r / 1e9);
        if (v > bestv) { bestv = v; bestoff = off; }
        fflush(stdout);
    }
    const size_t boundary = (bestoff + bytes / 2) % BLOCK;      // slab offset (mod 32 GiB) of the block boundaries
    printf("peak at offset %zu GiB -> boundaries at slab offset %.2f GiB + k * 32 GiB\n", bestoff >> 30, boundary / 1073741824.0);
    // 2./3. streams dealt over B whole blocks; a piece starts `margin` into its block and ends before the block does
    const size_t margin = (size_t)2 << 30;
    for (int B : {1, 2, 3, 4, 6}) {
        const size_t first = boundary + margin;
        const size_t per_w = (size_t)((WAVES + B - 1) / B) * T * TILE;
        if (first + (size_t)(B - 1) * BLOCK + per_w > total) { printf("B=%d does not fit the slab\n", B); continue; }
        float w = best_of([&] { hipLaunchKernelGGL(k_w, dim3(256), dim3(256), 0, 0, slab, first, B, BLOCK); });
        float r = best_of([&] { hipLaunchKernelGGL(k_r, dim3(256), dim3(256), 0, 0, slab, first, B, BLOCK); });
        printf("streams over %d block(s): W %5.2f  R %5.2f TB/s   (W %.3f ms, R %.3f ms)\n", B, bytes / w / 1e9, bytes / r / 1e9, w, r);
        fflush(stdout);
    }
    // the same with the pieces at the END of their blocks and in the middle (is it the block, or the distance between streams?)
    for (int B : {2, 4}) {
        const size_t per_w = (size_t)((WAVES + B - 1) / B) * T * TILE;
        const size_t first = boundary + BLOCK / 2 - per_w / 2;
        if (first + (size_t)(B - 1) * BLOCK + per_w > total) continue;
        float w = best_of([&] { hipLaunchKernelGGL(k_w, dim3(256), dim3(256), 0, 0, slab, first, B, BLOCK); });
        float r = best_of([&] { hipLaunchKernelGGL(k_r, dim3(256), dim3(256), 0, 0, slab, first, B, BLOCK); });
        printf("streams over %d block(s), pieces mid-block: W %5.2f  R %5.2f TB/s\n", B, bytes / w / 1e9, bytes / r / 1e9);
    }
    // B pieces INSIDE one block, 32 GiB / B apart... (is it the block boundary, or just the distance between the streams?)
    for (int B : {2, 4}) {
        const size_t stride = B == 2 ? (size_t)16 << 30 : (size_t)7 << 30;
        const size_t first = boundary + ((size_t)1 << 30);
        float w = best_of([&] { hipLaunchKernelGGL(k_w, dim3(256), dim3(256), 0, 0, slab, first, B, stride); });
        float r = best_of([&] { hipLaunchKernelGGL(k_r, dim3(256), dim3(256), 0, 0, slab, first, B, stride); });
        printf("%d pieces inside ONE block, %zu GiB apart: W %5.2f  R %5.2f TB/s\n", B, stride >> 30, bytes / w / 1e9, bytes / r / 1e9);
    }
    // the contiguous tensor with its chunks dealt over B pieces (k_c): piece distance and chunk size
    {
        const size_t first = boundary + ((size_t)1 << 30);
        float r0 = best_of([&] { hipLaunchKernelGGL(k_c, dim3(256), dim3(256), 0, 0, slab, first, 1, (size_t)0, 21); });
        printf("contiguous (per-lane addresses, 1 piece): %5.2f TB/s\n", bytes / r0 / 1e9);
        for (int B : {2, 4})
            for (size_t smb : {32, 128, 512, 2048, 8192, 16384, 24576}) {
                const size_t stride = smb << 20;
                if (first + (size_t)(B - 1) * stride + bytes / B + ((size_t)4 << 20) > total) continue;
                if (stride < bytes / B + ((size_t)4 << 20)) continue;          // the pieces would overlap
                float r = best_of([&] { hipLaunchKernelGGL(k_c, dim3(256), dim3(256), 0, 0, slab, first, B, stride, 21); });
                printf("2 MB chunks over %d pieces %5zu MiB apart: %5.2f TB/s\n", B, smb, bytes / r / 1e9);
                fflush(stdout);
            }
        for (int cl : {12, 16, 18, 21, 23, 25, 27, 30}) {
            float r = best_of([&] { hipLaunchKernelGGL(k_c, dim3(256), dim3(256), 0, 0, slab, first, 2, (size_t)16 << 30, cl); });
            printf("chunks of 2^%d B over 2 pieces 16 GiB apart: %5.2f TB/s\n", cl, bytes / r / 1e9);
            fflush(stdout);
        }
    }
    // neighbouring 2 MB chunks 2 MB << b apart (bit swap), at two slab offsets
    for (size_t off_gib : {1, 40}) {
        const size_t first = boundary + (off_gib << 30);
        if (first + ((size_t)16 << 30) > total) continue;
        for (int b = 0; b <= 12; ++b) {
            float r = best_of([&] { hipLaunchKernelGGL((k_p<false>), dim3(256), dim3(256), 0, 0, slab, first, b); });
            float w = best_of([&] { hipLaunchKernelGGL((k_p<true>), dim3(256), dim3(256), 0, 0, slab, first, b); });
            printf("slab offset %5.2f GiB, chunk bit 0 <-> bit %2d (neighbours %6zu MiB apart): R %5.2f  W %5.2f TB/s\n", first / 1073741824.0, b, (size_t)2 << b, bytes / r / 1e9, bytes / w / 1e9);
            fflush(stdout);
        }
    }
    CK(hipFree(slab));
    return 0;
}
