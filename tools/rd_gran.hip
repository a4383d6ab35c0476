// rd_gran.hip -- what does a scattered READ cost at the fabric, by layout?  (round 6: k_step3dq reads 340 B per env for a 7 x 7 int16
// window whose cells are 98 B; is that the floor of its 40-byte rows?)  R = 2.5 M records of 800 B (2 GB: far beyond the 256 MB Infinity
// Cache), every record read once per launch, 16 records per wave, one 16-byte piece per lane and load instruction (k_step3dq's shape):
//   span   the ten-row span of the int16 record, all 26 pieces (k_step3dq without its piece test)
//   win16  only the 16-byte pieces a 7 x 7 window at a pseudo-random (q, c) overlaps, int16 rows of 40 B (k_step3dq with its piece test)
//   tile   the same window on a record tiled 4 rows x 8 columns x int16 = 64-byte tiles (5 x 3 tiles = 960 B records)
//   byte   the same window on a byte plane with 20-byte rows (400 B of the record)
//   colb   the same window on column-blocked rows: [3 column blocks of 8][20 rows][16 B] (960 B records)
//   strideS  one 16-byte piece per lane at stride S (S = 16 .. 256): the granularity itself
// Run under rocprofv3 --pmc FETCH_SIZE (tools/rd_gran.sh); the kernel names carry the variant.  Prints useful bytes and time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int R = 2621440, RECB = 1024;          // records; bytes reserved per record (every layout fits)
__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
enum { SPAN, WIN16, TILE, BYTEP, COLB };
template <int MODE>
__global__ __launch_bounds__(256) void k_rd(const char* base, int recb, unsigned* sink, unsigned long long* useful) {
    const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int env0 = wave * 16;
    if (env0 >= R) return;
    unsigned acc = 0; unsigned long long nb = 0;
    constexpr int NPC = MODE == BYTEP ? 9 : 26;                      // candidate pieces per record
#pragma unroll
    for (int it = 0; it < (16 * NPC + 63) / 64; ++it) {
        const int P = it * 64 + lane, e = P / NPC, pp = P - NPC * e;
        if (P >= 16 * NPC) break;
        const int env = env0 + e;
        const unsigned h = hash((unsigned)env);
        const int q = (int)(h % 14u), c = (int)((h >> 8) % 14u);     // window rows q .. q + 6, columns c .. c + 6 (interior 20 x 20)
        bool need = false; int off = 0;
        if (MODE == SPAN) { const int rlo = min(q, 10); off = ((rlo * 40) & ~15) + pp * 16; need = off + 16 <= 800; }
        else if (MODE == WIN16) {
            const int rlo = min(q, 10); off = ((rlo * 40) & ~15) + pp * 16;
            const int f = off >> 1, q1 = f / 20, c1 = f - 20 * q1, e1 = min(c1 + 7, 19), e2 = c1 + 7 - 20;
            need = off + 16 <= 800 && ((q1 >= q && q1 <= q + 6 && c1 <= c + 6 && e1 >= c) || (e2 >= 0 && q1 + 1 >= q && q1 + 1 <= q + 6 && c <= e2));
        } else if (MODE == TILE) {                                   // piece pp: tile pp / 4 (of the <= 6 the window overlaps, listed 3 x 2), quarter pp % 4
            const int t = pp >> 2, quarter = pp & 3;
            const int tr0 = q >> 2, tr1 = (q + 6) >> 2, tc0 = c >> 3, tc1 = (c + 6) >> 3;
            const int tr = tr0 + t / 2, tc = tc0 + (t & 1);
            need = t < 6 && tr <= tr1 && tc <= tc1;
            // inside a 64-byte tile (4 rows of 8 cells): quarter = row; needed if the window covers that row
            const int row = tr * 4 + quarter;
            need = need && row >= q && row <= q + 6;
            off = (tr * 3 + tc) * 64 + quarter * 16;
        } else if (MODE == BYTEP) {                                  // rows of 20 B: the window's 7 rows = 140 contiguous bytes -> <= 9 + 1 pieces
            const int lo = (q * 20) & ~15; off = lo + pp * 16;
            const int b0 = q * 20 + c, b1 = (q + 6) * 20 + c + 6;
            need = off + 16 > b0 && off <= b1 && off + 16 <= 400;
            if (need) { const int r0 = off / 20, r1 = (off + 15) / 20; bool hit = false;
                for (int rr = r0; rr <= r1; ++rr) { const int s0 = max(off, rr * 20 + c), s1 = min(off + 15, rr * 20 + c + 6); hit |= rr >= q && rr <= q + 6 && s0 <= s1; }
                need = hit; }
        } else {                                                     // COLB: [block b][row][8 cells]: piece = (block, row)
            const int b = pp / 7 + (c >> 3), row = q + pp % 7;
            need = pp < 14 && b <= ((c + 6) >> 3);
            off = (b * 20 + row) * 16;
        }
        if (need) { const u32x4 v = __builtin_nontemporal_load((const u32x4*)(base + (size_t)env * recb + off)); acc ^= v.x ^ v.y ^ v.z ^ v.w; nb += 16; }
    }
    if (acc == 0x12345678u) sink[0] = acc;
    for (int o = 32; o; o >>= 1) nb += __shfl_xor(nb, o);
    if (lane == 0 && useful) atomicAdd(useful, nb);
}
template <int S>
__global__ __launch_bounds__(256) void k_stride(const char* base, size_t n, unsigned* sink) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u32x4 v = __builtin_nontemporal_load((const u32x4*)(base + i * S));
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) sink[0] = v.x;
}
// the write side: 2 bytes (a built cell) / 16 bytes (a header) per lane at stride S
template <int S, int BYTES>
__global__ __launch_bounds__(256) void k_wr(char* base, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (BYTES == 2) *(unsigned short*)(base + i * S + 38) = (unsigned short)i;
    else *(uint4*)(base + i * S) = make_uint4((unsigned)i, 1u, 2u, 3u);
}
__global__ void k_fill(uint4* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4((unsigned)i, (unsigned)(i >> 7), 3u, 4u);
}
template <typename F> float best_of(F launch) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); float best = 1e30f;
    for (int it = 0; it < 5; ++it) { CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (it >= 1 && ms < best) best = ms; }
    return best;
}
int main() {
    const size_t total = (size_t)R * RECB;
    char* slab; CK(hipMalloc((void**)&slab, total));
    unsigned* sink; CK(hipMalloc((void**)&sink, 16));
    unsigned long long* useful; CK(hipMalloc((void**)&useful, 8));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint4*)slab, total / 16);
    CK(hipDeviceSynchronize());
    const dim3 g((R / 16 + 3) / 4), b(256);
    auto run = [&](const char* name, auto kern, int recb) {
        CK(hipMemset(useful, 0, 8));
        hipLaunchKernelGGL(kern, g, b, 0, 0, (const char*)slab, recb, sink, useful);
        unsigned long long u = 0; CK(hipMemcpy(&u, useful, 8, hipMemcpyDeviceToHost));
        const float ms = best_of([&] { hipLaunchKernelGGL(kern, g, b, 0, 0, (const char*)slab, recb, sink, (unsigned long long*)nullptr); });
        printf("%-8s record %4d B: %6.1f B loaded per record, %7.3f ms per %d records = %5.2f ns per 1000 records\n", name, recb, (double)u / R, ms, R, ms * 1e6 / (R / 1000.0) / 1000.0);
    };
    run("span", k_rd<SPAN>, 800);
    run("win16", k_rd<WIN16>, 800);
    run("tile", k_rd<TILE>, 960);
    run("byte", k_rd<BYTEP>, 400);
    run("byte800", k_rd<BYTEP>, 800);
    run("colb", k_rd<COLB>, 960);
    const size_t n = (size_t)1 << 23;                                // 8 M lanes
    auto st = [&](const char* name, auto kern, int S) {
        if ((n - 1) * (size_t)S + 16 > total) { printf("%s: skipped (slab too small)\n", name); return; }
        const float ms = best_of([&] { hipLaunchKernelGGL(kern, dim3((unsigned)(n / 256)), dim3(256), 0, 0, (const char*)slab, n, sink); });
        printf("%-9s 16 B per lane at stride %3d: %7.3f ms per %zu lanes (%.0f MB useful)\n", name, S, ms, n, n * 16 / 1e6);
    };
    st("stride16", k_stride<16>, 16); st("stride32", k_stride<32>, 32); st("stride64", k_stride<64>, 64); st("stride128", k_stride<128>, 128); st("stride256", k_stride<256>, 256);
    auto wr = [&](const char* name, auto kern, int S, int B) {
        const size_t m = ((total / (size_t)S < n ? total / (size_t)S : n) / 256 - 1) * 256;   // every lane's bytes inside the slab
        if ((m - 1) * (size_t)S + 64 > total) { printf("%s: skipped (slab too small)\n", name); return; }
        const float ms = best_of([&] { hipLaunchKernelGGL(kern, dim3((unsigned)(m / 256)), dim3(256), 0, 0, slab, m); });
        printf("%-9s %2d B per lane written at stride %3d: %7.3f ms per %zu lanes (%.0f MB useful)\n", name, B, S, ms, m, m * (double)B / 1e6);
    };
    wr("wr2_800", k_wr<800, 2>, 800, 2); wr("wr2_128", k_wr<128, 2>, 128, 2); wr("wr16_16", k_wr<16, 16>, 16, 16); wr("wr16_64", k_wr<64, 16>, 64, 16); wr("wr16_128", k_wr<128, 16>, 128, 16);
    return 0;
}
