// wr_affinity.hip -- what makes a framework fill (6.75 TB/s on the 16 GB observation tensor, wherever it lies) faster than every
// store-only kernel of tools/wr_ceiling.hip / wr_frontier.hip (5.7-6.1 TB/s, depending on the allocation)?
// One fill kernel, parameters:
//   CHUNK  contiguous bytes per workgroup (workgroups dispatched in address order)
//   PER    contiguous bytes per lane (16: one dwordx4 per lane per pass; 32 / 64: 2 / 4 adjacent dwordx4 per lane, as a
//          vectorised elementwise kernel writes)
//   SPREAD consecutive workgroups go to SPREAD far-apart regions (each region then filled chunk by chunk); the tensor is cleared
//          before and the bytes written are checked after; these variants flip between 5.7-6.2 and 6.9-7.2 TB/s from run to run
//   SHIFT  workgroup b writes chunk (b / 8) * 8 + ((b + SHIFT) & 7): the same chunks in the same order, but each chunk is written
//          by a workgroup that is SHIFT positions further round the 8 XCDs (workgroups are dealt to the XCDs round-robin) -- if
//          the rate depends on SHIFT, it depends on WHICH XCD writes a chunk (XCD <-> HBM-stack distance)
// build: hipcc -O3 --offload-arch=gfx950 -o wr_affinity wr_affinity.hip ; run: ./wr_affinity [GiB] [allocations]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef double v2 __attribute__((ext_vector_type(2)));

template <int PER>
__global__ __launch_bounds__(256) void k_fill(v2* out, size_t chunk_quads, int shift, size_t total_quads, int spread) {
    const size_t b = blockIdx.x;
    size_t c = (b & ~(size_t)7) + ((b + shift) & 7);
    if (spread > 1) { const size_t nch = gridDim.x, per = nch / spread; c = (b % spread) * per + b / spread; if (c >= nch) c = b; }   // consecutive workgroups `per` chunks apart
    v2* o = out + c * chunk_quads;
    constexpr int Q = PER / 16;                                 // quads per lane per pass
    const v2 v = {1.5, 2.5};
    for (size_t i = (size_t)threadIdx.x * Q; i < chunk_quads; i += (size_t)256 * Q) {
#pragma unroll
        for (int j = 0; j < Q; ++j)
            if (c * chunk_quads + i + j < total_quads) o[i + j] = v;
    }
}

__global__ void k_check(const v2* out, size_t quads, unsigned long long* bad) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += (size_t)gridDim.x * blockDim.x)
        if (out[i].x != 1.5 || out[i].y != 2.5) atomicAdd(bad, 1ull);
}

float time_it(void (*launch)(void*), void* ctx) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int it = 0; it < 6; ++it) {
        CK(hipEventRecord(a));
        launch(ctx);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (it >= 2 && ms < best) best = ms;
    }
    return best;
}

struct Ctx { v2* buf; size_t bytes; size_t chunk; int per; int shift; int spread; };

void launch(void* p) {
    Ctx* c = (Ctx*)p;
    const size_t quads = c->bytes / 16, cq = c->chunk / 16;
    size_t blocks = (quads + cq - 1) / cq;
    blocks = (blocks + 7) & ~(size_t)7;
    if (c->per == 16) hipLaunchKernelGGL((k_fill<16>), dim3((unsigned)blocks), dim3(256), 0, 0, c->buf, cq, c->shift, quads, c->spread);
    else if (c->per == 32) hipLaunchKernelGGL((k_fill<32>), dim3((unsigned)blocks), dim3(256), 0, 0, c->buf, cq, c->shift, quads, c->spread);
    else hipLaunchKernelGGL((k_fill<64>), dim3((unsigned)blocks), dim3(256), 0, 0, c->buf, cq, c->shift, quads, c->spread);
}

int main(int argc, char** argv) {
    const size_t bytes = (size_t)((argc > 1 ? atof(argv[1]) : 14.9414) * (double)(1ull << 30)) & ~(size_t)0xFFFFF;
    const int A = argc > 2 ? atoi(argv[2]) : 3;
    printf("fill of %.2f GB; TB/s\n", bytes / 1e9);
    for (int a = 0; a < A; ++a) {
        v2* buf;
        CK(hipMalloc((void**)&buf, bytes + (64 << 20)));
        printf("allocation %d (%p)\n", a, (void*)buf);
        for (size_t chunk : {(size_t)4096, (size_t)8192, (size_t)16384, (size_t)32768, (size_t)65536, (size_t)262144, (size_t)1048576}) {
            printf("  chunk %8zu B:", chunk);
            for (int per : {16, 32, 64}) {
                Ctx c{buf, bytes, chunk, per, 0, 0};
                printf("   %d B/lane %5.2f", per, bytes / time_it(launch, &c) / 1e9);
            }
            printf("   | 32 B/lane, XCD shift 1..7:");
            for (int s = 1; s < 8; ++s) {
                Ctx c{buf, bytes, chunk, 32, s, 0};
                printf(" %5.2f", bytes / time_it(launch, &c) / 1e9);
            }
            printf("   | 16 B/lane, consecutive workgroups spread over 64 / 2048 / 65536 regions:");
            for (int sp : {64, 2048, 65536}) {
                Ctx c{buf, bytes, chunk, 16, 0, sp};
                CK(hipMemset(buf, 0, bytes));
                const float ms = time_it(launch, &c);
                static unsigned long long* bad = nullptr;
                if (!bad) CK(hipMalloc(&bad, 8));
                CK(hipMemset(bad, 0, 8));
                hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, buf, bytes / 16, bad);
                unsigned long long h; CK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
                printf(" %5.2f (%.1f%% unwritten)", bytes / ms / 1e9, 100.0 * h / (bytes / 16));
            }
            printf("\n");
            fflush(stdout);
        }
        hipEvent_t x, y;
        CK(hipEventCreate(&x)); CK(hipEventCreate(&y));
        float best = 1e30f;
        for (int r = 0; r < 5; ++r) {
            CK(hipEventRecord(x)); CK(hipMemsetAsync(buf, r, bytes, 0)); CK(hipEventRecord(y)); CK(hipEventSynchronize(y));
            float ms; CK(hipEventElapsedTime(&ms, x, y));
            if (r && ms < best) best = ms;
        }
        printf("  hipMemsetAsync %5.2f\n", bytes / best / 1e9);
    }
    return 0;
}
