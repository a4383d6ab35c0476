// wr_occ.hip -- the fast fill of tools/wr_affinity.hip (1 MB chunk per workgroup, consecutive workgroups spread over 64 regions:
// 7.1 TB/s) with its occupancy cut down by a dynamic LDS allocation: does the write rate need many waves per CU?
// Also the same fill as PERSISTENT workgroups (grid = resident workgroups, each walks its XCD's regions chunk by chunk).
// build: hipcc -O3 --offload-arch=gfx950 -o wr_occ wr_occ.hip ; run: ./wr_occ [GiB]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef double v2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_fill(v2* out, size_t chunk_quads, size_t nch, int spread) {
    extern __shared__ int lds[];
    if (threadIdx.x == 100000) lds[0] = 1;
    const size_t b = blockIdx.x;
    const size_t per = nch / spread;
    size_t c = (b % spread) * per + b / spread;
    if (c >= nch) return;
    v2* o = out + c * chunk_quads;
    const v2 v = {1.5, 2.5};
    for (size_t i = threadIdx.x; i < chunk_quads; i += 256) o[i] = v;
}

// persistent: workgroup b (XCD b & 7) walks chunks b, b + G, b + 2G, ... of the spread order
__global__ __launch_bounds__(256) void k_persist(v2* out, size_t chunk_quads, size_t nch, int spread) {
    extern __shared__ int lds[];
    if (threadIdx.x == 100000) lds[0] = 1;
    const size_t per = nch / spread;
    const v2 v = {1.5, 2.5};
    for (size_t b = blockIdx.x; b < nch; b += gridDim.x) {
        size_t c = (b % spread) * per + b / spread;
        if (c >= nch) continue;
        v2* o = out + c * chunk_quads;
        for (size_t i = threadIdx.x; i < chunk_quads; i += 256) o[i] = v;
    }
}

template <typename F>
float best_of(F launch) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int it = 0; it < 6; ++it) {
        CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (it >= 2 && ms < best) best = ms;
    }
    return best;
}

int main(int argc, char** argv) {
    const size_t bytes = (size_t)((argc > 1 ? atof(argv[1]) : 16.0) * (double)(1ull << 30));
    v2* buf;
    CK(hipMalloc((void**)&buf, bytes));
    CK(hipFuncSetAttribute((const void*)k_fill, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)k_persist, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (size_t chunk : {(size_t)1 << 20, (size_t)32768}) {
        const size_t nch = bytes / chunk, cq = chunk / 16;
        printf("chunk %zu B, spread 64\n", chunk);
        for (int lds_kb : {0, 20, 40, 80, 160}) {                 // workgroups per CU: 8 (wave limit), 8, 4, 2, 1
            float ms = best_of([&] { hipLaunchKernelGGL(k_fill, dim3((unsigned)nch), dim3(256), (size_t)lds_kb * 1024, 0, buf, cq, nch, 64); });
            printf("  one-shot workgroups, %3d KB LDS each (<= %d per CU): %6.3f ms  %5.2f TB/s\n", lds_kb, lds_kb ? 160 / lds_kb : 8, ms, bytes / ms / 1e9);
        }
        for (int per_cu : {8, 4, 2, 1}) {
            const int lds_kb = per_cu == 8 ? 20 : 160 / per_cu;
            float ms = best_of([&] { hipLaunchKernelGGL(k_persist, dim3(256 * per_cu), dim3(256), (size_t)lds_kb * 1024, 0, buf, cq, nch, 64); });
            printf("  persistent, %d workgroups per CU (%d in all):              %6.3f ms  %5.2f TB/s\n", per_cu, 256 * per_cu, ms, bytes / ms / 1e9);
        }
    }
    return 0;
}
