// wr_ceiling.hip -- is ~6 TB/s really the write ceiling of this GPU?  Plain write-only kernels over a 16 GB buffer:
//   A  grid-stride 16 B/lane (what a framework fill does)          B  the same with nontemporal stores
//   C  every workgroup owns one contiguous chunk (bytes / blocks)   D  C with nontemporal stores
//   E  hipMemsetAsync (the runtime's own fill)                      F  C with 8 B/lane
// build: hipcc -O3 --offload-arch=gfx950 -o wr_ceiling wr_ceiling.hip ; run: ./wr_ceiling [GB]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef double v2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(v2* out, size_t quads) {
    if (MODE == 0 || MODE == 1) {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += (size_t)gridDim.x * blockDim.x) {
            v2 v = {(double)i, 1.0};
            if (MODE == 1) __builtin_nontemporal_store(v, out + i); else out[i] = v;
        }
    } else if (MODE == 2 || MODE == 3) {
        const size_t per = quads / gridDim.x, b0 = per * blockIdx.x;
        for (size_t i = threadIdx.x; i < per; i += blockDim.x) {
            v2 v = {(double)i, 2.0};
            if (MODE == 3) __builtin_nontemporal_store(v, out + b0 + i); else out[b0 + i] = v;
        }
    } else {
        double* o = (double*)out;
        const size_t n = quads * 2, per = n / gridDim.x, b0 = per * blockIdx.x;
        for (size_t i = threadIdx.x; i < per; i += blockDim.x) o[b0 + i] = (double)i;
    }
}

template <int MODE>
void run(const char* name, v2* buf, size_t bytes, int blocks) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9f;
    for (int r = 0; r < 6; ++r) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, buf, bytes / 16);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (r && ms < best) best = ms;
    }
    printf("%-52s blocks=%6d  %7.3f ms  %5.2f TB/s\n", name, blocks, best, bytes / (best * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
    const size_t bytes = (size_t)(argc > 1 ? atof(argv[1]) : 16.0) * (1ull << 30);
    v2* buf;
    CK(hipMalloc((void**)&buf, bytes));
    CK(hipMemset(buf, 0, bytes));
    printf("write-only ceilings over %.1f GiB\n", bytes / (double)(1ull << 30));
    for (int blocks : {1024, 2048, 4096, 16384}) {
        run<0>("A grid-stride 16 B/lane", buf, bytes, blocks);
        run<1>("B grid-stride 16 B/lane, nontemporal", buf, bytes, blocks);
        run<2>("C one contiguous chunk per workgroup, 16 B/lane", buf, bytes, blocks);
        run<3>("D chunk per workgroup, nontemporal", buf, bytes, blocks);
        run<4>("F chunk per workgroup, 8 B/lane", buf, bytes, blocks);
    }
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(a)); CK(hipMemsetAsync(buf, r, bytes, 0)); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (r && ms < best) best = ms;
    }
    printf("%-52s                %7.3f ms  %5.2f TB/s\n", "E hipMemsetAsync", best, bytes / (best * 1e-3) / 1e12);
    return 0;
}
