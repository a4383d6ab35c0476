// vmm_stale.hip -- tries to provoke what snac_traj_free works round (snac_hip.hip traj_release): in round 2 a block of the HIP
// virtual-memory API that was mapped at an address range recycled right after hipMemUnmap / hipMemAddressFree once read back
// zeros through a copy after a kernel had filled it.  Three ways of recycling, `iters` rounds each; every round a kernel fills
// the block with the round number, a copy (hipMemcpy to the host) and a second kernel read it back:
//   A  reserve -> create -> map -> fill -> check -> unmap -> release -> hipMemAddressFree; the next reserve asks for the SAME address
//   B  one range reserved once; every round maps FRESH handles there (unmap + release in between, no synchronisation beyond the copy)
//   C  as A, but with hipDeviceSynchronize before the unmap and other allocations (hipMalloc / hipFree) in between
//   D  as A, with the hipDeviceSynchronize only          E  as A, with the hipMalloc / hipFree only
// Prints one line per mode: rounds, rounds whose address was recycled, mismatching rounds (copy / kernel).
//   hipcc --offload-arch=gfx950 -O2 -o tools/vmm_stale tools/vmm_stale.hip && tools/vmm_stale [iters] [MB]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 2; } } while (0)

__global__ void fill(uint64_t* p, size_t n, uint64_t v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + i;
}
__global__ void check(const uint64_t* p, size_t n, uint64_t v, unsigned long long* bad) {
    unsigned long long k = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) k += p[n - 1 - i] != v + (n - 1 - i);
    if (k) atomicAdd(bad, k);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? std::atoi(argv[1]) : 200;
    const size_t bytes = (size_t)(argc > 2 ? std::atoi(argv[2]) : 64) << 20, chunk = (size_t)32 << 20, k = bytes / chunk, words = bytes / 8;
    hipMemAllocationProp prop;
    std::memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc;
    std::memset(&acc, 0, sizeof(acc));
    acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    unsigned long long* bad = nullptr;
    CK(hipMalloc((void**)&bad, 8));
    std::vector<uint64_t> host(words);
    for (int mode = 0; mode < 5; ++mode) {
        char* fixed = nullptr;
        if (mode == 1) CK(hipMemAddressReserve((void**)&fixed, bytes, chunk, nullptr, 0));
        char* last = nullptr;
        int recycled = 0, bad_copy = 0, bad_kernel = 0;
        for (int it = 0; it < iters; ++it) {
            char* va = fixed;
            if (mode != 1) CK(hipMemAddressReserve((void**)&va, bytes, chunk, last, 0));
            recycled += (va == last || mode == 1) && it > 0;
            std::vector<hipMemGenericAllocationHandle_t> hs(k);
            for (size_t j = 0; j < k; ++j) { CK(hipMemCreate(&hs[j], chunk, &prop, 0)); CK(hipMemMap(va + j * chunk, chunk, 0, hs[j], 0)); }
            CK(hipMemSetAccess(va, bytes, &acc, 1));
            const uint64_t v = ((uint64_t)(mode + 1) << 56) + ((uint64_t)(it + 1) << 32);
            CK(hipMemsetAsync(bad, 0, 8, nullptr));
            hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, nullptr, (uint64_t*)va, words, v);
            CK(hipMemcpy(host.data(), va, bytes, hipMemcpyDeviceToHost));
            size_t mism = 0;
            for (size_t i = 0; i < words; ++i) mism += host[i] != v + i;
            hipLaunchKernelGGL(check, dim3(1024), dim3(256), 0, nullptr, (const uint64_t*)va, words, v, bad);
            unsigned long long hb = 0;
            CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
            if (mism) { if (!bad_copy) std::printf("mode %c round %d: %zu of %zu words differ through the copy (first word %llx, wanted %llx)\n", 'A' + mode, it, mism, words, (unsigned long long)host[0], (unsigned long long)v); ++bad_copy; }
            if (hb) ++bad_kernel;
            if (mode == 2 || mode == 3) CK(hipDeviceSynchronize());
            if (mode == 2 || mode == 4) {
                void* spare = nullptr;
                CK(hipMalloc(&spare, (size_t)(3 + it % 5) << 20));
                CK(hipFree(spare));
            }
            for (size_t j = 0; j < k; ++j) { CK(hipMemUnmap(va + j * chunk, chunk)); CK(hipMemRelease(hs[j])); }
            if (mode != 1) { CK(hipMemAddressFree(va, bytes)); last = va; }
        }
        std::printf("mode %c: %d rounds of %zu MB, %d at a recycled address, mismatching rounds: %d through the copy, %d through a kernel\n",
                    'A' + mode, iters, bytes >> 20, recycled, bad_copy, bad_kernel);
        if (fixed) { CK(hipMemAddressFree(fixed, bytes)); }
    }
    return 0;
}
