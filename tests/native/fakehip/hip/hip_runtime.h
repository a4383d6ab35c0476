// A FAKE HIP runtime for host-side tests (tests/native/*_host_test.cpp): the host logic of snac_amd/csrc (trajectory-memory
// bookkeeping over the virtual-memory API, the mailbox protocol) compiled by gcc with -fsanitize=address,undefined / thread and driven
// without a GPU.  Test infrastructure only: nothing under snac_amd/ includes it.  What it models:
//   * memory      hipMalloc / hipHostMalloc are the heap; the virtual-memory API is real virtual memory: a handle is a memfd, a
//                 reservation an inaccessible anonymous mapping, hipMemMap maps the memfd over it, hipMemUnmap puts the reservation back
//   * kernels     hipLaunchKernelGGL runs the kernel on the calling thread, block by block, thread by thread (grid-stride loops work;
//                 barriers and LDS do not exist here), and charges the stream's simulated clock what the test's cost model says
//   * "physics"   every handle gets a simulated physical address from a first-fit allocator; the cost model of the test sees which
//                 handles a kernel's pointer arguments are mapped to (fakehip::handles_at) -- enough to model the slices of HBM
//   * streams     fakehip::enqueue(stream, fn) runs fn on the stream's worker thread, in order: hipStreamQuery / Synchronize see it
//   * failures    fakehip::fail_nth(k): the k-th call (counted from now) of any call that can fail returns an error
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <type_traits>
#include <vector>

// ---- language --------------------------------------------------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint4 { unsigned x, y, z, w; };
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
extern thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }

// ---- types ------------------------------------------------------------------------------------------------------------------------
typedef enum hipError_t {
    hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorInvalidDevice = 101, hipErrorNotReady = 600, hipErrorUnknown = 999
} hipError_t;
struct fakehip_stream;
struct fakehip_event;
struct fakehip_handle;
typedef fakehip_stream* hipStream_t;
typedef fakehip_event* hipEvent_t;
typedef fakehip_handle* hipMemGenericAllocationHandle_t;
enum { hipHostMallocPortable = 1, hipHostMallocMapped = 2, hipHostMallocCoherent = 0x40000000 };
enum { hipStreamNonBlocking = 1 };
typedef enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 } hipMemcpyKind;
typedef enum hipMemAllocationType { hipMemAllocationTypeInvalid = 0, hipMemAllocationTypePinned = 1 } hipMemAllocationType;
typedef enum hipMemLocationType { hipMemLocationTypeInvalid = 0, hipMemLocationTypeDevice = 1 } hipMemLocationType;
typedef enum hipMemAccessFlags { hipMemAccessFlagsProtNone = 0, hipMemAccessFlagsProtRead = 1, hipMemAccessFlagsProtReadWrite = 3 } hipMemAccessFlags;
typedef enum hipMemAllocationGranularity_flags { hipMemAllocationGranularityMinimum = 0, hipMemAllocationGranularityRecommended = 1 } hipMemAllocationGranularity_flags;
struct hipMemLocation { hipMemLocationType type; int id; };
struct hipMemAllocationProp { hipMemAllocationType type; int requestedHandleType; hipMemLocation location; void* win32HandleMetaData; struct { unsigned char c, g; unsigned short u; } allocFlags; };
struct hipMemAccessDesc { hipMemLocation location; hipMemAccessFlags flags; };

// ---- runtime ----------------------------------------------------------------------------------------------------------------------
const char* hipGetErrorString(hipError_t e);
hipError_t hipGetLastError();
hipError_t hipGetDeviceCount(int* n);
hipError_t hipGetDevice(int* d);
hipError_t hipSetDevice(int d);
hipError_t hipDeviceSynchronize();
hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b);
hipError_t hipMalloc(void** p, size_t n);
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags);
hipError_t hipHostFree(void* p);
hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t s);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind k, hipStream_t s);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamQuery(hipStream_t s);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
// the virtual-memory API
hipError_t hipMemGetAllocationGranularity(size_t* gran, const hipMemAllocationProp* prop, hipMemAllocationGranularity_flags f);
hipError_t hipMemCreate(hipMemGenericAllocationHandle_t* h, size_t size, const hipMemAllocationProp* prop, unsigned long long flags);
hipError_t hipMemRelease(hipMemGenericAllocationHandle_t h);
hipError_t hipMemAddressReserve(void** va, size_t size, size_t align, void* hint, unsigned long long flags);
hipError_t hipMemAddressFree(void* va, size_t size);
hipError_t hipMemMap(void* va, size_t size, size_t offset, hipMemGenericAllocationHandle_t h, unsigned long long flags);
hipError_t hipMemUnmap(void* va, size_t size);
hipError_t hipMemSetAccess(void* va, size_t size, const hipMemAccessDesc* desc, size_t count);

// ---- what the tests steer and read ---------------------------------------------------------------------------------------------------
namespace fakehip {
struct Launch {
    const char* name;                    // the kernel's name as written at the launch site
    dim3 grid, block;
    std::vector<uintptr_t> ptrs;         // its pointer arguments, in order
    std::vector<long long> ints;         // its integral arguments, in order
};
// simulated duration of a launch in microseconds (default: 1)
void set_cost_model(std::function<double(const Launch&)> f);
// the handles mapped in [va, va + bytes): simulated physical addresses, one per mapping, in address order
std::vector<uint64_t> phys_at(const void* va, size_t bytes);
void set_device_memory(size_t total_bytes);          // what hipMemGetInfo reports as total (default 288 "GiB" of the test's scale)
void set_devices(int n);
void fail_nth(long k);                               // the k-th fallible call from now fails (k = 0: the next one); < 0: never
long fallible_calls();                               // fallible calls seen since the last fail_nth()
bool failure_fired();
struct Counts { long handles, reservations, mappings, mallocs, streams, events; size_t reserved_bytes, handle_bytes; };
Counts counts();                                     // live objects: the leak check of the tests
void enqueue(hipStream_t s, std::function<void()> fn);   // asynchronous work on the stream's worker thread, in order
uint64_t launches();

template <typename T>
inline void pack_one(Launch& l, const T& v) {
    if constexpr (std::is_pointer<T>::value) l.ptrs.push_back((uintptr_t)v);
    else if constexpr (std::is_integral<T>::value || std::is_enum<T>::value) l.ints.push_back((long long)v);
}
template <typename... A>
inline Launch pack(const char* name, dim3 g, dim3 b, const A&... a) {
    Launch l{name, g, b, {}, {}};
    (pack_one(l, a), ...);
    return l;
}
void launch(const Launch& l, hipStream_t s, const std::function<void()>& body);
}  // namespace fakehip

#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
    ::fakehip::launch(::fakehip::pack(#kern, (grid), (block), __VA_ARGS__), (stream), [&]() { kern(__VA_ARGS__); })
