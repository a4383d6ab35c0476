// The fake HIP runtime of tests/native/fakehip/hip/hip_runtime.h (test infrastructure; see the header).
#include <hip/hip_runtime.h>

#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <set>
#include <thread>

thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

struct fakehip_handle { int fd; size_t size; uint64_t phys; };
struct fakehip_event { double t_us; };
struct fakehip_stream {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    bool busy = false, stop = false;
    std::thread worker;
    bool started = false;
};

namespace {
std::mutex g_mu;
int g_ndev = 1;
thread_local int t_dev = 0;
thread_local hipError_t t_last = hipSuccess;
size_t g_total = (size_t)288 << 17;                  // 288 "GiB" at the tests' scale (1 GiB = 128 KiB)
std::map<uint64_t, size_t> g_phys;                   // simulated physical allocations: address -> size (first fit)
std::set<fakehip_handle*> g_handles;
std::map<char*, size_t> g_reservations;              // base -> size
struct Mapping { size_t size; fakehip_handle* h; };
std::map<char*, Mapping> g_mappings;                 // va -> mapping
std::set<void*> g_mallocs;
std::set<fakehip_stream*> g_streams;
std::set<fakehip_event*> g_events;
std::function<double(const fakehip::Launch&)> g_cost;
double g_clock_us = 0.0;                             // one simulated clock for all streams (kernels run inline, in program order)
std::atomic<uint64_t> g_launches{0};
long g_fail_at = -1, g_calls = 0;
bool g_fired = false;

hipError_t set(hipError_t e) { t_last = e; return e; }
// a call that a test may make fail
bool inject() {
    std::lock_guard<std::mutex> lk(g_mu);
    const long k = g_calls++;
    if (g_fail_at >= 0 && k == g_fail_at) { g_fired = true; return true; }
    return false;
}
size_t used_bytes() { size_t u = 0; for (auto& kv : g_phys) u += kv.second; return u; }
}  // namespace

const char* hipGetErrorString(hipError_t e) {
    switch (e) {
        case hipSuccess: return "no error";
        case hipErrorInvalidValue: return "invalid argument";
        case hipErrorOutOfMemory: return "out of memory";
        case hipErrorInvalidDevice: return "invalid device ordinal";
        case hipErrorNotReady: return "device not ready";
        default: return "unknown error";
    }
}
hipError_t hipGetLastError() { const hipError_t e = t_last; t_last = hipSuccess; return e; }
hipError_t hipGetDeviceCount(int* n) { *n = g_ndev; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = t_dev; return hipSuccess; }
hipError_t hipSetDevice(int d) { if (d < 0 || d >= g_ndev) return set(hipErrorInvalidDevice); t_dev = d; return hipSuccess; }
hipError_t hipDeviceSynchronize() {
    std::vector<fakehip_stream*> ss;
    { std::lock_guard<std::mutex> lk(g_mu); ss.assign(g_streams.begin(), g_streams.end()); }
    for (auto s : ss) (void)hipStreamSynchronize(s);
    return hipSuccess;
}
hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b) {
    if (inject()) return set(hipErrorUnknown);
    std::lock_guard<std::mutex> lk(g_mu);
    const size_t u = used_bytes();
    *total_b = g_total; *free_b = g_total > u ? g_total - u : 0;
    return hipSuccess;
}
hipError_t hipMalloc(void** p, size_t n) {
    if (inject()) return set(hipErrorOutOfMemory);
    *p = std::calloc(1, n ? n : 1);
    std::lock_guard<std::mutex> lk(g_mu);
    g_mallocs.insert(*p);
    return hipSuccess;
}
hipError_t hipFree(void* p) {
    if (!p) return hipSuccess;
    { std::lock_guard<std::mutex> lk(g_mu); if (!g_mallocs.erase(p)) return set(hipErrorInvalidValue); }
    std::free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t n, unsigned) {
    if (inject()) return set(hipErrorOutOfMemory);
    void* q = nullptr;
    if (posix_memalign(&q, 64, (n + 63) & ~(size_t)63) != 0) return set(hipErrorOutOfMemory);
    *p = q;
    std::lock_guard<std::mutex> lk(g_mu);
    g_mallocs.insert(q);
    return hipSuccess;
}
hipError_t hipHostFree(void* p) { return hipFree(p); }
hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { std::memset(p, v, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind, hipStream_t) {
    if (inject()) return set(hipErrorUnknown);
    std::memcpy(dst, src, n);
    return hipSuccess;
}

// ---- streams: a worker thread per stream that has asynchronous work -------------------------------------------------------------------
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) {
    if (inject()) return set(hipErrorOutOfMemory);
    *s = new fakehip_stream();
    std::lock_guard<std::mutex> lk(g_mu);
    g_streams.insert(*s);
    return hipSuccess;
}
static void stream_stop(fakehip_stream* s) {
    {
        std::lock_guard<std::mutex> lk(s->mu);
        s->stop = true;
    }
    s->cv.notify_all();
    if (s->started) s->worker.join();
}
hipError_t hipStreamDestroy(hipStream_t s) {
    if (!s) return set(hipErrorInvalidValue);
    { std::lock_guard<std::mutex> lk(g_mu); if (!g_streams.erase(s)) return set(hipErrorInvalidValue); }
    (void)hipStreamSynchronize(s);
    stream_stop(s);
    delete s;
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s) {
    if (!s) return hipSuccess;                                        // the null stream: everything on it ran inline
    std::unique_lock<std::mutex> lk(s->mu);
    s->cv.wait(lk, [&] { return s->q.empty() && !s->busy; });
    return hipSuccess;
}
hipError_t hipStreamQuery(hipStream_t s) {
    if (!s) return hipSuccess;
    std::lock_guard<std::mutex> lk(s->mu);
    return (s->q.empty() && !s->busy) ? hipSuccess : hipErrorNotReady;
}
void fakehip::enqueue(hipStream_t s, std::function<void()> fn) {
    std::lock_guard<std::mutex> lk(s->mu);
    s->q.push_back(std::move(fn));
    if (!s->started) {
        s->started = true;
        s->worker = std::thread([s] {
            for (;;) {
                std::function<void()> f;
                {
                    std::unique_lock<std::mutex> l2(s->mu);
                    s->cv.wait(l2, [&] { return s->stop || !s->q.empty(); });
                    if (s->q.empty()) return;
                    f = std::move(s->q.front());
                    s->q.pop_front();
                    s->busy = true;
                }
                f();
                {
                    std::lock_guard<std::mutex> l2(s->mu);
                    s->busy = false;
                }
                s->cv.notify_all();
            }
        });
    }
    s->cv.notify_all();
}

// ---- events on the simulated clock ---------------------------------------------------------------------------------------------------
hipError_t hipEventCreate(hipEvent_t* e) {
    if (inject()) return set(hipErrorOutOfMemory);
    *e = new fakehip_event{0.0};
    std::lock_guard<std::mutex> lk(g_mu);
    g_events.insert(*e);
    return hipSuccess;
}
hipError_t hipEventDestroy(hipEvent_t e) {
    if (!e) return set(hipErrorInvalidValue);
    { std::lock_guard<std::mutex> lk(g_mu); if (!g_events.erase(e)) return set(hipErrorInvalidValue); }
    delete e;
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { std::lock_guard<std::mutex> lk(g_mu); e->t_us = g_clock_us; return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return inject() ? set(hipErrorUnknown) : hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)((b->t_us - a->t_us) / 1000.0); return hipSuccess; }

// ---- the virtual-memory API on real virtual memory ------------------------------------------------------------------------------------
hipError_t hipMemGetAllocationGranularity(size_t* gran, const hipMemAllocationProp*, hipMemAllocationGranularity_flags) {
    if (inject()) return set(hipErrorUnknown);
    *gran = (size_t)sysconf(_SC_PAGESIZE);
    return hipSuccess;
}
hipError_t hipMemCreate(hipMemGenericAllocationHandle_t* h, size_t size, const hipMemAllocationProp*, unsigned long long) {
    if (inject()) return set(hipErrorOutOfMemory);
    std::lock_guard<std::mutex> lk(g_mu);
    // first fit in the simulated physical space
    uint64_t at = 0;
    bool found = false;
    for (auto& kv : g_phys) {
        if (kv.first >= at + size) { found = true; break; }
        at = kv.first + kv.second;
    }
    if (!found && at + size > g_total) return set(hipErrorOutOfMemory);
    const int fd = memfd_create("fakehip", 0);
    if (fd < 0 || ftruncate(fd, (off_t)size) != 0) { if (fd >= 0) close(fd); return set(hipErrorOutOfMemory); }
    g_phys[at] = size;
    *h = new fakehip_handle{fd, size, at};
    g_handles.insert(*h);
    return hipSuccess;
}
hipError_t hipMemRelease(hipMemGenericAllocationHandle_t h) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_handles.erase(h)) return set(hipErrorInvalidValue);
    g_phys.erase(h->phys);                                           // (mappings of the handle stay valid until they are unmapped: the memfd's pages live on)
    close(h->fd);
    h->fd = -1;
    bool mapped = false;
    for (auto& kv : g_mappings) mapped = mapped || kv.second.h == h;
    if (!mapped) delete h;                                           // else: deleted with its last mapping
    return hipSuccess;
}
hipError_t hipMemAddressReserve(void** va, size_t size, size_t, void*, unsigned long long) {
    if (inject()) return set(hipErrorOutOfMemory);
    void* p = mmap(nullptr, size, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (p == MAP_FAILED) return set(hipErrorOutOfMemory);
    *va = p;
    std::lock_guard<std::mutex> lk(g_mu);
    g_reservations[(char*)p] = size;
    return hipSuccess;
}
hipError_t hipMemAddressFree(void* va, size_t size) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_reservations.find((char*)va);
    if (it == g_reservations.end() || it->second != size) return set(hipErrorInvalidValue);
    g_reservations.erase(it);
    munmap(va, size);
    return hipSuccess;
}
static bool inside_reservation(char* va, size_t size) {
    auto it = g_reservations.upper_bound(va);
    if (it == g_reservations.begin()) return false;
    --it;
    return va >= it->first && va + size <= it->first + it->second;
}
hipError_t hipMemMap(void* va, size_t size, size_t offset, hipMemGenericAllocationHandle_t h, unsigned long long) {
    if (inject()) return set(hipErrorOutOfMemory);
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_handles.count(h) || offset != 0 || size != h->size || !inside_reservation((char*)va, size)) return set(hipErrorInvalidValue);
    auto nx = g_mappings.lower_bound((char*)va);
    if (nx != g_mappings.end() && nx->first < (char*)va + size) return set(hipErrorInvalidValue);     // overlaps the next mapping
    if (nx != g_mappings.begin()) { auto pv = std::prev(nx); if (pv->first + pv->second.size > (char*)va) return set(hipErrorInvalidValue); }
    void* p = mmap(va, size, PROT_NONE, MAP_SHARED | MAP_FIXED, h->fd, 0);
    if (p == MAP_FAILED) return set(hipErrorOutOfMemory);
    g_mappings[(char*)va] = Mapping{size, h};
    return hipSuccess;
}
hipError_t hipMemUnmap(void* va, size_t size) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_mappings.find((char*)va);
    if (it == g_mappings.end() || it->second.size != size) return set(hipErrorInvalidValue);
    fakehip_handle* h = it->second.h;
    g_mappings.erase(it);
    (void)mmap(va, size, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE | MAP_FIXED, -1, 0);   // the reservation again
    if (h->fd < 0) {                                                 // released already: this was its last mapping?
        bool mapped = false;
        for (auto& kv : g_mappings) mapped = mapped || kv.second.h == h;
        if (!mapped) delete h;
    }
    return hipSuccess;
}
hipError_t hipMemSetAccess(void* va, size_t size, const hipMemAccessDesc* desc, size_t) {
    if (inject()) return set(hipErrorUnknown);
    const int prot = desc->flags == hipMemAccessFlagsProtReadWrite ? (PROT_READ | PROT_WRITE) : (desc->flags == hipMemAccessFlagsProtRead ? PROT_READ : PROT_NONE);
    std::lock_guard<std::mutex> lk(g_mu);
    // every byte of the range must be mapped
    char* p = (char*)va;
    while (p < (char*)va + size) {
        auto it = g_mappings.find(p);
        if (it == g_mappings.end()) return set(hipErrorInvalidValue);
        p += it->second.size;
    }
    if (mprotect(va, size, prot) != 0) return set(hipErrorUnknown);
    return hipSuccess;
}

// ---- the tests' side --------------------------------------------------------------------------------------------------------------------
namespace fakehip {
void set_cost_model(std::function<double(const Launch&)> f) { std::lock_guard<std::mutex> lk(g_mu); g_cost = std::move(f); }
std::vector<uint64_t> phys_at(const void* va, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_mu);
    std::vector<uint64_t> out;
    for (auto it = g_mappings.lower_bound((char*)va); it != g_mappings.end() && it->first < (const char*)va + bytes; ++it) out.push_back(it->second.h->phys);
    return out;
}
void set_device_memory(size_t total) { std::lock_guard<std::mutex> lk(g_mu); g_total = total; }
void set_devices(int n) { std::lock_guard<std::mutex> lk(g_mu); g_ndev = n; }
void fail_nth(long k) { std::lock_guard<std::mutex> lk(g_mu); g_fail_at = k; g_calls = 0; g_fired = false; }
long fallible_calls() { std::lock_guard<std::mutex> lk(g_mu); return g_calls; }
bool failure_fired() { std::lock_guard<std::mutex> lk(g_mu); return g_fired; }
Counts counts() {
    std::lock_guard<std::mutex> lk(g_mu);
    Counts c{(long)g_handles.size(), (long)g_reservations.size(), (long)g_mappings.size(), (long)g_mallocs.size(), (long)g_streams.size(), (long)g_events.size(), 0, 0};
    for (auto& kv : g_reservations) c.reserved_bytes += kv.second;
    for (auto h : g_handles) c.handle_bytes += h->size;
    return c;
}
uint64_t launches() { return g_launches.load(); }
void launch(const Launch& l, hipStream_t, const std::function<void()>& body) {
    ++g_launches;
    double cost = 1.0;
    {
        std::function<double(const Launch&)> f;
        { std::lock_guard<std::mutex> lk(g_mu); f = g_cost; }
        if (f) cost = f(l);
    }
    gridDim = l.grid; blockDim = l.block;
    for (unsigned b = 0; b < l.grid.x; ++b)
        for (unsigned t = 0; t < l.block.x; ++t) {
            blockIdx = dim3(b); threadIdx = dim3(t);
            body();
        }
    std::lock_guard<std::mutex> lk(g_mu);
    g_clock_us += cost;
}
}  // namespace fakehip
