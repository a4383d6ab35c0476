// Is fine-grained DEVICE memory writable by the host (large BAR) on this box?  A doorbell that the host stores into device memory
// and the GPU polls locally would take the GPU's read over the bus out of a mailbox step.  The host access runs in a forked child
// (a box without host-visible device memory answers with SIGSEGV).   hipcc --offload-arch=gfx950 -o bar_probe bar_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <sys/wait.h>
#include <unistd.h>

__global__ void echo(volatile unsigned long long* bell, volatile unsigned long long* ack, int rounds) {
    unsigned long long seen = 0;
    for (int i = 0; i < rounds; ++i) {
        unsigned long long v;
        long long t0 = wall_clock64();
        while ((v = __hip_atomic_load((unsigned long long*)bell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) == seen)
            if (wall_clock64() - t0 > 200000000LL) return;        // 2 s: give up
        seen = v;
        __hip_atomic_store((unsigned long long*)ack, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

static int run(int mode) {
    unsigned long long *bell = nullptr, *ack = nullptr;
    if (hipHostMalloc((void**)&ack, 4096, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) return 10;
    if (mode == 0) {
        if (hipHostMalloc((void**)&bell, 4096, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) return 11;
    } else {
        if (hipExtMallocWithFlags((void**)&bell, 4096, mode == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached) != hipSuccess) return 12;
    }
    *ack = 0;
    if (mode == 0) *bell = 0; else if (hipMemset(bell, 0, 8) != hipSuccess) return 13;
    hipDeviceSynchronize();
    const int rounds = 20000;
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipLaunchKernelGGL(echo, dim3(1), dim3(1), 0, s, bell, ack, rounds);
    auto t0 = std::chrono::steady_clock::now();
    for (unsigned long long i = 1; i <= (unsigned long long)rounds; ++i) {
        __atomic_store_n(bell, i, __ATOMIC_RELEASE);               // mode 1 / 2: a host store into device memory
        auto w0 = std::chrono::steady_clock::now();
        while (__atomic_load_n(ack, __ATOMIC_ACQUIRE) != i)
            if (std::chrono::steady_clock::now() - w0 > std::chrono::seconds(3)) { printf("mode %d: no echo at round %llu\n", mode, i); return 14; }
    }
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / rounds;
    hipStreamSynchronize(s);
    printf("mode %d (%s doorbell): %.2f us per round trip\n", mode, mode == 0 ? "host-memory" : (mode == 1 ? "fine-grained device-memory" : "uncached device-memory"), us);
    return 0;
}

int main() {
    for (int mode = 0; mode < 3; ++mode) {
        fflush(stdout);
        pid_t pid = fork();
        if (pid == 0) { int rc = run(mode); fflush(stdout); _exit(rc); }
        int st = 0;
        waitpid(pid, &st, 0);
        if (WIFSIGNALED(st)) printf("mode %d: child killed by signal %d (device memory is not host-accessible here)\n", mode, WTERMSIG(st));
        else if (WEXITSTATUS(st)) printf("mode %d: failed with code %d\n", mode, WEXITSTATUS(st));
    }
    return 0;
}
