// The C-level floor of a single-env step through the resident wavefront: a plain C++ host (no Python) calling snac_mailbox_step in a
// loop -- what of tools/mailbox_time.py's 5.2-6.0 us per raw step is ctypes, what is the bus and the wave.
//   hipcc -O2 -std=c++17 -Iinclude tools/mailbox_native.cpp -Lsnac_amd -lsnac_hip -Wl,-rpath,$PWD/snac_amd -o tools/mailbox_native
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#include "snac_hip.h"

#define OK(x) do { if ((x) != 0) { std::fprintf(stderr, "%s failed: %s\n", #x, snac_last_error()); return 1; } } while (0)
#define HOK(x) do { if ((x) != hipSuccess) { std::fprintf(stderr, "%s failed\n", #x); return 1; } } while (0)

int main() {
    snac_sizes sz;
    OK(snac_env_sizes(SNAC_ENV_2D, 0, &sz));
    const int P = 1;
    std::vector<uint32_t> plan(20, 0u);
    for (int q = 5; q < 15; ++q) plan[q] = 0x3FF00u;                  // a 10 x 10 square
    int16_t tb = 100;
    snac_env_desc d;
    std::memset(&d, 0, sizeof(d));
    d.kind = SNAC_ENV_2D; d.dynamic = 0; d.num_envs = 1; d.num_plans = P; d.obs_dtype = SNAC_OBS_F64; d.seed = 1; d.obs_tail = SNAC_TAIL_RECORD;
    snac_state st;
    int64_t* stats;
    HOK(hipMalloc((void**)&st.hdr, sizeof(snac_env_hdr))); HOK(hipMalloc((void**)&st.episode, 4)); HOK(hipMalloc(&st.grid, 80));
    HOK(hipMalloc((void**)&st.plans, 80)); HOK(hipMalloc((void**)&st.plan_tb, 2)); HOK(hipMalloc((void**)&stats, 24));
    HOK(hipMemset(st.hdr, 0, sizeof(snac_env_hdr))); HOK(hipMemset(st.episode, 0, 4)); HOK(hipMemset(st.grid, 0, 80)); HOK(hipMemset(stats, 0, 24));
    HOK(hipMemcpy((void*)st.plans, plan.data(), 80, hipMemcpyHostToDevice)); HOK(hipMemcpy((void*)st.plan_tb, &tb, 2, hipMemcpyHostToDevice));
    st.stat_episodes = stats; st.stat_return = stats + 1; st.stat_iou_fx = stats + 2;
    hipStream_t stream;
    HOK(hipStreamCreate(&stream));
    snac_mailbox* mb = nullptr;
    OK(snac_mailbox_create(&d, 0, &mb));
    OK(snac_reset_scalar(&d, &st, 0, snac_mailbox_row(mb), stream));
    OK(snac_stream_sync(stream));
    OK(snac_mailbox_touch(mb));
    for (int i = 0; i < 2000; ++i) OK(snac_mailbox_step(mb, &d, &st, i % 4, 1));
    for (int rep = 0; rep < 5; ++rep) {
        const int K = 20000;
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < K; ++i) OK(snac_mailbox_step(mb, &d, &st, i % 4, 1 + i % 3));
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / K;
        uint32_t stt[8];
        OK(snac_mailbox_stats(mb, stt));
        std::printf("native snac_mailbox_step: %.2f us per step   (wave: transition %.2f, row %.2f, fence %.2f us; launches %u)\n", us, stt[4] / 100.0, stt[5] / 100.0, stt[6] / 100.0, stt[0]);
    }
    OK(snac_mailbox_destroy(mb));
    return 0;
}
