// host_parity.cpp -- a plain C++ host (no Python, no torch) driving libsnac_hip.so through include/snac_hip.h and
// checking it against the C oracle.  Test infrastructure: built and run by tests/test_gpu_native_host.py.
//   hipcc -O2 -I include -I oracle tests/native/host_parity.cpp -L snac_amd -lsnac_hip -L oracle -lsnac_oracle -o host_parity
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "snac_hip.h"
#include "snac_oracle.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
#define SNAC_CHECK(x) do { int rc_ = (x); if (rc_ != SNAC_OK) { std::printf("snac error %d: %s (line %d)\n", rc_, snac_last_error(), __LINE__); return 3; } } while (0)

int main() {
    const int N = 777, T = 200, P = 2;            // ragged tile tail on purpose
    const uint64_t seed = 31337;
    // two static 2D plans (dense / sparse circle) from the oracle's tables, as full 26x26 grids
    std::vector<int32_t> full(P * 676);
    for (int p = 0; p < P; ++p) if (orc_static_plan(2, p, full.data() + p * 676) != 676) return 1;
    // device layout of include/snac_hip.h: 20 row words per plan, bit j = interior column j; total_brick floored at 30
    std::vector<uint32_t> packed(P * 20, 0u);
    std::vector<int16_t> tb(P);
    for (int p = 0; p < P; ++p) {
        int area = 0;
        for (int r = 0; r < 20; ++r)
            for (int c = 0; c < 20; ++c)
                if (full[p * 676 + (r + 3) * 26 + (c + 3)]) { packed[p * 20 + r] |= 1u << c; ++area; }
        tb[p] = (int16_t)(area < 30 ? 30 : area);
    }
    snac_sizes sz;
    SNAC_CHECK(snac_env_sizes(SNAC_ENV_2D, 1, &sz));
    snac_env_desc d;
    std::memset(&d, 0, sizeof(d));
    d.kind = SNAC_ENV_2D; d.dynamic = 1; d.num_envs = N; d.num_plans = P; d.obs_dtype = SNAC_OBS_F64; d.seed = seed; d.env_id_base = 4242;
    snac_state st;
    int64_t* stats;
    HIP_OK(hipMalloc((void**)&st.hdr, N * sizeof(snac_env_hdr)));
    HIP_OK(hipMalloc((void**)&st.episode, N * sizeof(int32_t)));
    HIP_OK(hipMalloc(&st.grid, (size_t)N * sz.grid_elems * sz.grid_elem_bytes));
    HIP_OK(hipMalloc((void**)&st.plans, packed.size() * 4));
    HIP_OK(hipMalloc((void**)&st.plan_tb, P * 2));
    HIP_OK(hipMalloc((void**)&stats, 3 * N * sizeof(int64_t)));
    st.stat_episodes = stats; st.stat_return = stats + N; st.stat_iou_fx = stats + 2 * N;
    std::vector<int32_t> minus1(N, -1);
    HIP_OK(hipMemset(st.hdr, 0, N * sizeof(snac_env_hdr)));
    HIP_OK(hipMemcpy(st.episode, minus1.data(), N * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemset(st.grid, 0, (size_t)N * sz.grid_elems * sz.grid_elem_bytes));
    HIP_OK(hipMemcpy((void*)st.plans, packed.data(), packed.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy((void*)st.plan_tb, tb.data(), P * 2, hipMemcpyHostToDevice));
    HIP_OK(hipMemset(stats, 0, 3 * N * sizeof(int64_t)));
    double* d_obs;
    float* d_rew;
    uint8_t* d_done;
    HIP_OK(hipMalloc((void**)&d_obs, (size_t)T * N * 51 * 8));
    HIP_OK(hipMalloc((void**)&d_rew, (size_t)T * N * 4));
    HIP_OK(hipMalloc((void**)&d_done, (size_t)T * N));
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));

    SNAC_CHECK(snac_reset(&d, &st, nullptr, nullptr, d_obs, stream));
    SNAC_CHECK(snac_rollout(&d, &st, T, 0, nullptr, nullptr, SNAC_OBS_ALL, d_obs, d_rew, d_done, stream));
    HIP_OK(hipStreamSynchronize(stream));
    std::vector<double> obs((size_t)T * N * 51);
    std::vector<float> rew((size_t)T * N);
    std::vector<uint8_t> done((size_t)T * N);
    HIP_OK(hipMemcpy(obs.data(), d_obs, obs.size() * 8, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(rew.data(), d_rew, rew.size() * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(done.data(), d_done, done.size(), hipMemcpyDeviceToHost));

    orc_batch* b = orc_batch_create(2, 1, N, full.data(), P, seed, 4242);
    if (!b || orc_batch_reset(b, nullptr, nullptr, nullptr)) return 4;
    std::vector<double> o2(obs.size());
    std::vector<float> r2(rew.size());
    std::vector<uint8_t> d2(done.size());
    if (orc_batch_rollout(b, T, 0, nullptr, nullptr, o2.data(), 0, r2.data(), d2.data(), 4)) return 5;
    const bool same = std::memcmp(obs.data(), o2.data(), obs.size() * 8) == 0 && std::memcmp(rew.data(), r2.data(), rew.size() * 4) == 0 &&
                      std::memcmp(done.data(), d2.data(), done.size()) == 0;
    std::vector<int64_t> hs(3 * N);
    HIP_OK(hipMemcpy(hs.data(), stats, hs.size() * 8, hipMemcpyDeviceToHost));
    long long eps = 0, eps2 = 0;
    for (int i = 0; i < N; ++i) { eps += hs[i]; eps2 += b->stat_episodes[i]; }
    orc_batch_destroy(b);
    std::printf("%s: %d envs x %d ticks, %lld episodes (oracle %lld)\n", same && eps == eps2 ? "PARITY OK" : "MISMATCH", N, T, eps, eps2);
    return same && eps == eps2 ? 0 : 6;
}
