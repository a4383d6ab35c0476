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
    // the trajectory in trajectory memory (snac_traj_alloc: one virtual range over chunks from two slices of physical memory), asked
    // for generously so that the block takes the measured layout (>= 1 GiB); everything else stays hipMalloc'ed
    const size_t obs_bytes = (size_t)T * N * 51 * 8, block_bytes = obs_bytes < ((size_t)1100 << 20) ? ((size_t)1100 << 20) : obs_bytes;
    SNAC_CHECK(snac_traj_alloc(block_bytes, 0, (void**)&d_obs));
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
    if (!(same && eps == eps2)) { std::printf("MISMATCH in the rollout\n"); return 6; }

    // ---- tree search from the same host: the batch as a node pool (snac_transition), reference-format states in and out
    // (snac_export_grid / snac_import_state), equality_operator (snac_obs_equal) --------------------------------------------
    const int M = 300;                                  // edges: parents in [0, 300), children written to rows [400, 700)
    std::vector<int32_t> src(M), dst(M);
    std::vector<int8_t> acts(M), ks(M);
    for (int i = 0; i < M; ++i) { src[i] = (i * 7) % 300; dst[i] = 400 + i; acts[i] = (int8_t)(i % 5); ks[i] = (int8_t)(1 + i % 3); }
    int32_t *d_src, *d_dst;
    int8_t *d_a, *d_k;
    uint8_t* d_eq;
    HIP_OK(hipMalloc((void**)&d_src, M * 4)); HIP_OK(hipMalloc((void**)&d_dst, M * 4));
    HIP_OK(hipMalloc((void**)&d_a, M)); HIP_OK(hipMalloc((void**)&d_k, M)); HIP_OK(hipMalloc((void**)&d_eq, M));
    HIP_OK(hipMemcpy(d_src, src.data(), M * 4, hipMemcpyHostToDevice)); HIP_OK(hipMemcpy(d_dst, dst.data(), M * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_a, acts.data(), M, hipMemcpyHostToDevice)); HIP_OK(hipMemcpy(d_k, ks.data(), M, hipMemcpyHostToDevice));
    SNAC_CHECK(snac_transition(&d, &st, M, d_src, d_dst, 0, d_a, d_k, d_obs, d_rew, d_done, stream));
    // a child's observation compared with itself and with its neighbour's (equality_operator)
    std::vector<int32_t> other(M);
    for (int i = 0; i < M; ++i) other[i] = (i + 1) % M;
    int32_t* d_other;
    HIP_OK(hipMalloc((void**)&d_other, M * 4));
    HIP_OK(hipMemcpy(d_other, other.data(), M * 4, hipMemcpyHostToDevice));
    SNAC_CHECK(snac_obs_equal(&d, d_obs, nullptr, M, d_obs, d_other, M, M, d_eq, stream));
    // environment_memory of the children out, and back in as (position, memory, count_brick, count_step) tuples to rows 0..
    double* d_mem;
    HIP_OK(hipMalloc((void**)&d_mem, (size_t)N * 676 * 8));
    SNAC_CHECK(snac_export_grid(&d, &st, d_mem, stream));
    HIP_OK(hipStreamSynchronize(stream));
    std::vector<double> o3((size_t)M * 51), o4((size_t)M * 51);
    std::vector<float> r3(M), r4(M);
    std::vector<uint8_t> d3(M), d4(M), eq(M);
    HIP_OK(hipMemcpy(o3.data(), d_obs, o3.size() * 8, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(r3.data(), d_rew, M * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(d3.data(), d_done, M, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(eq.data(), d_eq, M, hipMemcpyDeviceToHost));
    if (orc_batch_transition(b, M, src.data(), dst.data(), 0, acts.data(), ks.data(), o4.data(), r4.data(), d4.data())) return 7;
    bool ok = std::memcmp(o3.data(), o4.data(), o3.size() * 8) == 0 && std::memcmp(r3.data(), r4.data(), M * 4) == 0 &&
              std::memcmp(d3.data(), d4.data(), M) == 0;
    for (int i = 0; i < M && ok; ++i)
        ok = eq[i] == (std::memcmp(&o4[(size_t)i * 51], &o4[(size_t)other[i] * 51], 51 * 8) == 0 ? 1 : 0);
    std::vector<snac_env_hdr> hdr(N);
    HIP_OK(hipMemcpy(hdr.data(), st.hdr, N * sizeof(snac_env_hdr), hipMemcpyDeviceToHost));
    std::vector<int32_t> pos(2 * M), cb(M), cs(M), pidx(M);
    for (int i = 0; i < M; ++i) {
        const snac_env_hdr& h = hdr[dst[i]];
        const orc_env& e = b->envs[dst[i]];
        ok = ok && h.pos_r == e.pos[0] && h.pos_c == e.pos[1] && h.count_brick == e.cb && h.count_step == e.cs && h.total_brick == e.tb;
        pos[2 * i] = h.pos_r; pos[2 * i + 1] = h.pos_c; cb[i] = h.count_brick; cs[i] = h.count_step; pidx[i] = h.plan_idx;
    }
    int32_t *d_pos, *d_cb, *d_cs, *d_pidx;
    HIP_OK(hipMalloc((void**)&d_pos, 2 * M * 4)); HIP_OK(hipMalloc((void**)&d_cb, M * 4));
    HIP_OK(hipMalloc((void**)&d_cs, M * 4)); HIP_OK(hipMalloc((void**)&d_pidx, M * 4));
    HIP_OK(hipMemcpy(d_pos, pos.data(), 2 * M * 4, hipMemcpyHostToDevice)); HIP_OK(hipMemcpy(d_cb, cb.data(), M * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_cs, cs.data(), M * 4, hipMemcpyHostToDevice)); HIP_OK(hipMemcpy(d_pidx, pidx.data(), M * 4, hipMemcpyHostToDevice));
    SNAC_CHECK(snac_import_state(&d, &st, M, nullptr, d_pos, d_cb, d_cs, d_pidx, nullptr, d_mem + (size_t)400 * 676, stream));
    SNAC_CHECK(snac_observe(&d, &st, d_obs, stream));   // rows 0..M-1 now hold the children: same observations as above
    HIP_OK(hipStreamSynchronize(stream));
    std::vector<double> o5((size_t)M * 51);
    HIP_OK(hipMemcpy(o5.data(), d_obs, o5.size() * 8, hipMemcpyDeviceToHost));
    ok = ok && std::memcmp(o5.data(), o4.data(), o5.size() * 8) == 0;
    orc_batch_destroy(b);
    SNAC_CHECK(snac_traj_free(d_obs));
    std::printf("%s: %d envs x %d ticks, %lld episodes (oracle %lld); %d tree edges, import / export, equality\n",
                ok ? "PARITY OK" : "MISMATCH", N, T, eps, eps2, M);
    return ok ? 0 : 8;
}
