// k_mailbox.hip -- the resident single-env stepper behind the drop-in classes (round 5).
//
// A script that drives ONE env (script/DQN/2d/DQN_2d_dynamic.py:214: `state_next, r, done = env.step(action)` per loop turn) pays,
// on the launch path, one kernel launch and one stream wait per step: 15.5 of the 17 us a facade step takes, against the 9 us of the
// reference's own Python step (Env/2D/DMP_Env_2D_dynamic_usedata_plan.py:85-147).  Here ONE wavefront stays resident for as long as
// the host keeps stepping: it polls a doorbell in page-locked host memory (the mailbox: sequence number, action, step size), steps the
// env with the same K::step as every tile kernel, writes the observation row (+ the record tail: reward, done, position, counters) back
// into the mailbox over the bus, writes the env's state through to its records in HBM, and acknowledges with the sequence number.
// A step is then a store, a spin on a host cache line and a copy of 59 doubles: no launch, no stream wait.
//
// Exit conditions every wave reaches: the QUIT command (snac_mailbox_quit / destroy / atexit of the Python side), or `idle_us`
// microseconds of the GPU's constant 100 MHz clock without a command (default 1 ms) -- so the wave is gone a millisecond after the host
// stops stepping, whatever the host does (a trainer that spends its time in the network between steps simply finds the mailbox unarmed
// and re-arms it: one launch, as before), and device-wide synchronisations elsewhere in the process wait a millisecond at most.
// The state in HBM is complete whenever the host has seen an acknowledgement (the write-through is fenced before the ack), so every
// other entry point may READ it at any time; entry points that WRITE it bump the mailbox's generation (snac_mailbox_touch) and the
// wave reloads the records before its next step.
#include "snac_dev.h"

#include "mailbox_host.h"   // struct snac_mailbox, the host half of the protocol and the entry points (also built by gcc against a fake HIP layer: tests/native)

namespace {

template <typename T>
__device__ __forceinline__ T sys_load(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
template <typename T>
__device__ __forceinline__ void sys_store(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// The row of a ONE-env batch in one store instruction: lane l holds value l of the row (window cell, scalar slot, position / record tail
// value) -- write_obs walks the tile machinery for eight envs and sends window, tails and record as separate stores over the bus (0.9 us of
// a 5.5 us step; this: see profiles/r05_mailbox.txt).  Layout rules are write_obs's / write_scalars'; rows with the plan tail keep the
// general path.  s, reward, done: lane 0's.
template <class K, typename OT, bool VAR>
__device__ __forceinline__ void emit_one(uint32_t* lds, OT* row, const Lane& s, const KArgs& a, int lane, int reward, int done) {
    const int r = __builtin_amdgcn_readlane(s.r, 0), c = __builtin_amdgcn_readlane(s.c, 0), cb = __builtin_amdgcn_readlane(s.cb, 0);
    const int cs = __builtin_amdgcn_readlane(s.cs, 0), tb = __builtin_amdgcn_readlane(s.tb, 0), pidx = __builtin_amdgcn_readlane(s.pidx, 0);
    const int rw = __builtin_amdgcn_readlane(reward, 0), dn = __builtin_amdgcn_readlane(done, 0);
    const bool norm = VAR ? (a.sc_norm != 0) : K::DYN;
    const double v0 = norm ? (double)cb / (double)tb : (double)cb, v1 = norm ? (double)cs / (double)a.total_step : (double)cs;
    const int LD = VAR ? a.ld : K::D;
    double val;
    if (lane < K::W) {
        int v;
        if constexpr (K::D == 7) v = K::hmap(lds)[r - 2 + lane];                               // 1D: the five cells round the position
        else if constexpr (K::A == 8) { const int i = lane / 7, j = lane - 7 * i; v = K::hmap(lds)[(r - 3 + i) * 26 + (c - 3 + j)]; }
        else { const int i = lane / 7, j = lane - 7 * i; const uint64_t w = K::cells(lds)[(r - 3 + i) * K::RS]; v = ((int)((uint32_t)(w >> (2 * (c - 3 + j))) << 30)) >> 30; }
        if constexpr (VAR) v = v < 0 ? a.frame_val : v;
        val = (double)v;
    } else if (lane < K::D) {
        val = lane == K::W ? v0 : v1;
    } else {
        int ti = lane - K::D, out = 0;
        if constexpr (VAR) {
            if (a.tail & SNAC_TAIL_POSITION) {
                constexpr int PN = K::D == 7 ? 1 : 2;
                if (ti >= 0 && ti < PN) out = ti == 0 ? r : c;
                ti -= PN;
            }
            if ((a.tail & SNAC_TAIL_RECORD) && ti >= 0) out = record_value(min(ti, 7), rw, dn, r, K::D == 7 ? 0 : c, cb, cs, tb, pidx);
        }
        val = (double)out;
    }
    if (lane < LD) row[lane] = (OT)val;
}

// One block = one wave: wave `wv` (the block index) of the mailbox takes envs [64 wv, 64 wv + 64) of the batch, env 64 wv + e on lane e (the
// facades: N = 1, one wave; a batch of up to 256 envs: four blocks of one launch, on four CUs, polling the same doorbell -- no barrier
// between them, every wave leaves on its own idle clock, the host waits for every wave's acknowledgement).  VAR: the batch's layout is a variant (the facades carry the record
// tail), OT = its observation type.
template <class K, typename OT, bool VAR>
__global__ __launch_bounds__(64) void k_mailbox(const KArgs a, snac_mailbox* mb, long long idle_ticks) {
    const int lane = threadIdx.x & 63, wv = (int)blockIdx.x;
    const int env0 = wv * MB_WAVE_ENVS;
    const int nenv = min(MB_WAVE_ENVS, a.n - env0);
    const bool active = lane < nenv;
    const int env = env0 + (active ? lane : 0);
    uint32_t* lds = wave_lds<K, 1>();
    Lane s;
    int episode = 0;
    auto load_state = [&]() {
        s.clear();
        s.r = 3; s.c = 3;
        if (active) { s.unpack(a.hdr[env]); episode = a.episode[env]; }
        K::load_grid(lds, a, env0, nenv, lane);
        for (int e = 0; e < nenv; ++e) K::load_plan(lds, a, e, __builtin_amdgcn_readlane(s.pidx, e), lane);
    };
    uint32_t seen = sys_load(&mb->ack_seq[wv]);                      // the host sets ack = req before it arms: nothing pending is lost
    uint32_t gen = 0xFFFFFFFFu, served = sys_load(&mb->steps_served[wv]);
    bool quit = false;
    long long last = wall_clock64();
    bool wt_pending = false;
    uint32_t dbg_wt = 0;
    for (;;) {
        const uint64_t cmd = sys_load(&mb->cmd);
        if (wt_pending) {
            // the write-through of the step before: its stores were issued behind that step's acknowledgement and have had this poll's
            // trip over the bus to complete -- fenced and reported here, they cost the next step nothing
            const long long w0 = wall_clock64();
            __threadfence_system();
            if (lane == 0) {
                if (wv == 0 && (served & 63u) == 0u) sys_store(&mb->dbg[3], dbg_wt + (uint32_t)(wall_clock64() - w0));
                sys_store(&mb->steps_served[wv], served);
                __hip_atomic_store(&mb->wt_seq[wv], seen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            wt_pending = false;
        }
        const uint32_t req = (uint32_t)cmd;
        if (req == seen) {
            if (wall_clock64() - last > idle_ticks) break;           // nobody is stepping: leave the GPU
            __builtin_amdgcn_s_sleep(1);
            continue;
        }
        const int op = (int)((cmd >> 32) & 0xffu);
        if (op == MB_QUIT) { seen = req; quit = true; break; }       // (also a withdrawn step: mailbox_host.h withdraw())
        int act = (int)(int8_t)((cmd >> 40) & 0xffu);
        int k = (int)((cmd >> 48) & 0xfu);
        if (a.n > 1) {                                               // a batch: every lane its own action and step size (one more trip)
            act = (int)sys_load(&mb->actions[env]);
            k = (int)sys_load(&mb->steps[env]);
        }
        k = min(max(k, 1), 3);
        const uint32_t g = (uint32_t)(cmd >> 52);
        if (g != gen) {                                              // first command of this wave, or the records were changed under it
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");             // (system scope: nothing stale from this CU's caches)
            load_state();
            gen = g;
        }
        int reward = 0;
        bool done = false;
        const long long c0 = wall_clock64();
        if (active) {
            K::step(lds, a, s, act, k, a.ts_done, a.brick_gt, lane, reward, done);
            s.ep_ret = clamp16(s.ep_ret + reward);
            s.flags = done ? SNAC_FLAG_NEED_RESET : 0;
        }
        const long long c1 = wall_clock64();
        if (a.n == 1 && !(VAR && (a.tail & SNAC_TAIL_PLAN))) emit_one<K, OT, VAR>(lds, (OT*)mb->row, s, a, lane, reward, done ? 1 : 0);
        else emit_obs<K, OT, VAR, VAR && K::A != 8>(lds, (OT*)mb->row + (size_t)env0 * (VAR ? a.ld : K::D), nenv, s, a, lane, StepOut{reward, done ? 1 : 0});
        // (a single env whose row carries the record tail -- the drop-in classes -- has reward and done in the row: two stores over the bus less)
        if (active && (op == MB_STEP_N || !(VAR && (a.tail & SNAC_TAIL_RECORD)))) { mb->reward[env] = (float)reward; mb->done[env] = done ? 1 : 0; }
        const long long c2 = wall_clock64();
        __threadfence_system();                                      // the rows have left before the acknowledgement does
        const long long c3 = wall_clock64();
        if (lane == 0) __hip_atomic_store(&mb->ack_seq[wv], req, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        // ---- behind the acknowledgement, off the host's critical path: the episodic sums and the write-through of the records
        if (__any(done)) {                                           // snac_step's episodic sums (the plans are in LDS; 3D: the running sum)
            const double v = K::iou(lds, s, active ? lane : 0);
            if (done) {
                a.stat_episodes[env] += 1;
                a.stat_return[env] += s.ep_ret;
                a.stat_iou_fx[env] += __double2ll_rn(v * FX40);
            }
        }
        K::store_grid(lds, a, env0, nenv, lane);
        if (active) { a.hdr[env] = s.pack(); a.episode[env] = episode; }
        served += 1u;
        if (lane == 0 && wv == 0 && (served & 63u) == 0u) {                     // the wave's own stamps, now and then (each is a store over the bus)
            sys_store(&mb->dbg[0], (uint32_t)(c1 - c0)); sys_store(&mb->dbg[1], (uint32_t)(c2 - c1)); sys_store(&mb->dbg[2], (uint32_t)(c3 - c2));
        }
        seen = req;
        wt_pending = true;                                           // fenced and reported behind the next poll (top of the loop)
        last = wall_clock64();
        dbg_wt = (uint32_t)(last - c3);
    }
    __threadfence_system();                                          // (a write-through still pending included)
    if (lane == 0) {
        sys_store(&mb->steps_served[wv], served);
        if (quit) sys_store(&mb->quit_seq[wv], seen);                 // before the acknowledgement: this sequence number was a QUIT, nothing was stepped
        __hip_atomic_store(&mb->ack_seq[wv], seen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // (QUIT is acknowledged too)
        __hip_atomic_store(&mb->wt_seq[wv], seen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&mb->alive[wv], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <class K>
void launch_mb_k(const snac_env_desc* d, const KArgs& a, snac_mailbox* mb, long long ticks) {
    const bool f32 = d->obs_dtype == SNAC_OBS_F32;
    const dim3 g((unsigned)mb->num_waves), b(64);
    hipStream_t st = mb->stream;
    if (a.variant) { if (f32) hipLaunchKernelGGL((k_mailbox<K, float, true>), g, b, 0, st, a, mb, ticks); else hipLaunchKernelGGL((k_mailbox<K, double, true>), g, b, 0, st, a, mb, ticks); }
    else { if (f32) hipLaunchKernelGGL((k_mailbox<K, float, false>), g, b, 0, st, a, mb, ticks); else hipLaunchKernelGGL((k_mailbox<K, double, false>), g, b, 0, st, a, mb, ticks); }
}
template <template <bool, int> class KT>
void launch_mb(const snac_env_desc* d, const KArgs& a, snac_mailbox* mb, long long ticks) {
    // the facades' single env keeps the 8-env LDS image (a 64-env 3D image is 87 KB to set up); batches take the 64-env one
    if (a.n <= 8) { if (d->dynamic) launch_mb_k<KT<true, 8>>(d, a, mb, ticks); else launch_mb_k<KT<false, 8>>(d, a, mb, ticks); }
    else { if (d->dynamic) launch_mb_k<KT<true, 64>>(d, a, mb, ticks); else launch_mb_k<KT<false, 64>>(d, a, mb, ticks); }
}

}  // namespace

// ---- the hooks of mailbox_host.h: descriptor / state checks and the launch of the waves ---------------------------------------------
namespace snac_mb {

int hook_check_desc(const snac_env_desc* d, int* row_values) {
    using namespace snac_detail;
    if (d->kind < SNAC_ENV_1D || d->kind > SNAC_ENV_3D) return fail(SNAC_ERR_ARG, "unknown env kind");
    if (int rc = check_layout(d)) return rc;
    *row_values = base_obs_dim(d->kind) + tail_len(d->kind, d->obs_tail);
    return SNAC_OK;
}

int hook_check_state(const snac_env_desc* d, const snac_state* st) { return snac_detail::check_common(d, st); }

int hook_launch(snac_mailbox* mb, const snac_env_desc* d, const snac_state* st) {
    using namespace snac_detail;
    KArgs a = make_args(d, st);
    a.pool = d->num_envs; a.stats_on = 1; a.T = 1; a.obs_mode = SNAC_OBS_ALL;
    const long long ticks = (long long)mb->idle_us * 100;           // wall_clock64(): the constant 100 MHz counter
    if (d->kind == SNAC_ENV_1D) launch_mb<K1D>(d, a, mb, ticks);
    else if (d->kind == SNAC_ENV_2D) launch_mb<K2D>(d, a, mb, ticks);
    else launch_mb<K3D>(d, a, mb, ticks);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "mailbox launch");
    g_kernel = "k_mailbox";
    return SNAC_OK;
}

}  // namespace snac_mb
