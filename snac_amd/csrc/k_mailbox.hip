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

#include <chrono>
#include <thread>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

struct snac_mailbox {
    // ---- host -> device (one cache line, written by the host only).  The whole command is ONE 8-byte word, stored atomically: every
    // word the wave would have to fetch separately is one more round trip over the bus (a first version with four words: 10.4 us per step)
    //   bits 0-31 sequence number | 32-39 op (MB_STEP / MB_STEP_N / MB_QUIT) | 40-47 action (int8) | 48-51 step size | 52-63 state generation
    uint64_t cmd;
    uint32_t pad0[14];
    // a batch of 2 .. 64 envs (one env per lane): its actions and step sizes, written by the host BEFORE the command word (which then
    // carries no action); the wave fetches both lines with one trip
    int8_t actions[64];
    int8_t steps[64];
    // ---- device -> host (its own cache line)
    uint32_t ack_seq;            // = the command's sequence number once the row below is complete
    uint32_t alive;              // 1 while a wave is resident, 0 stored as its last act
    uint32_t steps_served;       // statistics
    uint32_t wt_seq;             // = ack_seq once the env's records in HBM are complete too (written behind the acknowledgement)
    uint32_t dbg[4];             // ticks of the 100 MHz clock of the last step: command seen -> stepped -> row stored -> fenced (+ the write-through)
    uint32_t pad1[8];
    float reward[64];            // per env, written with the rows
    uint8_t done[64];
    // ---- host side bookkeeping (never read by the device)
    hipStream_t stream;
    uint32_t req_seq, state_gen; // the host's own copies of what it last posted
    int32_t armed;               // a launch has been made and not yet seen to end
    int32_t launches;
    int32_t row_values;          // values per row
    int32_t device;
    uint32_t idle_us;
    int32_t num_envs;
    uint32_t pad2[6];
    double row[512];             // the observation rows [num_envs][row_values] of obs_dtype, written by the device (the allocation extends past 512 values when they need it)
};
static_assert(offsetof(snac_mailbox, actions) == 64 && offsetof(snac_mailbox, ack_seq) == 192 && offsetof(snac_mailbox, row) % 64 == 0, "mailbox cache lines");

namespace {

enum { MB_STEP = 1, MB_STEP_N = 2, MB_QUIT = 3 };   // MB_STEP_N: snac_mailbox_step_n -- reward / done always go to their own arrays too

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

// One wave, env e of the batch on lane e (N <= K::E <= 64; the facades: N = 1).  VAR: the batch's layout is a variant (the facades carry
// the record tail), OT = its observation type.
template <class K, typename OT, bool VAR>
__global__ __launch_bounds__(64) void k_mailbox(const KArgs a, snac_mailbox* mb, long long idle_ticks) {
    const int lane = threadIdx.x & 63;
    const int nenv = a.n;
    const bool active = lane < nenv;
    uint32_t* lds = wave_lds<K, 1>();
    Lane s;
    int episode = 0;
    auto load_state = [&]() {
        s.clear();
        s.r = 3; s.c = 3;
        if (active) { s.unpack(a.hdr[lane]); episode = a.episode[lane]; }
        K::load_grid(lds, a, 0, nenv, lane);
        for (int e = 0; e < nenv; ++e) K::load_plan(lds, a, e, __builtin_amdgcn_readlane(s.pidx, e), lane);
    };
    uint32_t seen = sys_load(&mb->ack_seq);                          // the host sets ack = req before it arms: nothing pending is lost
    uint32_t gen = 0xFFFFFFFFu, served = sys_load(&mb->steps_served);
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
                if ((served & 63u) == 0u) sys_store(&mb->dbg[3], dbg_wt + (uint32_t)(wall_clock64() - w0));
                sys_store(&mb->steps_served, served);
                __hip_atomic_store(&mb->wt_seq, seen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
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
        if (op == MB_QUIT) { seen = req; break; }
        int act = (int)(int8_t)((cmd >> 40) & 0xffu);
        int k = (int)((cmd >> 48) & 0xfu);
        if (nenv > 1) {                                              // a batch: every lane its own action and step size (one more trip)
            act = (int)sys_load(&mb->actions[active ? lane : 0]);
            k = (int)sys_load(&mb->steps[active ? lane : 0]);
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
        if (nenv == 1 && !(VAR && (a.tail & SNAC_TAIL_PLAN))) emit_one<K, OT, VAR>(lds, (OT*)mb->row, s, a, lane, reward, done ? 1 : 0);
        else emit_obs<K, OT, VAR, VAR && K::A != 8>(lds, (OT*)mb->row, nenv, s, a, lane, StepOut{reward, done ? 1 : 0});
        // (a single env whose row carries the record tail -- the drop-in classes -- has reward and done in the row: two stores over the bus less)
        if (active && (op == MB_STEP_N || !(VAR && (a.tail & SNAC_TAIL_RECORD)))) { mb->reward[lane] = (float)reward; mb->done[lane] = done ? 1 : 0; }
        const long long c2 = wall_clock64();
        __threadfence_system();                                      // the rows have left before the acknowledgement does
        const long long c3 = wall_clock64();
        if (lane == 0) __hip_atomic_store(&mb->ack_seq, req, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        // ---- behind the acknowledgement, off the host's critical path: the episodic sums and the write-through of the records
        if (__any(done)) {                                           // snac_step's episodic sums (the plans are in LDS; 3D: the running sum)
            const double v = K::iou(lds, s, active ? lane : 0);
            if (done) {
                a.stat_episodes[lane] += 1;
                a.stat_return[lane] += s.ep_ret;
                a.stat_iou_fx[lane] += __double2ll_rn(v * FX40);
            }
        }
        K::store_grid(lds, a, 0, nenv, lane);
        if (active) { a.hdr[lane] = s.pack(); a.episode[lane] = episode; }
        served += 1u;
        if (lane == 0 && (served & 63u) == 0u) {                     // the wave's own stamps, now and then (each is a store over the bus)
            sys_store(&mb->dbg[0], (uint32_t)(c1 - c0)); sys_store(&mb->dbg[1], (uint32_t)(c2 - c1)); sys_store(&mb->dbg[2], (uint32_t)(c3 - c2));
        }
        seen = req;
        wt_pending = true;                                           // fenced and reported behind the next poll (top of the loop)
        last = wall_clock64();
        dbg_wt = (uint32_t)(last - c3);
    }
    __threadfence_system();                                          // (a write-through still pending included)
    if (lane == 0) {
        sys_store(&mb->steps_served, served);
        __hip_atomic_store(&mb->ack_seq, seen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // (QUIT is acknowledged too)
        __hip_atomic_store(&mb->wt_seq, seen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&mb->alive, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <class K>
void launch_mb_k(const snac_env_desc* d, const KArgs& a, snac_mailbox* mb, long long ticks) {
    const bool f32 = d->obs_dtype == SNAC_OBS_F32;
    const dim3 g(1), b(64);
    if (a.variant) { if (f32) hipLaunchKernelGGL((k_mailbox<K, float, true>), g, b, 0, mb->stream, a, mb, ticks); else hipLaunchKernelGGL((k_mailbox<K, double, true>), g, b, 0, mb->stream, a, mb, ticks); }
    else { if (f32) hipLaunchKernelGGL((k_mailbox<K, float, false>), g, b, 0, mb->stream, a, mb, ticks); else hipLaunchKernelGGL((k_mailbox<K, double, false>), g, b, 0, mb->stream, a, mb, ticks); }
}
template <template <bool, int> class KT>
void launch_mb(const snac_env_desc* d, const KArgs& a, snac_mailbox* mb, long long ticks) {
    // the facades' single env keeps the 8-env LDS image (a 64-env 3D image is 87 KB to set up); batches take the 64-env one
    if (a.n <= 8) { if (d->dynamic) launch_mb_k<KT<true, 8>>(d, a, mb, ticks); else launch_mb_k<KT<false, 8>>(d, a, mb, ticks); }
    else { if (d->dynamic) launch_mb_k<KT<true, 64>>(d, a, mb, ticks); else launch_mb_k<KT<false, 64>>(d, a, mb, ticks); }
}

inline uint32_t host_load(const uint32_t* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
inline void post(snac_mailbox* mb, int op, int action, int step_size) {
    mb->req_seq += 1u;
    const uint64_t w = (uint64_t)mb->req_seq | ((uint64_t)(op & 0xff) << 32) | ((uint64_t)(uint8_t)(int8_t)action << 40) |
                       ((uint64_t)(step_size & 0xf) << 48) | ((uint64_t)(mb->state_gen & 0xfffu) << 52);
    __atomic_store_n(&mb->cmd, w, __ATOMIC_RELEASE);
}
inline void cpu_relax() {
#if defined(__x86_64__)
    _mm_pause();
#endif
}

int arm(snac_mailbox* mb, const snac_env_desc* d, const snac_state* st) {
    using namespace snac_detail;
    if (mb->armed) {                                                 // the previous wave has ended (alive == 0): retire its launch
        const hipError_t e = hipStreamSynchronize(mb->stream);
        if (e != hipSuccess) return fail_hip(e, "mailbox stream");
        mb->armed = 0;
    }
    KArgs a = make_args(d, st);
    a.pool = d->num_envs; a.stats_on = 1; a.T = 1; a.obs_mode = SNAC_OBS_ALL;
    __atomic_store_n(&mb->alive, 1u, __ATOMIC_RELEASE);             // (the wave clears it as its last act)
    const long long ticks = (long long)mb->idle_us * 100;           // wall_clock64(): the constant 100 MHz counter
    if (d->kind == SNAC_ENV_1D) launch_mb<K1D>(d, a, mb, ticks);
    else if (d->kind == SNAC_ENV_2D) launch_mb<K2D>(d, a, mb, ticks);
    else launch_mb<K3D>(d, a, mb, ticks);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { mb->alive = 0; return fail_hip(e, "mailbox launch"); }
    mb->armed = 1;
    mb->launches += 1;
    g_kernel = "k_mailbox";
    return SNAC_OK;
}

}  // namespace

extern "C" {

int snac_mailbox_create(const snac_env_desc* d, uint32_t idle_us, snac_mailbox** out) {
    using namespace snac_detail;
    if (!d || !out) return fail(SNAC_ERR_ARG, "null desc / out");
    if (d->num_envs < 1 || d->num_envs > 64) return fail(SNAC_ERR_UNSUPPORTED, "the mailbox steps a batch of 1 .. 64 envs (one wavefront, an env per lane)");
    if (d->kind < SNAC_ENV_1D || d->kind > SNAC_ENV_3D) return fail(SNAC_ERR_ARG, "unknown env kind");
    if (int rc = check_layout(d)) return rc;
    const int ld = base_obs_dim(d->kind) + tail_len(d->kind, d->obs_tail);
    const size_t row_bytes = (size_t)d->num_envs * (size_t)ld * sizeof(double);
    const size_t total = sizeof(snac_mailbox) + (row_bytes > sizeof(((snac_mailbox*)nullptr)->row) ? row_bytes - sizeof(((snac_mailbox*)nullptr)->row) : 0);
    snac_mailbox* mb = nullptr;
    // coherent (fine-grained) page-locked memory, mapped: device stores and host stores are visible to the other side while the
    // wave runs -- what the doorbell and the acknowledgement need (plain pinned memory is only guaranteed at kernel boundaries)
    hipError_t e = hipHostMalloc((void**)&mb, total, hipHostMallocCoherent | hipHostMallocMapped | hipHostMallocPortable);
    if (e != hipSuccess) return fail_hip(e, "hipHostMalloc(mailbox)");
    std::memset(mb, 0, total);
    mb->num_envs = d->num_envs;
    e = hipStreamCreateWithFlags(&mb->stream, hipStreamNonBlocking);  // its own queue: a resident wave must not sit in front of anyone's work
    if (e != hipSuccess) { (void)hipHostFree(mb); return fail_hip(e, "hipStreamCreate(mailbox)"); }
    mb->row_values = ld;
    mb->idle_us = idle_us ? idle_us : 1000;
    (void)hipGetDevice(&mb->device);
    *out = mb;
    return SNAC_OK;
}

double* snac_mailbox_row(snac_mailbox* mb) { return mb ? mb->row : nullptr; }
float* snac_mailbox_reward(snac_mailbox* mb) { return mb ? mb->reward : nullptr; }
uint8_t* snac_mailbox_done(snac_mailbox* mb) { return mb ? mb->done : nullptr; }

int snac_mailbox_touch(snac_mailbox* mb) {
    if (!mb) return snac_detail::fail(SNAC_ERR_ARG, "null mailbox");
    mb->state_gen += 1u;                                             // travels with the next command
    return SNAC_OK;
}

static int await_ack(snac_mailbox* mb, const snac_env_desc* d, const snac_state* st) {
    using namespace snac_detail;
    const uint32_t req = mb->req_seq;
    if (!host_load(&mb->alive)) {
        if (int rc = check_common(d, st)) return rc;
        if (d->num_envs != mb->num_envs) return fail(SNAC_ERR_ARG, "mailbox of another batch size");
        if (int rc = arm(mb, d, st)) return rc;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        if (host_load(&mb->ack_seq) == req) return SNAC_OK;
        if (!host_load(&mb->alive)) {                                // the wave left (idle timeout) -- before or after it saw this command?
            if (host_load(&mb->ack_seq) == req) return SNAC_OK;
            if (int rc = arm(mb, d, st)) return rc;                  // before: a new wave finds req != ack and serves it
        }
        cpu_relax();
        if ((spins & 0xFFFFu) == 0xFFFFu && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2))
            return fail(SNAC_ERR_HIP, "mailbox: no acknowledgement within 2 s");
    }
}

// One step of a ONE-env batch: posts (action, step_size), arms the wave if none is resident, spins until the row is acknowledged.
// Returns SNAC_OK with the row in snac_mailbox_row(); SNAC_ERR_HIP if no acknowledgement arrives within ~2 s (the state in HBM is as
// the last acknowledged step left it).
int snac_mailbox_step(snac_mailbox* mb, const snac_env_desc* d, const snac_state* st, int32_t action, int32_t step_size) {
    using namespace snac_detail;
    if (!mb) return fail(SNAC_ERR_ARG, "null mailbox");
    if (mb->num_envs != 1) return fail(SNAC_ERR_ARG, "snac_mailbox_step is for a batch of one env (snac_mailbox_step_n)");
    const int lim = d->kind == SNAC_ENV_1D ? 3 : (d->kind == SNAC_ENV_2D ? 5 : 8);
    post(mb, MB_STEP, (action >= 0 && action < lim) ? action : -1, step_size < 1 ? 1 : (step_size > 3 ? 3 : step_size));   // (every invalid action steps alike)
    return await_ack(mb, d, st);
}

// One vector step of a batch of 1 .. 64 envs: actions / step_size int8[num_envs] (host memory) are copied into the mailbox, the wave
// steps env e on lane e and writes rows [num_envs][obs_dim], reward float[num_envs] and done uint8[num_envs] back
// (snac_mailbox_row / _reward / _done).
int snac_mailbox_step_n(snac_mailbox* mb, const snac_env_desc* d, const snac_state* st, const int8_t* actions, const int8_t* step_size) {
    using namespace snac_detail;
    if (!mb || !actions || !step_size) return fail(SNAC_ERR_ARG, "null mailbox / actions / step_size");
    std::memcpy(mb->actions, actions, (size_t)mb->num_envs);
    std::memcpy(mb->steps, step_size, (size_t)mb->num_envs);
    post(mb, MB_STEP_N, actions[0], step_size[0] < 1 ? 1 : (step_size[0] > 3 ? 3 : step_size[0]));
    return await_ack(mb, d, st);
}

// Waits until the records in HBM hold the last acknowledged step (the wave writes them through BEHIND its acknowledgement): what an
// entry point that reads or changes them calls first.  A microsecond at most; SNAC_ERR_HIP after 2 s.
int snac_mailbox_settle(snac_mailbox* mb) {
    using namespace snac_detail;
    if (!mb || !mb->armed) return SNAC_OK;
    const uint32_t want = host_load(&mb->ack_seq);
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0; host_load(&mb->wt_seq) != want; ++spins) {
        if (!host_load(&mb->alive) && host_load(&mb->wt_seq) == host_load(&mb->ack_seq)) break;
        cpu_relax();
        if ((spins & 0xFFFFu) == 0xFFFFu && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2))
            return fail(SNAC_ERR_HIP, "mailbox: write-through not seen within 2 s");
    }
    return SNAC_OK;
}

// Ends the resident wave (if any) and waits until it is gone.
int snac_mailbox_quit(snac_mailbox* mb) {
    using namespace snac_detail;
    if (!mb) return SNAC_OK;
    if (mb->armed) {
        if (host_load(&mb->alive)) post(mb, MB_QUIT, 0, 1);
        const hipError_t e = hipStreamSynchronize(mb->stream);       // (an idle wave ends by itself within idle_us)
        mb->armed = 0;
        if (e != hipSuccess) return fail_hip(e, "mailbox stream");
        __atomic_store_n(&mb->ack_seq, mb->req_seq, __ATOMIC_RELEASE);   // nothing is pending for the next wave
        __atomic_store_n(&mb->wt_seq, mb->req_seq, __ATOMIC_RELEASE);
    }
    return SNAC_OK;
}

int snac_mailbox_destroy(snac_mailbox* mb) {
    if (!mb) return SNAC_OK;
    const int rc = snac_mailbox_quit(mb);
    (void)hipStreamDestroy(mb->stream);
    (void)hipHostFree(mb);
    return rc;
}

// {launches, steps served by resident waves, alive, idle_us, and of the last step, in ticks of the GPU's 100 MHz clock: step, row
// stores issued, fence before the acknowledgement, write-through behind it}
int snac_mailbox_stats(const snac_mailbox* mb, uint32_t out[8]) {
    if (!mb || !out) return snac_detail::fail(SNAC_ERR_ARG, "null mailbox / out");
    out[0] = (uint32_t)mb->launches; out[1] = host_load(&mb->steps_served); out[2] = host_load(&mb->alive); out[3] = mb->idle_us;
    for (int i = 0; i < 4; ++i) out[4 + i] = host_load(&mb->dbg[i]);
    return SNAC_OK;
}

}  // extern "C"
