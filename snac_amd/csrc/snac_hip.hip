// snac_hip.hip -- the C ABI of include/snac_hip.h and the dispatch: which kernel a call runs on (launch()).  The kernels live in the
// k_*.hip translation units beside this one (snac_dev.h has the map), trajectory memory in snac_traj.hip.
#include "snac_dev.h"

namespace snac_detail {
thread_local char g_err[256] = "";
thread_local const char* g_kernel = "";

// ------------------------------------------------------------------------------------------------
// host side
// the layout flags of the descriptor (include/snac_hip.h "Observation-layout variants")
int check_layout(const snac_env_desc* d) {
    if (d->frame_value != 0 && d->frame_value != -1 && d->frame_value != 2) return fail(SNAC_ERR_ARG, "frame_value must be -1 (or 0) or 2");
    if (d->frame_value == 2 && d->kind == SNAC_ENV_3D) return fail(SNAC_ERR_UNSUPPORTED, "frame_value 2 is a 1D / 2D layout (the 3D rules test the frame for -1)");
    if (d->obs_scalars < SNAC_SCALARS_DEFAULT || d->obs_scalars > SNAC_SCALARS_NORM) return fail(SNAC_ERR_ARG, "unknown obs_scalars");
    if (d->obs_tail & ~(SNAC_TAIL_POSITION | SNAC_TAIL_PLAN | SNAC_TAIL_RECORD)) return fail(SNAC_ERR_ARG, "unknown bits in obs_tail");
    if (d->reserved != 0) return fail(SNAC_ERR_ARG, "snac_env_desc.reserved must be 0");
    return SNAC_OK;
}
int base_obs_dim(int kind) { return kind == SNAC_ENV_1D ? 7 : 51; }
int tail_len(int kind, int tail) {
    return ((tail & SNAC_TAIL_POSITION) ? (kind == SNAC_ENV_1D ? 1 : 2) : 0) + ((tail & SNAC_TAIL_PLAN) ? (kind == SNAC_ENV_1D ? 30 : 400) : 0) +
           ((tail & SNAC_TAIL_RECORD) ? 8 : 0);
}

int check_common(const snac_env_desc* d, const snac_state* st) {
    if (!d || !st) return fail(SNAC_ERR_ARG, "null desc/state");
    if (d->kind < SNAC_ENV_1D || d->kind > SNAC_ENV_3D) return fail(SNAC_ERR_ARG, "unknown env kind");
    if (d->num_envs <= 0) return fail(SNAC_ERR_ARG, "num_envs must be positive");
    if (d->num_plans <= 0 || d->num_plans > 32767) return fail(SNAC_ERR_ARG, "num_plans out of range");
    if (d->static_plan < 0 || d->static_plan >= d->num_plans) return fail(SNAC_ERR_ARG, "static_plan out of range");
    if (d->obs_dtype != SNAC_OBS_F64 && d->obs_dtype != SNAC_OBS_F32) return fail(SNAC_ERR_ARG, "unknown obs_dtype");
    // the running return is an int16 and a step pays at most 10: 3000 steps cannot overflow it (the reference: <= 1300)
    if (d->total_step < 0 || d->total_step > 3000) return fail(SNAC_ERR_ARG, "total_step out of range (0..3000)");
    if (d->rules & ~(SNAC_RULE_BRICK_GT | SNAC_RULE_TIME_GT)) return fail(SNAC_ERR_ARG, "unknown bits in rules");
    if (int rc = check_layout(d)) return rc;
    if (!st->hdr || !st->episode || !st->grid || !st->plans || !st->plan_tb || !st->stat_episodes || !st->stat_return ||
        !st->stat_iou_fx)
        return fail(SNAC_ERR_ARG, "null pointer in snac_state");
    return SNAC_OK;
}

KArgs make_args(const snac_env_desc* d, const snac_state* st) {
    KArgs a;
    std::memset(&a, 0, sizeof(a));
    a.n = d->num_envs; a.num_plans = d->num_plans; a.static_plan = d->static_plan;
    a.total_step = d->total_step > 0 ? d->total_step
                                     : (d->kind == SNAC_ENV_1D ? 750 : (d->kind == SNAC_ENV_2D ? 600 : (d->dynamic ? 1000 : 1300)));
    a.brick_gt = (d->rules & SNAC_RULE_BRICK_GT) ? 1 : 0;
    a.ts_done = a.total_step + ((d->rules & SNAC_RULE_TIME_GT) ? 1 : 0);
    a.key_step = stream_key(d->seed, 0); a.key_plan = stream_key(d->seed, 1);
    a.env_id_base = d->env_id_base;
    a.hdr = (int4*)st->hdr; a.episode = st->episode; a.grid = st->grid; a.plans = st->plans; a.plan_tb = st->plan_tb;
    a.stat_episodes = st->stat_episodes; a.stat_return = st->stat_return; a.stat_iou_fx = st->stat_iou_fx;
    a.plan_scalar = -1;
    a.frame_val = d->frame_value == 2 ? 2 : -1;
    a.sc_norm = d->obs_scalars == SNAC_SCALARS_DEFAULT ? (d->dynamic ? 1 : 0) : (d->obs_scalars == SNAC_SCALARS_NORM ? 1 : 0);
    a.tail = d->obs_tail;
    a.ld = base_obs_dim(d->kind) + tail_len(d->kind, d->obs_tail);
    a.variant = (a.frame_val != -1 || a.sc_norm != (d->dynamic ? 1 : 0) || a.tail != 0) ? 1 : 0;
    return a;
}


// tile size: enough tiles to give every SIMD of the 256 CUs a few waves; SNAC_TILE overrides (tuning)
int pick_tile(int kind, int n) {
    static const int forced = [] { const char* e = std::getenv("SNAC_TILE"); return e ? std::atoi(e) : 0; }();
    if (kind == SNAC_ENV_3D) return forced == 16 ? 16 : 8;   // 8: 17 KB of LDS per wave, 9 waves per CU; measured +10-15 % over 16
    if (forced == 8 || forced == 16 || forced == 32 || forced == 64) return forced;
    // measured per kind (tools/ab_time.py sweeps): 2D wants large tiles early (E x 408-byte store runs),
    // 1D's 56-byte rows do not care and prefer more, smaller waves
    const int shift = kind == SNAC_ENV_1D ? 1 : 0;
    if (n >= (64 * 1024) << shift) return 64;   // 2D: >= one wave per SIMD on 256 CUs; best at N = 65536 (profiles/)
    if (n >= (32 * 1024) << shift) return 32;
    if (n >= (16 * 1024) << shift) return 16;
    return 8;                        // small batches: one-wave blocks of 8 envs, so that 4096 envs still reach every CU
}

// SNAC_3D_PIPELINE=0 keeps every launch on the generic tile kernels (A/B timing, tests of both paths)
bool pipeline_off() {
    static const bool off = [] { const char* e = std::getenv("SNAC_3D_PIPELINE"); return e && e[0] == '0'; }();
    return off;
}

// 3D rollouts by blocks of 64 envs; SNAC_3D_BLOCK=0 keeps them on k_rollout3d (A/B timing, tests of both paths)
bool roll3db_ok(const KArgs& a, bool f32) {
    static const bool off = [] { const char* e = std::getenv("SNAC_3D_BLOCK"); return e && e[0] == '0'; }();
    static const int nmin = [] { const char* e = std::getenv("SNAC_3D_BLOCK_MIN"); return e ? std::atoi(e) : -1; }();   // (tuning)
    // where k_rollout3d's one-wave blocks stop being faster: float64 rows 4096 envs 1.02 against 1.04 ms, 6144 level, 8192 1.08 against
    // 1.04; float32 rows 4096 envs 1.02 against 0.97 already
    const int lim = nmin >= 0 ? nmin : (f32 ? 4096 : 6144);
    return !off && a.n >= lim && !a.variant && (a.obs_mode == SNAC_OBS_ALL || a.obs_mode == SNAC_OBS_TILED) && a.num_plans <= TB_MAX && (a.n & 3) == 0 &&
           (((uintptr_t)a.obs) & 15) == 0 && !pipeline_off();
}

// SNAC_2D_STAGE=0 keeps 2D rollouts on the tile kernel (A/B timing, tests of both paths)
bool stage2d_off() {
    static const bool off = [] { const char* e = std::getenv("SNAC_2D_STAGE"); return e && e[0] == '0'; }();
    return off;
}
bool roll2d_ok(const KArgs& a, int E) {
    return E == 64 && (a.obs_mode == SNAC_OBS_ALL || a.obs_mode == SNAC_OBS_TILED) &&
           (a.n & 3) == 0 && ((uintptr_t)a.obs & 15) == 0 && !pipeline_off() && !stage2d_off();
}

// time-parallel 2D rollouts (one wave per env, lane = tick): small and middle batches, where the lane-per-env kernels are bound by the
// chain of their ticks (0.5-0.7 ms per 600 ticks at every N <= 16 384) or leave CUs empty (one wave of 64 envs per CU at N = 16 384).
// Where it stops paying was measured on trajectory memory (profiles/r04_2d_midrange.txt, part 3): float64 rows up to 19 456 envs --
// except just below 16 384, where the tile kernel's 256 waves fill the chip exactly (5.8 against 5.55 TB/s) --, float32 rows up to
// 30 719 (16 384 envs: 4.2 against 2.9 TB/s); batches whose per-tick runs are not 16-byte pieces (odd N) only up to 8192.
// SNAC_2D_TP=0 keeps every 2D rollout off this kernel (A/B timing, tests of both paths), SNAC_2D_TP_MAX=n replaces the limits by n
bool roll2dt_ok(const KArgs& a, bool f32) {
    static const bool off = [] { const char* e = std::getenv("SNAC_2D_TP"); return e && e[0] == '0'; }();
    static const int nmax = [] { const char* e = std::getenv("SNAC_2D_TP_MAX"); return e ? std::atoi(e) : 0; }();   // (tuning)
    if (off || !(a.obs_mode == SNAC_OBS_ALL || a.obs_mode == SNAC_OBS_TILED) || pipeline_off()) return false;
    if (a.variant) {
        // the layout variants (rows of a.ld values: k_rollout2dt<.., VAR>): whole groups of four envs and 16-byte pieces only.  Where
        // the lane-per-env kernels take over again was measured with the 451-value rows of the PPO copies (profiles/r04_2d_layouts.txt)
        // (profiles/r04_2d_layouts.txt, part 3): 6.0-6.4 TB/s from 1024 envs on against k_rollout2d's 5.45 at 49 152 envs and 7.2 at 65 536;
        // short rows (no plan tail: 53 .. 61 values) level off at 5.8e9 env-steps/s and hand over near 6144 envs
        static const int vmax = [] { const char* e = std::getenv("SNAC_2D_TP_VAR_MAX"); return e ? std::atoi(e) : 0; }();   // (tuning)
        const int lim = vmax ? vmax : ((a.tail & SNAC_TAIL_PLAN) ? 49152 : 6144);
        return (a.n & 3) == 0 && (((uintptr_t)a.obs) & 15) == 0 && a.n <= lim;
    }
    if (nmax) return a.n <= nmax;
    const size_t rowb = (size_t)K2D<true, 64>::D * (f32 ? 4 : 8);
    const bool pieces = (((uintptr_t)a.obs) & 15) == 0 && (a.obs_mode == SNAC_OBS_TILED || (((size_t)a.n * rowb) & 15) == 0);
    if (!pieces) return a.n <= 8192;
    if (f32) return a.n < 30720;
    return a.n <= 15872 || (a.n > 16384 && a.n <= 19456);
}

// time-parallel 1D rollouts (one wave per env, lane = tick).  Its rate levels off at 6-7e10 env-steps/s (instruction issue: ~9 per
// env-step), the tile kernel's keeps growing with the batch: float64 rows 49 152 envs 0.54 against 0.68 ms per 750 ticks, 65 536
// 0.72-0.87 against 0.72; float32 rows 65 536 envs 0.65 against 0.73, 131 072 1.35 against 0.94 (profiles/r03_1d_time_parallel.txt).
// SNAC_1D_TP=0 keeps every 1D rollout on the tile kernel (A/B timing)
bool roll1dt_ok(const KArgs& a, bool f32) {
    static const bool off = [] { const char* e = std::getenv("SNAC_1D_TP"); return e && e[0] == '0'; }();
    static const int nmax = [] { const char* e = std::getenv("SNAC_1D_TP_MAX"); return e ? std::atoi(e) : 0; }();   // (tuning)
    const int lim = nmax ? nmax : (f32 ? 65536 : 49152);
    if (off || !(a.obs_mode == SNAC_OBS_ALL || a.obs_mode == SNAC_OBS_TILED) || pipeline_off()) return false;
    if (a.variant) {
        // the layout variants (k_rollout1dt<.., VAR>: blocks of four envs, one per CU): whole groups of four envs and 16-byte pieces;
        // it levels off at 1.2e10 env-steps/s with the 37-value PPO rows (the tile kernel: 8.6e9 at 65 536 envs) and at 4.4-5.1e10 with the
        // 8-value L-Net rows (the tile kernel: 3.1e10 at 65 536 envs) (profiles/r04_1d_layouts.txt)
        static const int vmax = [] { const char* e = std::getenv("SNAC_1D_TP_VAR_MAX"); return e ? std::atoi(e) : 65536; }();   // (tuning)
        return (a.n & 3) == 0 && (((uintptr_t)a.obs) & 15) == 0 && a.n <= vmax;
    }
    return a.n <= lim;
}

// SNAC_STEP_STAGE=0 keeps snac_step on k_transition2d / k_transition3d (A/B timing, tests of both paths)
bool step_stage_ok(const KArgs& a) {
    static const bool off = [] { const char* e = std::getenv("SNAC_STEP_STAGE"); return e && e[0] == '0'; }();
    return !off && !a.src_index && !a.dst_index && (a.n & 3) == 0 && ((uintptr_t)a.obs & 15) == 0;
}
// the layout variants on k_step2d<.., VAR>: 64 envs per wave are 64 rows of kilobytes per wave -- batches large enough to fill the
// CUs that way; below, the 8-env tiles of k_transition spread the rows over more waves.  PPO rows (451 values), us per tick at
// 40 960 / 49 152 / 65 536 envs: 36.7 / 36.8 / 37.7 against 35.4 / 40.8 / 88 (float32 rows at 32 768: 26.7 against 32.0); L-Net rows
// 24 576 / 32 768 / 65 536: 8.3 / 8.8 / 9.9 against 8.5 / 10.5 / 20.2 (profiles/r04_step_layouts.txt).  SNAC_STEP_VAR_MIN=n replaces
// the limits.
bool step_var_ok(const KArgs& a, bool f32) {
    static const int nmin = [] { const char* e = std::getenv("SNAC_STEP_VAR_MIN"); return e ? std::atoi(e) : 0; }();   // (tuning)
    if (nmin) return a.n >= nmin;
    if (!(a.tail & SNAC_TAIL_PLAN)) return a.n >= 24576;
    // rows with the plan tail: half-filled tiles for 24 577 .. 32 768 envs (one round of 1024 waves: 22.7 us at 32 768 envs against
    // k_transition's 28.8 and the full tiles' 36.6; float32 18.2 / 23.7 / 26.7), full tiles from 45 056 (float32: above 32 768)
    return (a.n > 24576 && a.n <= 32768) || a.n >= (f32 ? 32769 : 45056);
}
// half-filled tiles for rows with the plan tail up to 32 768 envs (SNAC_STEP_VAR_HALF=0 / 1 forces)
bool step_var_half(const KArgs& a) {
    static const int force = [] { const char* e = std::getenv("SNAC_STEP_VAR_HALF"); return e ? std::atoi(e) : -1; }();   // (tuning)
    if (force >= 0) return force != 0;
    return (a.tail & SNAC_TAIL_PLAN) && a.n <= 32768;
}
// 3D: k_step3d<.., VAR>.  PPO rows, us per tick at 16 384 / 65 536 envs: 37.8 / 42.8 against k_transition's 34.2 / 112.6 (float32 rows
// at 524 288 envs: 403 against 1032) (profiles/r04_step_layouts.txt).  SNAC_STEP_VAR3_MIN=n moves the limit.
bool step_var3_ok(const KArgs& a) {
    static const int nmin = [] { const char* e = std::getenv("SNAC_STEP_VAR3_MIN"); return e ? std::atoi(e) : 24576; }();   // (tuning)
    return a.n >= nmin && a.frame_val == -1;
}

int launch(Op op, const snac_env_desc* d, const KArgs& a, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const bool dyn = d->dynamic != 0;
    const int E = pick_tile(d->kind, a.n);
    const char* const tile_name = op == OP_ROLLOUT ? "k_rollout" : (op == OP_TRANSITION ? "k_transition" : "k_aux");
    g_kernel = tile_name;
    switch (d->kind) {
        case SNAC_ENV_1D:
            // rollouts that write every row: the time-parallel kernel while its rate beats the tile kernel's (lane-per-env transition)
            if (op == OP_ROLLOUT && roll1dt_ok(a, d->obs_dtype == SNAC_OBS_F32)) { g_kernel = "k_rollout1dt"; launch_roll1dt(d, a, s); break; }
            launch_tile1d(op, dyn, E, d->obs_dtype, a, s); break;
        case SNAC_ENV_2D:
            if (op == OP_TRANSITION && !pipeline_off() && step_stage_ok(a) && (!a.variant || step_var_ok(a, d->obs_dtype == SNAC_OBS_F32))) { g_kernel = "k_step2d"; launch_step2d(d, a, a.variant && step_var_half(a), s); break; }
            if (op == OP_TRANSITION && !a.variant && !pipeline_off()) { g_kernel = "k_transition2d"; launch_trans2d(d, a, s); break; }
            // float32 rows from N = 32 768: 512 staged waves (1.05 -> 0.74 ms per 600 ticks); float64 rows there are level (1.26-1.60 ms
            // by box for either kernel) and stay on 32-env tiles
            if (op == OP_ROLLOUT && roll2dt_ok(a, d->obs_dtype == SNAC_OBS_F32)) { g_kernel = "k_rollout2dt"; launch_roll2dt(d, a, s); break; }
            {
                // float64 rows from 32 769 envs (40 960: 1.56 against 1.99 ms per 600 ticks for the tile kernel's 32-env tiles, 49 152: 1.75 / 2.08,
                // 57 344: 2.05 / 2.26; at 32 768 and below the tile kernel's 1024 waves of 32 envs are ahead: 1.25 against 1.43), float32
                // rows from 32 768 (0.81 against 1.06); half-filled tiles -- 32 envs per wave on twice the waves -- were tried for
                // 24 576 .. 32 768 envs and lose on trajectory memory (profiles/r04_2d_midrange.txt).  SNAC_2D_STAGE_MIN=n moves the limit.
                static const int stage_min = [] { const char* e = std::getenv("SNAC_2D_STAGE_MIN"); return e ? std::atoi(e) : 0; }();
                const int from = stage_min ? stage_min : (d->obs_dtype == SNAC_OBS_F32 ? 32768 : 32769);
                if (op == OP_ROLLOUT && roll2d_ok(a, a.n >= from ? 64 : E)) { g_kernel = "k_rollout2d"; launch_roll2d(d, a, s); break; }
            }
            launch_tile2d(op, dyn, E, d->obs_dtype, a, s); break;
        default:   // 3D: 2.1 KB of LDS per env -> tiles of 16 (or 8 for small batches: two waves per SIMD sooner)
            if (op == OP_ROLLOUT && roll3db_ok(a, d->obs_dtype == SNAC_OBS_F32)) { g_kernel = "k_rollout3db"; launch_roll3db(d, a, s); break; }
            if (op == OP_ROLLOUT && E == 8 && !a.variant && (a.obs_mode == SNAC_OBS_ALL || a.obs_mode == SNAC_OBS_TILED) && a.num_plans <= TB_MAX && !pipeline_off()) { g_kernel = "k_rollout3d"; launch_roll3d(d, a, s); break; }
            if (op == OP_TRANSITION && !pipeline_off() && step_stage_ok(a) && (!a.variant || step_var3_ok(a))) { g_kernel = "k_step3d"; launch_step3d(d, a, s); break; }
            if (op == OP_TRANSITION && !a.variant && !pipeline_off()) { g_kernel = "k_transition3d"; launch_trans3d(d, a, s); break; }
            launch_tile3d(op, dyn, E, d->obs_dtype, a, s);
            break;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "kernel launch");
    return SNAC_OK;
}

}  // namespace snac_detail
using namespace snac_detail;

// ------------------------------------------------------------------------------------------------
extern "C" {

int snac_version(void) { return SNAC_ABI_VERSION; }

const char* snac_last_error(void) { return g_err; }

const char* snac_last_kernel(void) { return g_kernel; }

int snac_stream_sync(void* stream) {
    const hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    return e == hipSuccess ? SNAC_OK : fail_hip(e, "hipStreamSynchronize");
}

int snac_env_sizes(int kind, int dynamic, snac_sizes* o) {
    if (!o) return fail(SNAC_ERR_ARG, "null out");
    std::memset(o, 0, sizeof(*o));
    if (kind == SNAC_ENV_1D) {
        *o = snac_sizes{7, 3, 750, 2, 1, 34, 1, 30, 32, 2, 32, 2};
    } else if (kind == SNAC_ENV_2D) {
        *o = snac_sizes{51, 5, 600, 3, 26, 26, 20, 20, 20, 4, 20, 4};
    } else if (kind == SNAC_ENV_3D) {
        *o = snac_sizes{51, 8, dynamic ? 1000 : 1300, 3, 26, 26, 20, 20, 400, 2, 400, 2};
    } else {
        return fail(SNAC_ERR_ARG, "unknown env kind");
    }
    return SNAC_OK;
}

int snac_obs_dim(const snac_env_desc* d) {
    if (!d) return fail(SNAC_ERR_ARG, "null desc");
    if (d->kind < SNAC_ENV_1D || d->kind > SNAC_ENV_3D) return fail(SNAC_ERR_ARG, "unknown env kind");
    if (int rc = check_layout(d)) return rc;
    return base_obs_dim(d->kind) + tail_len(d->kind, d->obs_tail);
}

int snac_reset_scalar(const snac_env_desc* d, const snac_state* st, int32_t plan_idx, void* obs, void* stream) {
    if (int rc = check_common(d, st)) return rc;
    if (plan_idx < 0 || plan_idx >= d->num_plans) return fail(SNAC_ERR_ARG, "plan_idx out of range");
    KArgs a = make_args(d, st);
    a.aux_op = AUX_RESET; a.plan_scalar = plan_idx; a.obs = obs;
    return launch(OP_AUX, d, a, stream);
}

int snac_step_scalar(const snac_env_desc* d, const snac_state* st, uint32_t t, int32_t action, int32_t step_size, int auto_reset,
                     void* obs, float* reward, uint8_t* done, void* stream) {
    if (int rc = check_common(d, st)) return rc;
    KArgs a = make_args(d, st);
    a.pool = d->num_envs; a.stats_on = 1;
    a.T = 1; a.t0 = t; a.auto_reset = auto_reset ? 1 : 0; a.obs_mode = obs ? SNAC_OBS_ALL : SNAC_OBS_NONE;
    a.use_scalar = 1; a.act_scalar = action; a.k_scalar = step_size;
    a.obs = obs; a.reward = reward; a.done = done;
    return launch(OP_TRANSITION, d, a, stream);
}

int snac_reset(const snac_env_desc* d, const snac_state* st, const uint8_t* mask, const int16_t* plan_idx_in, void* obs,
               void* stream) {
    if (int rc = check_common(d, st)) return rc;
    KArgs a = make_args(d, st);
    a.aux_op = AUX_RESET; a.mask = mask; a.plan_idx_in = plan_idx_in; a.obs = obs;
    return launch(OP_AUX, d, a, stream);
}

int snac_rollout_rec(const snac_env_desc* d, const snac_state* st, int32_t T, uint32_t t0, const int8_t* actions,
                     const int8_t* step_size, int obs_mode, void* obs, float* reward, uint8_t* done,
                     const snac_rollout_record* rec, void* stream) {
    if (int rc = check_common(d, st)) return rc;
    if (T < 0) return fail(SNAC_ERR_ARG, "T must be >= 0");
    if (obs_mode < SNAC_OBS_NONE || obs_mode > SNAC_OBS_TILED) return fail(SNAC_ERR_ARG, "unknown obs_mode");
    if (obs_mode != SNAC_OBS_NONE && !obs) return fail(SNAC_ERR_ARG, "obs_mode set but obs is null");
    if (T == 0) return SNAC_OK;
    KArgs a = make_args(d, st);
    a.T = T; a.t0 = t0; a.auto_reset = 1; a.obs_mode = obs_mode;
    a.tiled_T = T; a.tiled_t0 = 0;
    a.actions = actions; a.step_size = step_size; a.obs = obs; a.reward = reward; a.done = done;
    if (rec) {
        a.actions_out = rec->actions; a.step_size_out = rec->step_size; a.plan_idx_out = rec->plan_idx; a.first_out = rec->first;
    }
    return launch(OP_ROLLOUT, d, a, stream);
}

int snac_rollout_tiled(const snac_env_desc* d, const snac_state* st, int32_t T, uint32_t t0, const int8_t* actions,
                       const int8_t* step_size, int32_t ring_ticks, int32_t first_tick, void* obs, float* reward, uint8_t* done,
                       const snac_rollout_record* rec, void* stream) {
    if (int rc = check_common(d, st)) return rc;
    if (T < 0 || ring_ticks < 1 || first_tick < 0 || (long long)first_tick + T > ring_ticks) return fail(SNAC_ERR_ARG, "steps outside the ring");
    if (!obs) return fail(SNAC_ERR_ARG, "obs is null");
    if (T == 0) return SNAC_OK;
    KArgs a = make_args(d, st);
    a.T = T; a.t0 = t0; a.auto_reset = 1; a.obs_mode = SNAC_OBS_TILED;
    a.tiled_T = ring_ticks; a.tiled_t0 = first_tick;
    a.actions = actions; a.step_size = step_size; a.obs = obs; a.reward = reward; a.done = done;
    if (rec) {
        a.actions_out = rec->actions; a.step_size_out = rec->step_size; a.plan_idx_out = rec->plan_idx; a.first_out = rec->first;
    }
    return launch(OP_ROLLOUT, d, a, stream);
}

int snac_rollout(const snac_env_desc* d, const snac_state* st, int32_t T, uint32_t t0, const int8_t* actions,
                 const int8_t* step_size, int obs_mode, void* obs, float* reward, uint8_t* done, void* stream) {
    return snac_rollout_rec(d, st, T, t0, actions, step_size, obs_mode, obs, reward, done, nullptr, stream);
}


int snac_step(const snac_env_desc* d, const snac_state* st, uint32_t t, const int8_t* actions, const int8_t* step_size,
              int auto_reset, void* obs, float* reward, uint8_t* done, void* stream) {
    if (int rc = check_common(d, st)) return rc;
    KArgs a = make_args(d, st);
    a.pool = d->num_envs; a.stats_on = 1;                        // the single-step kernel on the identity rows
    a.T = 1; a.t0 = t; a.auto_reset = auto_reset ? 1 : 0; a.obs_mode = obs ? SNAC_OBS_ALL : SNAC_OBS_NONE;
    a.actions = actions; a.step_size = step_size; a.obs = obs; a.reward = reward; a.done = done;
    return launch(OP_TRANSITION, d, a, stream);
}

int snac_transition(const snac_env_desc* d, const snac_state* st, int32_t m, const int32_t* src_index, const int32_t* dst_index,
                    uint32_t t, const int8_t* actions, const int8_t* step_size, void* obs, float* reward, uint8_t* done,
                    void* stream) {
    if (int rc = check_common(d, st)) return rc;
    if (m < 0) return fail(SNAC_ERR_ARG, "m must be >= 0");
    // an absent index array means "row i": with either one absent, edge i touches pool row i, so m is bounded by the pool
    if ((!src_index || !dst_index) && m > d->num_envs) return fail(SNAC_ERR_ARG, "m exceeds the pool (num_envs)");
    if (m == 0) return SNAC_OK;
    KArgs a = make_args(d, st);
    a.pool = d->num_envs; a.n = m; a.src_index = src_index; a.dst_index = dst_index;
    a.T = 1; a.t0 = t; a.actions = actions; a.step_size = step_size; a.obs = obs; a.reward = reward; a.done = done;
    return launch(OP_TRANSITION, d, a, stream);
}

int snac_observe(const snac_env_desc* d, const snac_state* st, void* obs, void* stream) {
    if (int rc = check_common(d, st)) return rc;
    if (!obs) return fail(SNAC_ERR_ARG, "null obs");
    KArgs a = make_args(d, st);
    a.aux_op = AUX_OBSERVE; a.obs = obs;
    return launch(OP_AUX, d, a, stream);
}

int snac_iou(const snac_env_desc* d, const snac_state* st, double* out, void* stream) {
    if (int rc = check_common(d, st)) return rc;
    if (!out) return fail(SNAC_ERR_ARG, "null out");
    KArgs a = make_args(d, st);
    a.aux_op = AUX_IOU; a.out_f64 = out;
    return launch(OP_AUX, d, a, stream);
}


}  // extern "C"
