// snac_hip.hip -- the C ABI of include/snac_hip.h and the dispatch: which kernel a call runs on (launch()).  The kernels live in the
// k_*.hip translation units beside this one (snac_dev.h has the map), trajectory memory in snac_traj.hip.
#include "snac_dev.h"

namespace snac_detail {
void launch_step3dq(const snac_env_desc* d, const KArgs& a, hipStream_t s);                   // k_step3dq.hip
void launch_step1d(const snac_env_desc* d, const KArgs& a, hipStream_t s);                    // k_step1d.hip
void launch_edges1d(const snac_env_desc* d, const KArgs& a, hipStream_t s);                   // k_step1d.hip
void launch_reset(const snac_env_desc* d, const KArgs& a, hipStream_t s);                     // k_reset.hip
void launch_iou(const snac_env_desc* d, const KArgs& a, hipStream_t s);                       // k_reset.hip
void launch_aux1d(const snac_env_desc* d, const KArgs& a, hipStream_t s);                     // k_step1d.hip
void launch_aux2d(const snac_env_desc* d, const KArgs& a, hipStream_t s);                     // k_step.hip
void launch_aux3d(const snac_env_desc* d, const KArgs& a, hipStream_t s);                     // k_step3dq.hip
void launch_roll2db(const snac_env_desc* d, const KArgs& a, int steppers, hipStream_t s);   // k_roll2db.hip
void launch_roll2dbv(const snac_env_desc* d, const KArgs& a, int steppers, hipStream_t s);  // k_roll2dbv.hip: the same kernel for the layout variants: k_rollout2db (declared here for the same reason as the next one)
void launch_roll3dbv(const snac_env_desc* d, const KArgs& a, hipStream_t s);   // k_roll3dbv.hip: k_rollout3db for the layout variants (declared here: snac_dev.h is hashed into profiles/traffic.json)
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


// ------------------------------------------------------------------------------------------------
// THE DISPATCH TABLE: every batch-size threshold and switch that decides which kernel a call runs on, in one place.  Each entry is
// (environment variable, default, what it decides); the defaults are what was measured on the MI355X boxes of the build pool
// (profiles/ files named per entry).  Overrides are read once per process.  snac_tuning() prints the effective values;
// tools/retune.py re-measures the crossovers on the box it runs on and prints the overrides that would move them.
struct Knob { const char* env; int def; const char* what; };
const Knob KNOBS[TN_COUNT] = {
    /* TN_PIPELINE        */ {"SNAC_3D_PIPELINE", 1, "0: every launch on the generic tile kernels (k_rollout / k_transition / k_aux)"},
    /* TN_TILE            */ {"SNAC_TILE", 0, "envs per wave of the tile kernels (8 / 16 / 32 / 64; 0: by batch size, pick_tile)"},
    /* TN_2D_TILE32_MIN   */ {"SNAC_2D_TILE32_MIN", 22528, "2D rollouts on the tile kernel take 32 envs per wave from (24 576 envs: 1.05 against 1.12 ms per 600 ticks with 16; 28 672: 1.06 / 1.21; 20 480: 1.05 / 0.97; r05_midrange.txt)"},
    /* TN_3D_BLOCK        */ {"SNAC_3D_BLOCK", 1, "0: 3D rollouts stay on k_rollout3d instead of the block kernel k_rollout3db"},
    /* TN_3D_BLOCK_MIN    */ {"SNAC_3D_BLOCK_MIN", -1, "k_rollout3db from this many envs (-1: the two defaults below)"},
    /* TN_3D_BLOCK_MIN_F64*/ {"SNAC_3D_BLOCK_MIN_F64", 4096, "k_rollout3db, float64 rows, from (4096 envs: 0.982 against k_rollout3d's 1.031 ms per 1000 ticks, 2048: 0.986 / 0.974; r05_retune.txt)"},
    /* TN_3D_BLOCK_MIN_F32*/ {"SNAC_3D_BLOCK_MIN_F32", 4096, "k_rollout3db, float32 rows, from (4096 envs: 1.02 against 0.97 ms)"},
    /* TN_2D_STAGE        */ {"SNAC_2D_STAGE", 1, "0: 2D rollouts stay on the tile kernel instead of k_rollout2d"},
    /* TN_2D_STAGE_MIN    */ {"SNAC_2D_STAGE_MIN", 0, "k_rollout2d from this many envs (0: the two defaults below)"},
    /* TN_2D_STAGE_MIN_F64*/ {"SNAC_2D_STAGE_MIN_F64", 32769, "k_rollout2d, float64 rows, from (40 960 envs: 1.56 against 1.99 ms; 32 768: 1.43 against 1.25; r04_2d_midrange.txt)"},
    /* TN_2D_STAGE_MIN_F32*/ {"SNAC_2D_STAGE_MIN_F32", 32768, "k_rollout2d, float32 rows, from (32 768 envs: 0.81 against 1.06 ms)"},
    /* TN_2D_TP           */ {"SNAC_2D_TP", 1, "0: 2D rollouts stay off the time-parallel k_rollout2dt"},
    /* TN_2D_TP_MAX       */ {"SNAC_2D_TP_MAX", 0, "k_rollout2dt up to this many envs (0: the defaults below)"},
    /* TN_2D_TP_MAX_F64   */ {"SNAC_2D_TP_MAX_F64", 19456, "k_rollout2dt, float64 rows, up to (18 432 envs: 5.8 against 5.2 TB/s; r04_2d_midrange.txt part 3)"},
    /* TN_2D_TP_GAP_LO    */ {"SNAC_2D_TP_GAP_LO", 15873, "... except from here ..."},
    /* TN_2D_TP_GAP_HI    */ {"SNAC_2D_TP_GAP_HI", 16384, "... to here, where the tile kernel's 256 waves fill the chip exactly (5.8 against 5.55 TB/s)"},
    /* TN_2D_TP_MAX_F32   */ {"SNAC_2D_TP_MAX_F32", 30719, "k_rollout2dt, float32 rows, up to (16 384 envs: 4.2 against 2.9 TB/s)"},
    /* TN_2D_TP_MAX_ODD   */ {"SNAC_2D_TP_MAX_ODD", 8192, "k_rollout2dt for batches whose per-tick runs are not 16-byte pieces, up to"},
    /* TN_2D_TP_VAR_MAX   */ {"SNAC_2D_TP_VAR_MAX", 0, "k_rollout2dt<VAR> (layout variants) up to this many envs (0: the two defaults below)"},
    /* TN_2D_TP_VAR_PLAN  */ {"SNAC_2D_TP_VAR_MAX_PLAN", 49152, "rows with the plan tail (451 values) on k_rollout2dt<VAR> up to (r04_2d_layouts.txt part 3)"},
    /* TN_2D_TP_VAR_SHORT */ {"SNAC_2D_TP_VAR_MAX_SHORT", 6144, "short variant rows (53-61 values) on k_rollout2dt<VAR> up to"},
    /* TN_2D_TP_EB8       */ {"SNAC_2D_TP_EB8", 1025, "k_rollout2dt blocks hold 8 envs instead of 4 from (1536 envs: 0.086 against 0.097 ms)"},
    /* TN_1D_TP           */ {"SNAC_1D_TP", 1, "0: 1D rollouts stay on the tile kernel instead of k_rollout1dt"},
    /* TN_1D_TP_MAX       */ {"SNAC_1D_TP_MAX", 0, "k_rollout1dt up to this many envs (0: the two defaults below)"},
    /* TN_1D_TP_MAX_F64   */ {"SNAC_1D_TP_MAX_F64", 65536, "k_rollout1dt, float64 rows, up to (round 6, 8-env blocks: 65 536 envs 0.685 against the tile kernel's 0.720 ms per 750 ticks; 131 072: 1.68 against 1.44; r06_1d.txt)"},
    /* TN_1D_TP_MAX_F32   */ {"SNAC_1D_TP_MAX_F32", 65536, "k_rollout1dt, float32 rows, up to (65 536 envs: 0.65 against 0.73 ms)"},
    /* TN_1D_TP_VAR_MAX   */ {"SNAC_1D_TP_VAR_MAX", 65536, "k_rollout1dt<VAR> (layout variants) up to (r04_1d_layouts.txt)"},
    /* TN_1D_TP_EB16      */ {"SNAC_1D_TP_EB16", 3584, "k_rollout1dt blocks hold 16 envs instead of 4 from (3072 envs: 0.048 against 0.041 ms; 3584: level)"},
    /* TN_STEP_STAGE      */ {"SNAC_STEP_STAGE", 1, "0: snac_step stays on k_transition2d / 3d instead of k_step2d / 3d"},
    /* TN_STEP_VAR_MIN    */ {"SNAC_STEP_VAR_MIN", 0, "2D steps with a layout variant on k_step2d<VAR> from this many envs (0: the defaults below)"},
    /* TN_STEP_VAR_SHORT  */ {"SNAC_STEP_VAR_MIN_SHORT", 24576, "short variant rows on k_step2d<VAR> from (24 576 envs: 8.3 against 8.5 us per tick; r04_step_layouts.txt)"},
    /* TN_STEP_VAR_HALF_LO*/ {"SNAC_STEP_VAR_HALF_LO", 24577, "rows with the plan tail on half-filled tiles of k_step2d<VAR> from ..."},
    /* TN_STEP_VAR_HALF_HI*/ {"SNAC_STEP_VAR_HALF_HI", 32768, "... to (32 768 envs: 22.7 us against k_transition's 28.8 and the full tiles' 36.6)"},
    /* TN_STEP_VAR_F64    */ {"SNAC_STEP_VAR_FULL_F64", 45056, "rows with the plan tail, float64, on full tiles of k_step2d<VAR> from (49 152 envs: 36.8 against 40.8 us)"},
    /* TN_STEP_VAR_F32    */ {"SNAC_STEP_VAR_FULL_F32", 32769, "the same, float32 rows (32 768 envs: 26.7 against 32.0 us)"},
    /* TN_STEP_VAR_HALF   */ {"SNAC_STEP_VAR_HALF", -1, "0 / 1: never / always half-filled tiles for rows with the plan tail (-1: the range above)"},
    /* TN_STEP_VAR3_MIN   */ {"SNAC_STEP_VAR3_MIN", 24576, "3D steps with a layout variant on k_step3d<VAR> from (65 536 envs: 42.8 against 112.6 us)"},
    /* TN_STEP3D_SPAN     */ {"SNAC_STEP3D_SPAN", 1, "0: the canonical 3D snac_step stays on k_step3d (seven row loads per lane) instead of k_step3ds (cooperative span loads)"},
    /* TN_STEP3D_SPAN_MIN */ {"SNAC_STEP3D_SPAN_MIN", 81920, "k_step3ds from this many envs (98 304: 21.0 against k_step3d's 25.1 us per tick, 262 144: 41.7 / 53.3, 524 288: 78.6 / 95.4-101; 65 536: 15.9 / 14.1; r05_step_experiments.txt part 6)"},
    /* TN_T2D_E           */ {"SNAC_T2D_E", 0, "edges per wave of k_transition2d (16 / 32 / 64; 0: 32 from 65 536 edges, else 16)"},
    /* TN_EDGES3D         */ {"SNAC_EDGES3D", 1, "0: 3D tree edges with gathered rows stay on k_transition3d instead of k_edges3d"},
    /* TN_EDGES2D         */ {"SNAC_EDGES2D", 1, "0: 2D tree edges with gathered rows stay on k_transition2d instead of k_edges2d (records through LDS)"},
    /* TN_EDGES2D_MIN     */ {"SNAC_EDGES2D_MIN", 4, "k_edges2d from this many edges per call"},
    /* TN_3D_BLOCK_VAR    */ {"SNAC_3D_BLOCK_VAR", 1, "0: 3D rollouts with a layout variant stay on the tile kernel instead of k_rollout3db's variant forms"},
    /* TN_3D_BLOCK_VAR_MIN*/ {"SNAC_3D_BLOCK_VAR_MIN", 64, "variant rows without the plan tail (51-61 values) on k_rollout3db from (1.22 us per tick at any N against the tile kernel's 3.2; r05_var3d.txt)"},
    /* ..VAR_PLAN_F64     */ {"SNAC_3D_BLOCK_VAR_PLAN_F64", 10240, "rows with the plan tail, float64, on k_rollout3db from (10 240 envs x 200 ticks: 1.06 against 1.33 ms, 8192: 0.96 / 0.94, 16 384: 1.74 / 1.99; r05_var3d.txt)"},
    /* ..VAR_PLAN_F32     */ {"SNAC_3D_BLOCK_VAR_PLAN_F32", 16384, "the same, float32 rows (16 384 envs: 0.93 against 1.00 ms, 14 336: 0.89 / 0.86)"},
    /* TN_2D_BLOCK        */ {"SNAC_2D_BLOCK", 1, "0: no 2D rollout takes the block kernel k_rollout2db (stepper waves + eight writer waves per 64 / 128 envs)"},
    /* TN_2D_BLOCK_MIN_F64*/ {"SNAC_2D_BLOCK_MIN_F64", 11264, "k_rollout2db, float64 canonical rows, from this many envs (12 288: 0.443 against k_rollout2dt's 0.531 ms per 600 ticks, 10 240: 0.447 / 0.433; r05_2d_block.txt) ..."},
    /* TN_2D_BLOCK_MAX_F64*/ {"SNAC_2D_BLOCK_MAX_F64", 32768, "... up to this many (32 768: 1.194 against the tile kernel's 1.228 ms, 40 960: 1.71 against k_rollout2d's 1.54)"},
    /* TN_2D_BLOCK_MIN_F32*/ {"SNAC_2D_BLOCK_MIN_F32", 15360, "k_rollout2db, float32 canonical rows, from (16 384: 0.434 against k_rollout2dt's 0.462 ms, 12 288: 0.432 / 0.340) ..."},
    /* TN_2D_BLOCK_MAX_F32*/ {"SNAC_2D_BLOCK_MAX_F32", 32768, "... up to (32 768: 0.607 against k_rollout2d's 0.820 ms, 36 864: 1.08 / 0.82)"},
    /* TN_2D_BLOCK_TWO_F64*/ {"SNAC_2D_BLOCK_TWO_F64", 16384, "k_rollout2db with blocks of 128 envs (two stepper waves) from this many envs, float64 rows (16 384: 0.570 against 0.598 ms on 64-env blocks; 20 480: 0.746 / 0.948)"},
    /* TN_2D_BLOCK_TWO_F32*/ {"SNAC_2D_BLOCK_TWO_F32", 16385, "the same, float32 rows (16 384: 0.502 against 0.434; 20 480: 0.504 / 0.849)"},
    /* TN_2D_BLOCK_VAR_MIN*/ {"SNAC_2D_BLOCK_VAR_MIN", 6148, "2D rollouts with a layout variant without the plan tail (51-61 values) on k_rollout2db from this many envs (up to 6144: k_rollout2dt; 59-value rows at 6144 envs 0.56 against 0.63 ms, 4096: 0.55 / 0.42; r05_2d_block.txt) ..."},
    /* TN_2D_BLOCK_VAR_MAX*/ {"SNAC_2D_BLOCK_VAR_MAX", 32768, "... up to this many (32 768: 1.36 against the tile kernel's 5.92 ms; above: k_rollout2d, 49 152: 2.47 / 2.54, 65 536: 2.70 / 2.56)"},
    /* TN_2D_BLOCK_VAR_TWO*/ {"SNAC_2D_BLOCK_VAR_TWO", 16385, "the variant rows on blocks of 128 envs from this many envs (16 384 envs, L-Net rows: 0.92 ms on 128 blocks of 128 against the tile kernel's 0.82)"},
    /* TN_2D_BLOCK_FOUR_F64*/ {"SNAC_2D_BLOCK_FOUR_MAX_F64", 38912, "canonical float64 rows above SNAC_2D_BLOCK_MAX_F64 and up to this many envs: k_rollout2db with blocks of 256 envs (four stepper waves; 34 816 envs: 1.35 against k_rollout2d's 1.53 ms, 36 864: 1.46 / 1.53, 40 960: 1.55 / 1.54; 0: never)"},
    /* TN_2D_BLOCK_FOUR_F32*/ {"SNAC_2D_BLOCK_FOUR_MAX_F32", 45056, "the same, float32 rows (36 864 envs: 0.75 against 0.82 ms, 40 960: 0.77 / 0.83, 49 152: 0.91 / 0.88)"},
    /* TN_STEP3D_QUARTER  */ {"SNAC_STEP3D_QUARTER", 1, "0: the canonical 3D snac_step stays on k_step3d / k_step3ds instead of k_step3dq (16 envs per wave, four lanes per env)"},
    /* TN_STEP3D_QUARTER_MIN*/ {"SNAC_STEP3D_QUARTER_MIN", 4, "k_step3dq from this many envs (1024: 6.9 against k_step3d's 8.0 us per tick, 32 768: 9.5 / 11.5, 65 536: 13.1 / 14.1; r05_step3dq.txt) ..."},
    /* TN_STEP3D_QUARTER_MAX*/ {"SNAC_STEP3D_QUARTER_MAX", 1 << 30, "... up to this many (524 288: 67.5 against k_step3ds' 76.6 us, 131 072: 23.9 / 23.6)"},
    /* TN_1D_TP_EB8_MIN   */ {"SNAC_1D_TP_EB8_MIN", 3584, "k_rollout1dt blocks hold 8 envs from this many envs (two blocks share a CU: one computes its chunk while the other's rows leave; 4096 envs: 0.045 against 0.048 ms on 16-env blocks, 32 768: 0.351 / 0.421, 65 536: 0.685 / 0.920; r06_1d.txt) ..."},
    /* TN_1D_TP_EB8_MAX   */ {"SNAC_1D_TP_EB8_MAX", 1 << 30, "... up to this many"},
    /* TN_STEP3D_NTLOAD_MIN */ {"SNAC_STEP3D_NTLOAD_MIN", 376832, "k_step3dq from this many envs on: non-temporal span loads + plain row stores; below, where state and rows fit the Infinity Cache: plain span loads (the state stays cached) + non-temporal row stores (65 536 envs: 12.9 -> 10.5 us per tick, 131 072: 24.1 -> 19.1; 262 144: 39.6 against 41.3 / 43.5 the other way, 524 288: 71 against 101; r06_step_loads.txt)"},
    /* TN_STEP3D_HUGE_MIN */ {"SNAC_STEP3D_HUGE_MIN", 557056, "k_step3dq from this many envs on (the rows of one tick alone overflow the Infinity Cache): the form SNAC_STEP3D_HUGE_FORM"},
    /* TN_STEP3D_HUGE_FORM*/ {"SNAC_STEP3D_HUGE_FORM", 3, "bit 0: non-temporal span loads, bit 1: non-temporal row stores (655 360 envs: 135 us per tick with 1, 109 with 2, 99 with 3; 1 048 576: 237 / 160 / 157)"},
    /* TN_STEP3D_FORM     */ {"SNAC_STEP3D_FORM", -1, "k_step3dq: force a form (bits as above) whatever the batch size; -1: by batch size"},
    /* TN_STEP2D_PLAIN_LO */ {"SNAC_STEP2D_PLAIN_LO", 20480, "k_step2d reads its records with PLAIN loads from this many envs (below: non-temporal loads, plain rows: 16 384 envs 6.5 against 6.8 us) ..."},
    /* TN_STEP2D_RES_HI   */ {"SNAC_STEP2D_RES_HI", 278528, "k_step2d inside SNAC_STEP2D_PLAIN_LO .. this many envs also stores its rows NON-TEMPORALLY (the resident form: 32 768 envs 7.9 -> 6.8 us per tick, 65 536: 8.4 -> 7.7-7.8, 262 144: 22.6 -> 22.3; 294 912: 24.7 against 24.9, 458 752: 37.0 against 39.7)"},
    /* TN_STEP2D_PLAIN_HI */ {"SNAC_STEP2D_PLAIN_HI", 475136, "... up to this many, with non-temporal loads outside (81 920 envs: 10.6 -> 10.4 us, 262 144: 26.5 -> 22.7, 458 752: 41.5 -> 37.0; 65 536: 8.4 against 8.75 plain, 524 288: 46.5 against 49.4)"},
    /* TN_STEP2D_HUGE_MIN */ {"SNAC_STEP2D_HUGE_MIN", 475137, "k_step2d from this many envs on: the form SNAC_STEP2D_HUGE_FORM (the 2D state of a million envs is 105 MB: it fits the Infinity Cache at every batch size, the rows of a tick do not)"},
    /* TN_STEP2D_HUGE_FORM*/ {"SNAC_STEP2D_HUGE_FORM", 2, "bit 0: non-temporal record loads, bit 1: non-temporal row stores (524 288 envs: 46.4 us per tick with 1, 43.5 with 2; 1 048 576: 146 / 82; r06_step_loads.txt)"},
    /* TN_STEP2D_FORM     */ {"SNAC_STEP2D_FORM", -1, "k_step2d, canonical rows: force a form (bits as above) whatever the batch size; -1: by batch size"},
    /* TN_NODES2D_NT      */ {"SNAC_NODES2D_NT", 1, "k_edges2dp (2D edges on node records): 1 = the observation rows leave as NON-TEMPORAL stores -- streamed rows then do not displace the node records in the Infinity Cache (65.1 -> 50.3 us per 524 288 edges of a 2^20-record pool, r06_edges.txt)"},
    /* TN_1D_LANE         */ {"SNAC_1D_LANE", 1, "1D rollouts of large batches on k_rollout1dl (lane = env; canonical rows, every row written, N % 4 == 0, aligned obs) ..."},
    /* TN_1D_LANE_MIN_F64 */ {"SNAC_1D_LANE_MIN_F64", 45056, "... float64 rows from this many envs (a wave of 64 envs per SIMD at 65 536: 0.454 ms per 750 ticks = 6.6 TB/s against 0.69-0.73 on the time-parallel kernel; 45 056: 7.5e10 env-steps/s against 6.7e10, 40 960: 6.7 against 6.9; r06_1d_lane.txt) ..."},
    /* TN_1D_LANE_MIN_F32 */ {"SNAC_1D_LANE_MIN_F32", 36864, "... float32 rows from this many (the lane kernel: 0.37 ms per 750 ticks whatever the batch up to 65 536 envs; the time-parallel one by box: 36 864 envs 0.35-0.42 ms, 40 960 0.39-0.49, 32 768 0.31-0.39; r06_1d_lane.txt, r06_retune_1d.txt)"},
    /* TN_1D_LANE_NT      */ {"SNAC_1D_LANE_NT", 0, "k_rollout1dl: 1 = its rows leave as non-temporal stores (65 536 envs: 0.468 against 0.454 ms with float64 rows, level with float32: off)"},
    /* TN_STEP1D          */ {"SNAC_STEP1D", 1, "the canonical 1D snac_step on identity rows (N % 4 == 0, aligned obs) on k_step1d: 64 envs per wave, wide loads, the rows as one run per wave (0 = the tile kernel k_transition) ..."},
    /* TN_STEP1D_MIN      */ {"SNAC_STEP1D_MIN", 256, "... from this many envs"},
    /* TN_STEP1D_FORM     */ {"SNAC_STEP1D_FORM", 2, "k_step1d: bit 0 = its records by non-temporal loads, bit 1 = its rows by non-temporal stores.  Records and rows of every batch size fit the Infinity Cache: the resident form (2) at 524 288 envs 15.5 us per tick, both plain 16.2, non-temporal loads 18.4-18.9 (r06_step1d.txt)"},
    /* TN_EDGES1D         */ {"SNAC_EDGES1D", 1, "1D tree edges with gathered rows (snac_transition with index arrays, canonical layout) on k_edges1d: the records through LDS, four lanes per record (0 = the tile kernel k_transition) ..."},
    /* TN_EDGES1D_MIN     */ {"SNAC_EDGES1D_MIN", 64, "... from this many edges per call"},
    /* TN_STEP1D_VAR_MIN  */ {"SNAC_STEP1D_VAR_MIN", 256, "k_step1d takes the 1D layout variants (rows of 8 .. 46 values) from this many envs (1024 envs with the 37-value PPO rows: 5.4 against 6.5 us per tick on the tile kernel, 65 536: 6.9 / 23.0, 524 288 with 8-value L-Net rows: 18.3 / 44.1; r06_step1d.txt)"},
    /* TN_1D_LANE_VAR_MIN */ {"SNAC_1D_LANE_VAR_MIN", 28672, "k_rollout1dl takes the 1D layout variants with rows of more than 16 values from this many envs (37-value PPO rows: 1.75 ms per 750 ticks up to 32 768 envs, 2.19 at 65 536 = 6.8 TB/s; k_rollout1dt's VAR form 1.01 / 2.07 / 4.21 ms at 16 384 / 32 768 / 65 536; r06_1d_lane.txt) ..."},
    /* TN_1D_LANE_VAR_SHORT_MIN */ {"SNAC_1D_LANE_VAR_SHORT_MIN", 49152, "... and those with rows of at most 16 values from this many (8-value L-Net rows: 0.78 ms flat; k_rollout1dt 0.76 at 45 056, 1.05 at 65 536)"},
    /* TN_RESET_FAST      */ {"SNAC_RESET_FAST", 1, "snac_reset / snac_reset_scalar of every env (no mask), canonical layout, N % 4 == 0, aligned obs, on k_reset: nothing of the old state is read but the episode counter; records zeroed and rows written as runs (0 = the tile kernel k_aux) ..."},
    /* TN_RESET_FAST_MIN  */ {"SNAC_RESET_FAST_MIN", 256, "... from this many envs"},
    /* TN_IOU_FAST        */ {"SNAC_IOU_FAST", 1, "snac_iou on k_iou (lane = env, no LDS; 3D: the header alone) from 256 envs; 0 = the tile kernel k_aux, which loads every record into LDS first"},
    /* TN_AUX_STEP        */ {"SNAC_AUX_STEP", 1, "snac_reset with a mask and snac_observe (canonical layout, N % 4 == 0, aligned obs, from 256 envs) on the step kernels' AUX forms (k_step1d / k_step2d / k_step3dq without a step); 0 = the tile kernel k_aux"},
};

int tune(int id) {
    static int v[TN_COUNT];
    static std::once_flag once;
    std::call_once(once, [] {
        for (int i = 0; i < TN_COUNT; ++i) { const char* e = std::getenv(KNOBS[i].env); v[i] = e ? std::atoi(e) : KNOBS[i].def; }
    });
    return v[id];
}

// tile size of the tile kernels: enough tiles to give every SIMD of the 256 CUs a few waves
int pick_tile(int kind, int n) {
    const int forced = tune(TN_TILE);
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

bool pipeline_off() { return tune(TN_PIPELINE) == 0; }
inline bool every_row(const KArgs& a) { return a.obs_mode == SNAC_OBS_ALL || a.obs_mode == SNAC_OBS_TILED; }
inline bool pieces16(const KArgs& a) { return (a.n & 3) == 0 && (((uintptr_t)a.obs) & 15) == 0; }

// 3D rollouts by blocks of 64 envs (k_rollout3db): every row written, <= TB_MAX plans, 16-byte pieces; the layout variants on its VAR form
bool roll3db_ok(const KArgs& a, bool f32) {
    const int lim = tune(TN_3D_BLOCK_MIN) >= 0 ? tune(TN_3D_BLOCK_MIN) : tune(f32 ? TN_3D_BLOCK_MIN_F32 : TN_3D_BLOCK_MIN_F64);
    if (a.variant) {   // k_rollout3db<.., VAR>: its writers copy plan rows 16 bytes at a time
        const int from = (a.tail & SNAC_TAIL_PLAN) ? tune(f32 ? TN_3D_BLOCK_VAR_PLAN_F32 : TN_3D_BLOCK_VAR_PLAN_F64) : tune(TN_3D_BLOCK_VAR_MIN);
        if (tune(TN_3D_BLOCK_VAR) == 0 || a.n < from || (((uintptr_t)a.plans) & 15) != 0) return false;
        return tune(TN_3D_BLOCK) != 0 && every_row(a) && a.num_plans <= TB_MAX && pieces16(a) && !pipeline_off();
    }
    return tune(TN_3D_BLOCK) != 0 && a.n >= lim && every_row(a) && a.num_plans <= TB_MAX && pieces16(a) && !pipeline_off();
}

// 2D rollouts by blocks of 64 envs (k_rollout2db): the middle batches, every row written, canonical layout, 16-byte pieces
bool roll2db_ok(const KArgs& a, bool f32) {
    if (tune(TN_2D_BLOCK) == 0 || !every_row(a) || !pieces16(a) || pipeline_off()) return false;
    if (a.variant) return !(a.tail & SNAC_TAIL_PLAN) && a.n >= tune(TN_2D_BLOCK_VAR_MIN) && a.n <= tune(TN_2D_BLOCK_VAR_MAX);   // k_roll2dbv.hip
    return a.n >= tune(f32 ? TN_2D_BLOCK_MIN_F32 : TN_2D_BLOCK_MIN_F64) &&
           a.n <= std::max(tune(f32 ? TN_2D_BLOCK_MAX_F32 : TN_2D_BLOCK_MAX_F64), tune(f32 ? TN_2D_BLOCK_FOUR_F32 : TN_2D_BLOCK_FOUR_F64));
}

// the headline kernel k_rollout2d: tiles of 64 envs, every row written, 16-byte pieces
bool roll2d_ok(const KArgs& a, int E) {
    return E == 64 && every_row(a) && pieces16(a) && !pipeline_off() && tune(TN_2D_STAGE) != 0;
}
int roll2d_from(bool f32) { return tune(TN_2D_STAGE_MIN) ? tune(TN_2D_STAGE_MIN) : tune(f32 ? TN_2D_STAGE_MIN_F32 : TN_2D_STAGE_MIN_F64); }

// time-parallel 2D rollouts (k_rollout2dt: one wave per env, lane = tick): small and middle batches, where the lane-per-env kernels are
// bound by the chain of their ticks (0.5-0.7 ms per 600 ticks at every N <= 16 384) or leave CUs empty
bool roll2dt_ok(const KArgs& a, bool f32) {
    if (tune(TN_2D_TP) == 0 || !every_row(a) || pipeline_off()) return false;
    if (a.variant) {   // k_rollout2dt<.., VAR>: whole groups of four envs and 16-byte pieces only
        const int lim = tune(TN_2D_TP_VAR_MAX) ? tune(TN_2D_TP_VAR_MAX) : tune((a.tail & SNAC_TAIL_PLAN) ? TN_2D_TP_VAR_PLAN : TN_2D_TP_VAR_SHORT);
        return pieces16(a) && a.n <= lim;
    }
    if (tune(TN_2D_TP_MAX)) return a.n <= tune(TN_2D_TP_MAX);
    const size_t rowb = (size_t)K2D<true, 64>::D * (f32 ? 4 : 8);
    const bool pieces = (((uintptr_t)a.obs) & 15) == 0 && (a.obs_mode == SNAC_OBS_TILED || (((size_t)a.n * rowb) & 15) == 0);
    if (!pieces) return a.n <= tune(TN_2D_TP_MAX_ODD);
    if (f32) return a.n <= tune(TN_2D_TP_MAX_F32);
    return a.n <= tune(TN_2D_TP_MAX_F64) && !(a.n >= tune(TN_2D_TP_GAP_LO) && a.n <= tune(TN_2D_TP_GAP_HI));
}

// time-parallel 1D rollouts (k_rollout1dt).  Its rate levels off at 6-7e10 env-steps/s (instruction issue), the tile kernel's keeps growing
bool roll1dt_ok(const KArgs& a, bool f32) {
    if (tune(TN_1D_TP) == 0 || !every_row(a) || pipeline_off()) return false;
    if (a.variant) return pieces16(a) && a.n <= tune(TN_1D_TP_VAR_MAX);   // k_rollout1dt<.., VAR>: blocks of four envs
    return a.n <= (tune(TN_1D_TP_MAX) ? tune(TN_1D_TP_MAX) : tune(f32 ? TN_1D_TP_MAX_F32 : TN_1D_TP_MAX_F64));
}

// lane-per-env 1D rollouts (k_rollout1dl): batches with a 64-env wave for (nearly) every SIMD; every row written, 16-byte pieces; canonical rows and the layout variants
bool roll1dl_ok(const KArgs& a, bool f32) {
    if (tune(TN_1D_LANE) == 0 || !every_row(a) || !pieces16(a) || pipeline_off()) return false;
    if (a.variant) return a.ld <= 46 && a.n >= tune(a.ld > 16 ? TN_1D_LANE_VAR_MIN : TN_1D_LANE_VAR_SHORT_MIN);
    return a.n >= tune(f32 ? TN_1D_LANE_MIN_F32 : TN_1D_LANE_MIN_F64);
}

// snac_step on identity rows: k_step2d / k_step3d (wide loads, rows through emit_tile)
bool step_stage_ok(const KArgs& a) { return tune(TN_STEP_STAGE) != 0 && !a.src_index && !a.dst_index && pieces16(a); }
// ... with a layout variant: 64 envs per wave are 64 rows of kilobytes per wave -- batches large enough to fill the CUs that way;
// below, the 8-env tiles of k_transition spread the rows over more waves
bool step_var_ok(const KArgs& a, bool f32) {
    if (tune(TN_STEP_VAR_MIN)) return a.n >= tune(TN_STEP_VAR_MIN);
    if (!(a.tail & SNAC_TAIL_PLAN)) return a.n >= tune(TN_STEP_VAR_SHORT);
    return (a.n >= tune(TN_STEP_VAR_HALF_LO) && a.n <= tune(TN_STEP_VAR_HALF_HI)) || a.n >= tune(f32 ? TN_STEP_VAR_F32 : TN_STEP_VAR_F64);
}
bool step_var_half(const KArgs& a) {
    if (tune(TN_STEP_VAR_HALF) >= 0) return tune(TN_STEP_VAR_HALF) != 0;
    return (a.tail & SNAC_TAIL_PLAN) && a.n <= tune(TN_STEP_VAR_HALF_HI);
}
bool step_var3_ok(const KArgs& a) { return a.n >= tune(TN_STEP_VAR3_MIN) && a.frame_val == -1; }

int launch(Op op, const snac_env_desc* d, const KArgs& a, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const bool dyn = d->dynamic != 0;
    int E = pick_tile(d->kind, a.n);
    if (d->kind == SNAC_ENV_2D && op == OP_ROLLOUT && E == 16 && tune(TN_TILE) == 0 && a.n >= tune(TN_2D_TILE32_MIN)) E = 32;   // one wave of 32 per SIMD beats 1.5 of 16
    const char* const tile_name = op == OP_ROLLOUT ? "k_rollout" : (op == OP_TRANSITION ? "k_transition" : "k_aux");
    g_kernel = tile_name;
    if (op == OP_AUX && a.aux_op == AUX_RESET && !a.mask && !a.variant && pieces16(a) && !pipeline_off() && tune(TN_RESET_FAST) != 0 && a.n >= tune(TN_RESET_FAST_MIN)) {
        g_kernel = "k_reset";
        launch_reset(d, a, s);
        const hipError_t e = hipGetLastError();
        return e == hipSuccess ? SNAC_OK : fail_hip(e, "kernel launch");
    }
    if (op == OP_AUX && (a.aux_op == AUX_RESET || a.aux_op == AUX_OBSERVE) && a.obs && !a.variant && pieces16(a) && !pipeline_off() &&
        tune(TN_AUX_STEP) != 0 && tune(TN_STEP_STAGE) != 0 && a.n >= 256) {
        if (d->kind == SNAC_ENV_1D) { g_kernel = "k_step1d"; launch_aux1d(d, a, s); }
        else if (d->kind == SNAC_ENV_2D) { g_kernel = "k_step2d"; launch_aux2d(d, a, s); }
        else { g_kernel = "k_step3dq"; launch_aux3d(d, a, s); }
        const hipError_t e = hipGetLastError();
        return e == hipSuccess ? SNAC_OK : fail_hip(e, "kernel launch");
    }
    if (op == OP_AUX && a.aux_op == AUX_IOU && tune(TN_IOU_FAST) != 0 && a.n >= 256 && (((uintptr_t)a.grid | (uintptr_t)a.plans) & 15) == 0 && !pipeline_off()) {
        g_kernel = "k_iou";
        launch_iou(d, a, s);
        const hipError_t e = hipGetLastError();
        return e == hipSuccess ? SNAC_OK : fail_hip(e, "kernel launch");
    }
    switch (d->kind) {
        case SNAC_ENV_1D:
            // rollouts that write every row: the time-parallel kernel while its rate beats the tile kernel's (lane-per-env transition)
            if (op == OP_ROLLOUT && roll1dl_ok(a, d->obs_dtype == SNAC_OBS_F32)) { g_kernel = "k_rollout1dl"; launch_roll1dl(d, a, s); break; }
            if (op == OP_ROLLOUT && roll1dt_ok(a, d->obs_dtype == SNAC_OBS_F32)) { g_kernel = "k_rollout1dt"; launch_roll1dt(d, a, s); break; }
            if (op == OP_TRANSITION && !pipeline_off() && (a.src_index || a.dst_index) && !a.variant && tune(TN_EDGES1D) != 0 && a.n >= tune(TN_EDGES1D_MIN)) { g_kernel = "k_edges1d"; launch_edges1d(d, a, s); break; }
            if (op == OP_TRANSITION && !pipeline_off() && step_stage_ok(a) && tune(TN_STEP1D) != 0 && a.n >= tune(a.variant ? TN_STEP1D_VAR_MIN : TN_STEP1D_MIN) && a.ld <= 46) { g_kernel = "k_step1d"; launch_step1d(d, a, s); break; }
            launch_tile1d(op, dyn, E, d->obs_dtype, a, s); break;
        case SNAC_ENV_2D:
            if (op == OP_TRANSITION && !pipeline_off() && step_stage_ok(a) && (!a.variant || step_var_ok(a, d->obs_dtype == SNAC_OBS_F32))) { g_kernel = "k_step2d"; launch_step2d(d, a, a.variant && step_var_half(a), s); break; }
            if (op == OP_TRANSITION && !a.variant && !pipeline_off()) { g_kernel = "k_transition2d"; launch_trans2d(d, a, s); break; }
            if (op == OP_ROLLOUT && roll2db_ok(a, d->obs_dtype == SNAC_OBS_F32)) {
                g_kernel = "k_rollout2db";
                const bool f32 = d->obs_dtype == SNAC_OBS_F32;
                int steppers = a.n >= tune(a.variant ? TN_2D_BLOCK_VAR_TWO : (f32 ? TN_2D_BLOCK_TWO_F32 : TN_2D_BLOCK_TWO_F64)) ? 2 : 1;
                if (!a.variant && a.n > tune(f32 ? TN_2D_BLOCK_MAX_F32 : TN_2D_BLOCK_MAX_F64)) steppers = 4;   // (up to SNAC_2D_BLOCK_FOUR_MAX_*: roll2db_ok)
                if (a.variant) launch_roll2dbv(d, a, steppers, s); else launch_roll2db(d, a, steppers, s);
                break;
            }
            if (op == OP_ROLLOUT && roll2dt_ok(a, d->obs_dtype == SNAC_OBS_F32)) { g_kernel = "k_rollout2dt"; launch_roll2dt(d, a, s); break; }
            if (op == OP_ROLLOUT && roll2d_ok(a, a.n >= roll2d_from(d->obs_dtype == SNAC_OBS_F32) ? 64 : E)) { g_kernel = "k_rollout2d"; launch_roll2d(d, a, s); break; }
            launch_tile2d(op, dyn, E, d->obs_dtype, a, s); break;
        default:   // 3D: 2.1 KB of LDS per env -> tiles of 16 (or 8 for small batches: two waves per SIMD sooner)
            if (op == OP_ROLLOUT && roll3db_ok(a, d->obs_dtype == SNAC_OBS_F32)) {
                g_kernel = "k_rollout3db";
                if (a.variant) launch_roll3dbv(d, a, s); else launch_roll3db(d, a, s);
                break;
            }
            if (op == OP_ROLLOUT && E == 8 && !a.variant && (a.obs_mode == SNAC_OBS_ALL || a.obs_mode == SNAC_OBS_TILED) && a.num_plans <= TB_MAX && !pipeline_off()) { g_kernel = "k_rollout3d"; launch_roll3d(d, a, s); break; }
            if (op == OP_TRANSITION && !pipeline_off() && step_stage_ok(a) && !a.variant && tune(TN_STEP3D_QUARTER) != 0 &&
                a.n >= tune(TN_STEP3D_QUARTER_MIN) && a.n <= tune(TN_STEP3D_QUARTER_MAX)) { g_kernel = "k_step3dq"; launch_step3dq(d, a, s); break; }
            if (op == OP_TRANSITION && !pipeline_off() && step_stage_ok(a) && (!a.variant || step_var3_ok(a))) {
                const bool span = !a.variant && tune(TN_STEP3D_SPAN) != 0 && a.n >= tune(TN_STEP3D_SPAN_MIN);   // large batches of canonical rows
                g_kernel = span ? "k_step3ds" : "k_step3d";
                launch_step3d(d, a, span, s);
                break;
            }
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

int snac_tuning(char* out, int32_t cap) {
    if (!out || cap <= 0) return fail(SNAC_ERR_ARG, "null / empty buffer");
    int n = 0;
    for (int i = 0; i < TN_COUNT && n < cap - 1; ++i)
        n += std::snprintf(out + n, (size_t)(cap - n), "%s=%d  # %s%s\n", KNOBS[i].env, tune(i), tune(i) != KNOBS[i].def ? "(overridden) " : "", KNOBS[i].what);
    return n < cap ? SNAC_OK : fail(SNAC_ERR_ARG, "buffer too small");
}

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
