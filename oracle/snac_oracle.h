/*
 * snac_oracle.h -- CPU restatement of the reference env path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the parity oracle for the HIP path in snac_amd/csrc.  It restates, in plain C, the
 * algorithm of ai4ce/SNAC's deep_mobile_printing_{1d1r,2d1r,3d1r} reset()/step()/iou() (six canonical
 * classes, file:line cited at each function in snac_oracle.c) plus numpy's legacy MT19937
 * `seed`/`randint` so that seed-level parity with the reference can be checked.
 *
 * Pinning: tests/test_oracle_golden.py replays every trajectory in tests/golden/traj_*.npz (recorded
 * from the imported reference by tests/golden/make_golden.py) and the sha256 of six 100k-step
 * seed-driven streams (tests/golden/digests.json) through this code; all must match bit for bit.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.  The
 * product (snac_amd/) never does: it fails loudly when libsnac_hip.so is missing.
 *
 * Deliberately different from the product's data layout: the grid is the reference's full bordered
 * array (26x26 / 1x34, frame cells hold -1) in int32, one env per struct (array of structs), so the
 * two implementations share no indexing tricks.
 */
#ifndef SNAC_ORACLE_H
#define SNAC_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_CELLS 676

typedef struct orc_env {
    int32_t dim;         /* 1, 2, 3 */
    int32_t dynamic;     /* 0: static-plan class (obs scalars cb, cs); 1: *_usedata class (cb/tb, cs/T) */
    int32_t hw;          /* HALF_WINDOW_SIZE */
    int32_t H, W;        /* environment_height, environment_width (1D: 1 x 34) */
    int32_t total_step;  /* T */
    int32_t num_actions;
    int32_t obs_dim;     /* 7 or 51 */
    int32_t grid[ORC_MAX_CELLS]; /* environment_memory, row-major */
    int32_t plan[ORC_MAX_CELLS]; /* 1D: plan[0..29]; 2D/3D: the bordered 26x26 plan */
    int32_t pos[2];      /* position_memory[-1]; 1D uses pos[0] */
    int32_t cb, cs, tb;  /* count_brick, count_step, total_brick */
    int32_t step_size;   /* last step size used */
    int32_t plan_idx;
    int32_t obs_norm;    /* observation scalars cb/tb, cs/T (default: = dynamic) */
    int32_t rules_dyn;   /* 3D: termination rules of the dynamic class (default: = dynamic) */
    int32_t frame;       /* value of the frame cells: -1, or 2 in the 2D L-Net variant */
    int32_t brick_gt;    /* 1: done when cb > tb (script/PPO copies), 0: cb >= tb */
    int32_t time_gt;     /* 1: done when cs > T (script/PPO/3d_static), 0: cs >= T */
    int32_t tail;        /* ORC_TAIL_* bits appended to the observation row (layout variants, see orc_set_tail) */
    int32_t last_reward, last_done;   /* of the last step (0, 0 after reset): what ORC_TAIL_RECORD reports */
} orc_env;
enum { ORC_TAIL_POSITION = 1, ORC_TAIL_PLAN = 2, ORC_TAIL_RECORD = 4 };

/* ---- single env ---- */
int  orc_init(orc_env* e, int dim, int dynamic);
int  orc_configure(orc_env* e, int obs_norm, int rules_dyn, int total_step, int frame);
int  orc_set_rules(orc_env* e, int brick_gt, int time_gt);   /* the `>` termination tests of the script/PPO env copies */
/* observation-row tails of the reference's env copies (include/snac_hip.h obs_tail): position (Env/1D/DMP_Env_1D_static_Lnet.py:83),
 * the plan (script/PPO/2d_dynamic/DMP_Env_2d_dynamic_usedata_plan.py:70-71), and the build's own 8-value record; updates obs_dim */
int  orc_set_tail(orc_env* e, int tail);
/* plan: 30 (1D) or 676 (2D/3D) ints; obs: obs_dim doubles (may be NULL) */
int  orc_reset(orc_env* e, const int32_t* plan, int plan_idx, double* obs);
/* returns 0, or -1 for an action outside [0, num_actions) (the reference raises after cs += 1) */
int  orc_step(orc_env* e, int action, int k, double* obs, double* reward, int* done);
void orc_observe(const orc_env* e, double* obs);
/* MCTS variants: transition(state, action) = step() on an explicit state (src == dst allowed); gate_cb >= 0: the
 * brick-limit test of the 3D dynamic class uses this count instead of the state's (see snac_oracle.c) */
int  orc_transition(const orc_env* src, orc_env* dst, int action, int k, int gate_cb, double* obs, double* reward, int* done);
int  orc_set_state(orc_env* e, const int32_t* grid, int r, int c, int cb, int cs);
double orc_iou(const orc_env* e);
/* static plans (Env/1D/DMP_Env_1D_static.py:34-55, Env/2D/DMP_Env_2D_static.py:31-52,
 * Env/3D/DMP_simulator_3d_static_circle.py:42-65) as integer tables; returns cell count or -1 */
int  orc_static_plan(int dim, int plan_choose, int32_t* out);

/* ---- numpy legacy RandomState ---- */
typedef struct orc_mt { uint32_t mt[624]; int idx; } orc_mt;
void     orc_mt_seed(orc_mt* m, uint32_t seed);
uint32_t orc_mt_next(orc_mt* m);
int64_t  orc_mt_randint(orc_mt* m, int64_t lo, int64_t hi); /* np.random.randint(lo, hi) */

/* ---- counter RNG (the build's own batched semantics; include/snac_hip.h "Counter RNG") ---- */
uint32_t orc_rng_word(uint64_t seed, uint32_t stream, uint64_t env, uint32_t t);

/* ---- batched: N independent envs, the semantics of include/snac_hip.h ---- */
typedef struct orc_batch {
    int32_t dim, dynamic, n, num_plans, cells, obs_dim, total_step, num_actions;
    uint64_t seed;
    int64_t env_id_base;
    const int32_t* plans;     /* [num_plans][cells], not owned */
    orc_env* envs;            /* [n] */
    int32_t* episode;         /* [n] resets performed - 1 */
    int32_t* ep_return;       /* [n] running integer return of the current episode */
    uint8_t* need_reset;      /* [n] last step returned done */
    int64_t* stat_episodes;   /* [n] finished episodes */
    int64_t* stat_return;     /* [n] sum of finished-episode returns */
    int64_t* stat_iou_fx;     /* [n] sum of llrint(iou * 2^40) at episode end */
    int64_t* stat_steps;      /* [n] */
} orc_batch;

orc_batch* orc_batch_create(int dim, int dynamic, int n, const int32_t* plans, int num_plans,
                            uint64_t seed, int64_t env_id_base);
void orc_batch_destroy(orc_batch* b);
void orc_batch_set_rules(orc_batch* b, int brick_gt, int time_gt);
/* layout / rule switches of every env (orc_configure + orc_set_tail); updates the batch's obs_dim and total_step */
void orc_batch_configure(orc_batch* b, int obs_norm, int rules_dyn, int total_step, int frame, int tail);
/* mask NULL = all; plan_idx_in NULL = counter RNG (dynamic) / plan 0 (static); obs [n][obs_dim] or NULL */
int  orc_batch_reset(orc_batch* b, const uint8_t* mask, const int32_t* plan_idx_in, double* obs);
/* one vector step at tick t.  actions/step_size NULL = counter RNG.  auto_reset: envs whose previous
 * step returned done are reset first (plan from counter RNG).  Returns 0 or -1 on a bad action. */
int  orc_batch_step(orc_batch* b, uint32_t t, const int8_t* actions, const int8_t* step_size,
                    int auto_reset, double* obs, float* reward, uint8_t* done, int nthreads);
/* T vector steps with auto-reset; actions/step_size [T][n] or NULL; obs [T][n][obs_dim] or NULL (if
 * obs_last_only != 0, obs is [n][obs_dim] and holds the last step only); reward/done [T][n] or NULL */
int  orc_batch_rollout(orc_batch* b, int T, uint32_t t0, const int8_t* actions, const int8_t* step_size,
                       double* obs, int obs_last_only, float* reward, uint8_t* done, int nthreads);
void orc_batch_iou(const orc_batch* b, double* out);
/* m functional transitions, the batch as a node pool: env[dst_index[i]] <- step(env[src_index[i]], ...); NULL = i */
int  orc_batch_transition(orc_batch* b, int m, const int32_t* src_index, const int32_t* dst_index, uint32_t t,
                          const int8_t* actions, const int8_t* step_size, double* obs, float* reward, uint8_t* done);

/* ---- plan generators (the build's specification of the reference's random-triangle / random-sine create_plan()s) ---- */
int  orc_raster_triangle(const int* vx, const int* vy, int sparse, int32_t* img400);
int  orc_make_plan(int dim, int sparse, uint64_t seed, int64_t plan_id, int32_t* out, int32_t* tb);

#ifdef __cplusplus
}
#endif
#endif
