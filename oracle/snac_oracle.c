/*
 * snac_oracle.c -- CPU restatement of the reference env path.  TEST INFRASTRUCTURE ONLY
 * (see snac_oracle.h for who may load it and how it is pinned to the reference).
 *
 * Reference files restated (paths relative to the ai4ce/SNAC tree):
 *   S1  Env/1D/DMP_Env_1D_static.py
 *   D1  Env/1D/DMP_Env_1D_dynamic_usedata_plan.py
 *   S2  Env/2D/DMP_Env_2D_static.py
 *   D2  Env/2D/DMP_Env_2D_dynamic_usedata_plan.py
 *   S3  Env/3D/DMP_simulator_3d_static_circle.py
 *   D3  Env/3D/DMP_simulator_3d_dynamic_triangle_usedata.py
 * The reference keeps every quantity in float64; all of them are small integers except the two
 * observation scalars of the dynamic classes (cb/tb, cs/T) and the IoU, which are single IEEE-754
 * float64 divisions -- restated as such (compile without -ffast-math).
 */
#include "snac_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------- */
/* constants: S1:7-29, D1:8-31, S2:7-29, D2:7-32, S3:8-40, D3:7-43                              */
int orc_init(orc_env* e, int dim, int dynamic) {
    memset(e, 0, sizeof(*e));
    e->dim = dim;
    e->dynamic = dynamic ? 1 : 0;
    e->obs_norm = e->dynamic; e->rules_dyn = e->dynamic; e->frame = -1;
    if (dim == 1) {
        e->hw = 2; e->H = 1; e->W = 30 + 2 * 2; e->total_step = 750; e->num_actions = 3; e->obs_dim = 2 * 2 + 1 + 2;
    } else if (dim == 2) {
        e->hw = 3; e->H = 26; e->W = 26; e->total_step = 600; e->num_actions = 5; e->obs_dim = 7 * 7 + 2;
    } else if (dim == 3) {
        e->hw = 3; e->H = 26; e->W = 26; e->total_step = dynamic ? 1000 : 1300; e->num_actions = 8; e->obs_dim = 7 * 7 + 2;
    } else {
        return -1;
    }
    return 0;
}

/* L-Net env variants (Env/1D/DMP_Env_1D_static_Lnet.py, Env/2D/DMP_Env_2D_static_Lnet.py:61-76,
 * Env/3D/DMP_simulator_3d_static_circle_Lnet.py): a static plan with, per dimension, normalised observation scalars,
 * frame value 2 instead of -1 (2D), the dynamic class's termination rules with total_step 1300 (3D).
 * total_step <= 0 keeps the current value. */
int orc_configure(orc_env* e, int obs_norm, int rules_dyn, int total_step, int frame) {
    e->obs_norm = obs_norm ? 1 : 0;
    e->rules_dyn = rules_dyn ? 1 : 0;
    if (total_step > 0) e->total_step = total_step;
    e->frame = frame;
    return 0;
}

/* the env copies under script/PPO write `>` in some termination tests: count_brick > total_brick
 * (script/PPO/1d_dynamic/DMP_Env_1D_dynamic_usedata_plan.py:93, script/PPO/2d_static/DMP_Env_2D_static.py:137,
 * script/PPO/3d_static/DMP_simulator_3d_static_circle.py:205) and count_step > total_step (3d_static :221) */
int orc_set_rules(orc_env* e, int brick_gt, int time_gt) {
    e->brick_gt = brick_gt ? 1 : 0;
    e->time_gt = time_gt ? 1 : 0;
    return 0;
}

int orc_set_tail(orc_env* e, int tail) {
    int base = (e->dim == 1) ? 7 : 51;
    e->tail = tail;
    e->obs_dim = base + ((tail & ORC_TAIL_POSITION) ? (e->dim == 1 ? 1 : 2) : 0) + ((tail & ORC_TAIL_PLAN) ? (e->dim == 1 ? 30 : 400) : 0) +
                 ((tail & ORC_TAIL_RECORD) ? 8 : 0);
    return e->obs_dim;
}

#define G(e, r, c) ((e)->grid[(r) * (e)->W + (c)])
#define P(e, r, c) ((e)->plan[(r) * (e)->W + (c)])

/* observation: S1:81-83 / D1:66-70 (1D window g[p-2..p+2]); S2:78-82, D2:68-72, D3:125-129 (7x7 window,
 * row-major flatten) followed by [cb, cs] (static) or [cb/tb, cs/T] (dynamic: D2:64-66, D3:73-75, D1:68-70) */
static void observe_row(const orc_env* e, double* obs, int in_step);
/* the row outside a step (reset of an env a mask leaves alone, snac_observe): the record tail reports reward 0 and done = the
 * env's pending-reset flag (include/snac_hip.h, SNAC_TAIL_RECORD) */
void orc_observe(const orc_env* e, double* obs) { observe_row(e, obs, 0); }

static void observe_row(const orc_env* e, double* obs, int in_step) {
    int n = 0;
    if (e->dim == 1) {
        for (int j = -e->hw; j <= e->hw; ++j) obs[n++] = (double)e->grid[e->pos[0] + j];
    } else {
        for (int i = -e->hw; i <= e->hw; ++i)
            for (int j = -e->hw; j <= e->hw; ++j) obs[n++] = (double)G(e, e->pos[0] + i, e->pos[1] + j);
    }
    if (e->obs_norm) {
        obs[n++] = (double)e->cb / (double)e->tb;
        obs[n++] = (double)e->cs / (double)e->total_step;
    } else {
        obs[n++] = (double)e->cb;
        obs[n++] = (double)e->cs;
    }
    /* tails of the env copies: Env/1D/DMP_Env_1D_static_Lnet.py:83 appends the position; the script/PPO dataset classes append the
     * plan (1D: 30 heights, 2D / 3D: input_plan = plan[3:23, 3:23] flattened) */
    if (e->tail & ORC_TAIL_POSITION) {
        obs[n++] = (double)e->pos[0];
        if (e->dim != 1) obs[n++] = (double)e->pos[1];
    }
    if (e->tail & ORC_TAIL_PLAN) {
        if (e->dim == 1) for (int i = 0; i < 30; ++i) obs[n++] = (double)e->plan[i];
        else for (int r = 3; r < 23; ++r) for (int c = 3; c < 23; ++c) obs[n++] = (double)P(e, r, c);
    }
    if (e->tail & ORC_TAIL_RECORD) {
        obs[n++] = in_step ? (double)e->last_reward : 0.0; obs[n++] = (double)e->last_done;
        obs[n++] = (double)e->pos[0]; obs[n++] = (double)e->pos[1];
        obs[n++] = (double)e->cb; obs[n++] = (double)e->cs; obs[n++] = (double)e->tb; obs[n++] = (double)e->plan_idx;
    }
}

/* reset: S1:66-83, D1:40-70, S2:54-76, D2:34-66, S3:67-86, D3:45-75 */
int orc_reset(orc_env* e, const int32_t* plan, int plan_idx, double* obs) {
    int cells = (e->dim == 1) ? 30 : e->H * e->W;
    int64_t area = 0;
    memcpy(e->plan, plan, sizeof(int32_t) * (size_t)cells);
    for (int i = 0; i < cells; ++i) area += plan[i]; /* sum(y) S1:53; sum(sum(plan)) S2:50; count*z S3:62-64; D3:49 */
    e->tb = (int32_t)area;
    if (e->dim == 2 && e->tb < 30) e->tb = 30;       /* S2:56-57, D2:45-46 (2D only) */
    e->plan_idx = plan_idx;
    for (int r = 0; r < e->H; ++r)
        for (int c = 0; c < e->W; ++c) {
            int frame = (c < e->hw) || (c >= e->W - e->hw);
            if (e->dim != 1) frame = frame || (r < e->hw) || (r >= e->H - e->hw);
            G(e, r, c) = frame ? e->frame : 0;       /* S1:69-71, D2:49-53; 2 in the 2D L-Net variant */
        }
    e->cb = 0;
    e->cs = 0;
    e->pos[0] = e->hw;                               /* S1:77, D2:60 */
    e->pos[1] = (e->dim == 1) ? 0 : e->hw;
    if (e->dim == 3) e->step_size = 1;               /* S3:78, D3:66 */
    e->last_reward = 0; e->last_done = 0;
    if (obs) orc_observe(e, obs);
    return 0;
}

static int clip1(const orc_env* e, int p) {          /* S1:57-64, D1:32-39 */
    int lo = e->hw, hi = 30 + e->hw - 1;
    if (p <= lo) return lo;
    if (p >= hi) return hi;
    return p;
}

static void clip2(const orc_env* e, int* pos) {      /* S2:84-93, D2:74-83, S3:142-151, D3:131-140 */
    int lo = e->hw, hi = 20 + e->hw - 1;             /* plan_width used for both coordinates */
    if (pos[0] <= lo) pos[0] = lo;
    if (pos[1] <= lo) pos[1] = lo;
    if (pos[0] >= hi) pos[0] = hi;
    if (pos[1] >= hi) pos[1] = hi;
}

/* 1D step: S1:85-136, D1:71-120 */
static int step1(orc_env* e, int action, int k, double* reward, int* done) {
    int p;
    e->cs += 1;
    e->step_size = k;
    if (action == 0) {
        p = clip1(e, e->pos[0] - k);
    } else if (action == 1) {
        p = clip1(e, e->pos[0] + k);
    } else if (action == 2) {
        p = e->pos[0];
        e->cb += 1;
        e->grid[p] += 1;
        if (e->cb >= e->tb + e->brick_gt) {                        /* S1:107-114 */
            *reward = 0.0; *done = 1;
            return 0;
        }
        *done = (e->cs >= e->total_step + e->time_gt);            /* S1:116 */
        if (e->grid[p] > e->plan[p - e->hw]) *reward = -1.0;
        else if (e->grid[p] == e->plan[p - e->hw]) *reward = 10.0;
        else *reward = 1.0;
        return 0;
    } else {
        return -1;                                   /* `position` unbound in the reference */
    }
    e->pos[0] = p;
    *done = (e->cs >= e->total_step + e->time_gt);                /* S1:130 */
    *reward = 0.0;
    return 0;
}

/* 2D step: S2:95-154, D2:85-147 */
static int step2(orc_env* e, int action, int k, double* reward, int* done) {
    int pos[2] = { e->pos[0], e->pos[1] };
    e->cs += 1;
    e->step_size = k;
    if (action == 0) { pos[1] -= k; clip2(e, pos); }
    else if (action == 1) { pos[1] += k; clip2(e, pos); }
    else if (action == 2) { pos[0] += k; clip2(e, pos); }     /* "up" is row + k, D2:100-103 */
    else if (action == 3) { pos[0] -= k; clip2(e, pos); }
    else if (action == 4) {
        e->cb += 1;
        G(e, pos[0], pos[1]) += 1;
        if (e->cb >= e->tb + e->brick_gt) {                        /* D2:117-126 */
            if (G(e, pos[0], pos[1]) > 1) G(e, pos[0], pos[1]) = 1;
            *reward = 0.0; *done = 1;
            return 0;
        }
        *done = (e->cs >= e->total_step + e->time_gt);            /* D2:128 */
        /* compare the un-clamped cell, then clamp: D2:129-135.  The `<` case leaves `reward` unbound in
         * the reference; it cannot occur (cell >= 1 >= plan). */
        if (G(e, pos[0], pos[1]) > P(e, pos[0], pos[1])) *reward = 0.0;
        else if (G(e, pos[0], pos[1]) == P(e, pos[0], pos[1])) *reward = 5.0;
        else return -2;
        if (G(e, pos[0], pos[1]) > 1) G(e, pos[0], pos[1]) = 1;
        return 0;
    } else {
        return -1;
    }
    e->pos[0] = pos[0]; e->pos[1] = pos[1];
    *done = (e->cs >= e->total_step + e->time_gt);                /* D2:141 */
    *reward = 0.0;
    return 0;
}

static const int DR[4] = { 0, 0, 1, -1 };            /* left, right, "up" (row+1), "down" (row-1): S3:92-95 */
static const int DC[4] = { -1, 1, 0, 0 };

static void check_sur(const orc_env* e, int* check) { /* S3:88-102, D3:77-91 */
    for (int i = 0; i < 8; ++i) check[i] = 0;
    for (int i = 0; i < 4; ++i) {
        int v = G(e, e->pos[0] + DR[i], e->pos[1] + DC[i]);
        if (v == -1) { check[i] = 1; check[i + 4] = 1; }
        else if (v > 0) check[i] = 1;
    }
}

static int move_step(const orc_env* e, int action, int k) { /* S3:104-134, D3:93-123 */
    int move = 0;
    for (int i = 0; i < k; ++i) {
        if (G(e, e->pos[0] + DR[action] * (i + 1), e->pos[1] + DC[action] * (i + 1)) == 0) move += 1;
        else break;
    }
    return move;
}

static double reward_check(const orc_env* e, int r, int c) { /* S3:232-239, D3:233-240 */
    if (G(e, r, c) > P(e, r, c)) return -1.0;
    if (G(e, r, c) == P(e, r, c)) return 10.0;
    return 1.0;
}

/* 3D step: S3:153-230, D3:142-231 */
static int step3(orc_env* e, int action, int k, double* reward, int* done) {
    int check[8];
    e->cs += 1;
    e->step_size = k;
    if (action < 0 || action > 7) return -1;
    check_sur(e, check);
    if (action < 4 && check[action] == 0) {
        int m = move_step(e, action, k);
        int pos[2] = { e->pos[0] + DR[action] * m, e->pos[1] + DC[action] * m };
        clip2(e, pos);
        e->pos[0] = pos[0]; e->pos[1] = pos[1];
    } else if (action > 3) {
        int build = 0, tr = 0, tc = 0;
        if (check[action] == 0) {                    /* check[4..7]: target is not the frame */
            build = 1;
            e->cb += 1;
            tr = e->pos[0] + DR[action - 4]; tc = e->pos[1] + DC[action - 4];
            G(e, tr, tc) += 1;
        }
        if (e->rules_dyn) {
            int after[8];
            check_sur(e, after);                     /* D3:199: re-evaluated AFTER the build */
            if (after[0] && after[1] && after[2] && after[3]) { *done = 1; *reward = -100.0; return 0; }
            if (e->cb >= e->tb + e->brick_gt) { *done = 1; *reward = 0.0; return 0; }
            if (build) { *done = 0; *reward = reward_check(e, tr, tc); return 0; }  /* D3:214-221 */
        } else {
            int boxed = check[0] && check[1] && check[2] && check[3];  /* S3:210: neighbours BEFORE the build */
            if (e->cb >= e->tb + e->brick_gt || boxed) { *done = 1; *reward = 0.0; return 0; }
            if (build) { *done = 0; *reward = reward_check(e, tr, tc); return 0; }
        }
    }
    /* moves, blocked moves, blocked builds: S3:226-230, D3:226-231 */
    *done = (e->cs >= e->total_step + e->time_gt);
    if (!e->rules_dyn) *done = *done || (check[0] && check[1] && check[2] && check[3]);
    *reward = 0.0;
    return 0;
}

int orc_step(orc_env* e, int action, int k, double* obs, double* reward, int* done) {
    int rc;
    double r = 0.0;
    int d = 0;
    if (e->dim == 1) rc = step1(e, action, k, &r, &d);
    else if (e->dim == 2) rc = step2(e, action, k, &r, &d);
    else rc = step3(e, action, k, &r, &d);
    if (rc) return rc;
    e->last_reward = (int32_t)r; e->last_done = d;
    if (obs) observe_row(e, obs, 1);
    if (reward) *reward = r;
    if (done) *done = d;
    return 0;
}

/* transition(state, action) of the MCTS variants: step() restated on an explicit state tuple
 * (Env/1D/DMP_Env_1D_dynamic_MCTS.py:82-139, Env/2D/DMP_ENV_2D_dynamic_MCTS.py:117-175,
 * Env/3D/DMP_simulator_3d_static_circle_MCTS.py:215-288, Env/3D/DMP_simulator_3d_dynamic_triangle_MCTS.py:195-277).
 * gate_cb >= 0: the 3D dynamic class tests `self.count_brick >= self.total_brick` (:258) -- the brick count of the ENV
 * object, not the one of the state; pass the env's count here.  src == dst is allowed. */
int orc_transition(const orc_env* src, orc_env* dst, int action, int k, int gate_cb, double* obs, double* reward, int* done) {
    int rc, tb;
    if (dst != src) *dst = *src;
    tb = dst->tb;
    if (gate_cb >= 0 && dst->dim == 3 && dst->rules_dyn) dst->tb = (gate_cb >= tb) ? -0x3fffffff : 0x3fffffff;   /* always / never reached */
    rc = orc_step(dst, action, k, NULL, reward, done);
    dst->tb = tb;
    if (rc) return rc;
    if (obs) observe_row(dst, obs, 1);
    return 0;
}

/* load an explicit state: grid = the bordered environment_memory (H*W ints) */
int orc_set_state(orc_env* e, const int32_t* grid, int r, int c, int cb, int cs) {
    if (r < 0 || r >= (e->dim == 1 ? e->W : e->H) || c < 0 || c >= e->W) return -1;
    for (int i = 0; i < e->H * e->W; ++i) e->grid[i] = grid[i];
    e->pos[0] = r; e->pos[1] = c; e->cb = cb; e->cs = cs;
    return 0;
}

/* IoU: 1D S1:138-151 / D1:121-133; 2D caller-side boolean IoU script/DQN/2d/DQN_2d_dynamic.py:63-71 and
 * D2:153-159; 3D S3:257-276 / D3:258-277 */
double orc_iou(const orc_env* e) {
    if (e->dim == 1) {
        int64_t a1 = 0, a2 = 0, k = 0;
        for (int i = 0; i < 30; ++i) {
            int g = e->grid[e->hw + i], p = e->plan[i];
            a1 += p; a2 += g;
            if (g > p) k += g - p;
        }
        int64_t cross = a2 - k;
        return (double)cross / (double)(a1 + a2 - cross);
    }
    if (e->dim == 2) {
        int64_t inter = 0, uni = 0;
        for (int r = 3; r < 23; ++r)
            for (int c = 3; c < 23; ++c) {
                int g = G(e, r, c) != 0, p = P(e, r, c) != 0;
                inter += (g && p); uni += (g || p);
            }
        return (double)inter / (double)uni;
    }
    {
        int64_t cross = 0;
        for (int r = 3; r < 23; ++r)
            for (int c = 3; c < 23; ++c) cross += (G(e, r, c) > P(e, r, c)) ? P(e, r, c) : G(e, r, c);
        return (double)cross / (double)((int64_t)e->tb + e->cb - cross);
    }
}

/* ------------------------------------------------------------------------------------------- */
/* static plans.  The reference computes them with numpy sin / a gaussian / matplotlib's
 * CirclePolygon.contains_point; the integer results were captured from the reference in the build
 * container (tests/golden/static_plans.npz) and are tabulated here.                                */
static const int16_t PLAN1D[3][30] = {
    { 20, 22, 24, 26, 27, 29, 30, 30, 30, 30, 29, 27, 26, 24, 22, 20, 18, 16, 14, 13, 11, 10, 10, 10, 10, 11, 13, 14, 16, 18 },
    { 17, 17, 17, 17, 17, 17, 17, 17, 17, 18, 19, 22, 25, 28, 30, 30, 28, 25, 22, 19, 18, 17, 17, 17, 17, 17, 17, 17, 17, 17 },
    { 25, 25, 25, 25, 25, 15, 15, 15, 15, 15, 25, 25, 25, 25, 25, 15, 15, 15, 15, 15, 25, 25, 25, 25, 25, 15, 15, 15, 15, 15 },
};
/* 26-bit row masks of the bordered 26x26 plan, bit (25 - col); rows 0..25 */
static const uint32_t PLAN2D[2][26] = {
    { 0, 0, 0, 0, 0, 0, 30720, 130560, 261888, 524160, 524160, 1048512, 1048512, 1048512, 1048512, 524160, 524160,
      261888, 130560, 30720, 0, 0, 0, 0, 0, 0 },
    { 0, 0, 0, 0, 0, 64512, 231168, 393600, 786624, 524352, 1572960, 1048608, 1048608, 1048608, 1048608, 1572960,
      524352, 786624, 393600, 231168, 64512, 0, 0, 0, 0, 0 },
};

int orc_static_plan(int dim, int plan_choose, int32_t* out) {
    if (dim == 1) {
        if (plan_choose < 0 || plan_choose > 2) return -1;
        for (int i = 0; i < 30; ++i) out[i] = PLAN1D[plan_choose][i];
        return 30;
    }
    if (dim == 2 || dim == 3) {
        if (plan_choose < 0 || plan_choose > 1) return -1;
        for (int r = 0; r < 26; ++r)
            for (int c = 0; c < 26; ++c) {
                int bit = (PLAN2D[plan_choose][r] >> (25 - c)) & 1;
                out[r * 26 + c] = bit * (dim == 3 ? 6 : 1);   /* plan * self.z, S3:63 */
            }
        return 676;
    }
    return -1;
}

/* ------------------------------------------------------------------------------------------- */
/* numpy legacy RandomState: seed(int) = init_genrand; randint = masked rejection on 32-bit draws   */
void orc_mt_seed(orc_mt* m, uint32_t seed) {
    m->mt[0] = seed;
    for (int i = 1; i < 624; ++i) m->mt[i] = 1812433253u * (m->mt[i - 1] ^ (m->mt[i - 1] >> 30)) + (uint32_t)i;
    m->idx = 624;
}

uint32_t orc_mt_next(orc_mt* m) {
    uint32_t y;
    if (m->idx >= 624) {
        int i;
        for (i = 0; i < 624 - 397; ++i) {
            y = (m->mt[i] & 0x80000000u) | (m->mt[i + 1] & 0x7fffffffu);
            m->mt[i] = m->mt[i + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        for (; i < 623; ++i) {
            y = (m->mt[i] & 0x80000000u) | (m->mt[i + 1] & 0x7fffffffu);
            m->mt[i] = m->mt[i + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        y = (m->mt[623] & 0x80000000u) | (m->mt[0] & 0x7fffffffu);
        m->mt[623] = m->mt[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        m->idx = 0;
    }
    y = m->mt[m->idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

int64_t orc_mt_randint(orc_mt* m, int64_t lo, int64_t hi) {
    uint64_t rng = (uint64_t)(hi - lo - 1), mask = rng, v;
    if (rng == 0) return lo;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
    if (rng <= 0xffffffffull) {
        do { v = orc_mt_next(m) & mask; } while (v > rng);
    } else {
        do { v = (((uint64_t)orc_mt_next(m) << 32) | orc_mt_next(m)) & mask; } while (v > rng);
    }
    return lo + (int64_t)v;
}

/* ------------------------------------------------------------------------------------------- */
/* counter RNG (include/snac_hip.h "Counter RNG"; numpy statement in tests/rng_spec.py)            */
static uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

uint32_t orc_rng_word(uint64_t seed, uint32_t stream, uint64_t env, uint32_t t) {
    uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
    uint32_t key = mix32(lo ^ mix32(hi + 0x9E3779B9u * (stream + 1u)));
    uint32_t elo = (uint32_t)env, ehi = (uint32_t)(env >> 32);
    uint32_t e0 = mix32(key ^ mix32(elo + 0x85EBCA6Bu * ehi + 0x1B873593u));
    uint32_t e1 = mix32((key + 0x27D4EB2Fu) ^ mix32((elo ^ 0x165667B1u) + 0xC2B2AE35u * ehi));
    return mix32(mix32(e0 ^ (0x9E3779B9u * t)) + e1);
}

/* ------------------------------------------------------------------------------------------- */
/* batched semantics                                                                             */
orc_batch* orc_batch_create(int dim, int dynamic, int n, const int32_t* plans, int num_plans,
                            uint64_t seed, int64_t env_id_base) {
    orc_batch* b = (orc_batch*)calloc(1, sizeof(orc_batch));
    orc_env proto;
    if (!b || orc_init(&proto, dim, dynamic)) { free(b); return NULL; }
    b->dim = dim; b->dynamic = dynamic ? 1 : 0; b->n = n; b->num_plans = num_plans;
    b->cells = (dim == 1) ? 30 : 676;
    b->obs_dim = proto.obs_dim; b->total_step = proto.total_step; b->num_actions = proto.num_actions;
    b->seed = seed; b->env_id_base = env_id_base; b->plans = plans;
    b->envs = (orc_env*)calloc((size_t)n, sizeof(orc_env));
    b->episode = (int32_t*)calloc((size_t)n, sizeof(int32_t));
    b->ep_return = (int32_t*)calloc((size_t)n, sizeof(int32_t));
    b->need_reset = (uint8_t*)calloc((size_t)n, 1);
    b->stat_episodes = (int64_t*)calloc((size_t)n, sizeof(int64_t));
    b->stat_return = (int64_t*)calloc((size_t)n, sizeof(int64_t));
    b->stat_iou_fx = (int64_t*)calloc((size_t)n, sizeof(int64_t));
    b->stat_steps = (int64_t*)calloc((size_t)n, sizeof(int64_t));
    for (int i = 0; i < n; ++i) { b->envs[i] = proto; b->episode[i] = -1; }
    return b;
}

void orc_batch_set_rules(orc_batch* b, int brick_gt, int time_gt) {
    for (int i = 0; i < b->n; ++i) orc_set_rules(&b->envs[i], brick_gt, time_gt);
}

void orc_batch_configure(orc_batch* b, int obs_norm, int rules_dyn, int total_step, int frame, int tail) {
    for (int i = 0; i < b->n; ++i) { orc_configure(&b->envs[i], obs_norm, rules_dyn, total_step, frame); orc_set_tail(&b->envs[i], tail); }
    b->obs_dim = b->envs[0].obs_dim; b->total_step = b->envs[0].total_step;
}

void orc_batch_destroy(orc_batch* b) {
    if (!b) return;
    free(b->envs); free(b->episode); free(b->ep_return); free(b->need_reset);
    free(b->stat_episodes); free(b->stat_return); free(b->stat_iou_fx); free(b->stat_steps);
    free(b);
}

/* plan_idx_in >= 0: that row; -1: counter RNG (dynamic) / row 0 (static); -2 (auto-reset): counter RNG (dynamic) / the
 * env keeps its own row (static: per-env static plans survive an episode end) */
static void batch_reset_one(orc_batch* b, int i, int plan_idx_in, double* obs) {
    int idx;
    b->episode[i] += 1;
    if (plan_idx_in >= 0) idx = plan_idx_in;
    else if (b->dynamic) {
        uint32_t w = orc_rng_word(b->seed, 1u, (uint64_t)(b->env_id_base + i), (uint32_t)b->episode[i]);
        idx = (int)(((uint64_t)w * (uint64_t)b->num_plans) >> 32);
    } else idx = (plan_idx_in == -2) ? b->envs[i].plan_idx : 0;
    orc_reset(&b->envs[i], b->plans + (size_t)idx * (size_t)b->cells, idx, obs);
    b->ep_return[i] = 0;
    b->need_reset[i] = 0;
}

int orc_batch_reset(orc_batch* b, const uint8_t* mask, const int32_t* plan_idx_in, double* obs) {
    for (int i = 0; i < b->n; ++i) {
        if (mask && !mask[i]) { if (obs) orc_observe(&b->envs[i], obs + (size_t)i * b->obs_dim); continue; }
        if (plan_idx_in && (plan_idx_in[i] < 0 || plan_idx_in[i] >= b->num_plans)) return -1;
        batch_reset_one(b, i, plan_idx_in ? plan_idx_in[i] : -1, obs ? obs + (size_t)i * b->obs_dim : NULL);
    }
    return 0;
}

static int batch_step_one(orc_batch* b, int i, uint32_t t, int a_in, int k_in, int auto_reset,
                          double* obs, float* reward, uint8_t* done) {
    orc_env* e = &b->envs[i];
    double r = 0.0;
    int d = 0, a = a_in, k = k_in, rc;
    if (auto_reset && b->need_reset[i]) batch_reset_one(b, i, -2, NULL);
    if (a_in < 0 || k_in < 0) {
        uint32_t w = orc_rng_word(b->seed, 0u, (uint64_t)(b->env_id_base + i), t);
        if (a_in < 0) a = (int)(((w >> 16) * (uint32_t)b->num_actions) >> 16);
        if (k_in < 0) k = 1 + (int)(((w & 0xffffu) * 3u) >> 16);
    }
    rc = orc_step(e, a, k, obs, &r, &d);
    if (rc) return rc;
    if (reward) *reward = (float)r;
    if (done) *done = (uint8_t)d;
    b->ep_return[i] += (int32_t)r;                   /* rewards are integers: {-100,-1,0,1,5,10} */
    b->stat_steps[i] += 1;
    b->need_reset[i] = (uint8_t)d;
    if (d) {
        b->stat_episodes[i] += 1;
        b->stat_return[i] += b->ep_return[i];
        b->stat_iou_fx[i] += llrint(orc_iou(e) * 1099511627776.0); /* 2^40 fixed point */
    }
    return 0;
}

int orc_batch_step(orc_batch* b, uint32_t t, const int8_t* actions, const int8_t* step_size,
                   int auto_reset, double* obs, float* reward, uint8_t* done, int nthreads) {
    int bad = 0;
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(static) reduction(| : bad)
    for (int i = 0; i < b->n; ++i) {
        int rc = batch_step_one(b, i, t, actions ? actions[i] : -1, step_size ? step_size[i] : -1, auto_reset,
                                obs ? obs + (size_t)i * b->obs_dim : NULL, reward ? reward + i : NULL,
                                done ? done + i : NULL);
        bad |= (rc != 0);
    }
    return bad ? -1 : 0;
}

int orc_batch_rollout(orc_batch* b, int T, uint32_t t0, const int8_t* actions, const int8_t* step_size,
                      double* obs, int obs_last_only, float* reward, uint8_t* done, int nthreads) {
    int bad = 0;
    size_t n = (size_t)b->n, D = (size_t)b->obs_dim;
    if (nthreads < 1) nthreads = 1;
    /* envs are independent: each thread runs its envs through all T ticks */
#pragma omp parallel for num_threads(nthreads) schedule(static) reduction(| : bad)
    for (int i = 0; i < b->n; ++i) {
        for (int s = 0; s < T; ++s) {
            size_t o = (size_t)s * n + (size_t)i;
            double* op = NULL;
            if (obs) op = obs_last_only ? (s == T - 1 ? obs + (size_t)i * D : NULL) : obs + o * D;
            int rc = batch_step_one(b, i, t0 + (uint32_t)s, actions ? actions[o] : -1, step_size ? step_size[o] : -1, 1,
                                    op, reward ? reward + o : NULL, done ? done + o : NULL);
            bad |= (rc != 0);
        }
    }
    return bad ? -1 : 0;
}

void orc_batch_iou(const orc_batch* b, double* out) {
    for (int i = 0; i < b->n; ++i) out[i] = orc_iou(&b->envs[i]);
}

/* m functional transitions on the batch used as a node pool (include/snac_hip.h snac_transition):
 *   env[dst_index[i]] <- step(env[src_index[i]], actions[i], step size i),  NULL index = i.
 * No auto-reset and no episodic sums (a search is not an episode); the running return and the episode counter travel
 * with the state.  step_size / actions NULL: counter RNG stream 0 keyed by (env_id_base + i, t).  Every source is read
 * before any destination is written.  Returns -1 on a bad action / index. */
int orc_batch_transition(orc_batch* b, int m, const int32_t* src_index, const int32_t* dst_index, uint32_t t,
                         const int8_t* actions, const int8_t* step_size, double* obs, float* reward, uint8_t* done) {
    orc_env* tmp;
    int32_t *ep, *ret;
    uint8_t* nr;
    int bad = 0;
    if (m < 0) return -1;
    for (int i = 0; i < m; ++i) {
        int s = src_index ? src_index[i] : i, d = dst_index ? dst_index[i] : i;
        if (s < 0 || s >= b->n || d < 0 || d >= b->n) return -1;
    }
    tmp = (orc_env*)malloc((size_t)(m ? m : 1) * sizeof(orc_env));
    ep = (int32_t*)malloc((size_t)(m ? m : 1) * sizeof(int32_t));
    ret = (int32_t*)malloc((size_t)(m ? m : 1) * sizeof(int32_t));
    nr = (uint8_t*)malloc((size_t)(m ? m : 1));
    for (int i = 0; i < m; ++i) {
        int s = src_index ? src_index[i] : i;
        double r = 0.0;
        int d = 0, a, k;
        uint32_t w = orc_rng_word(b->seed, 0u, (uint64_t)(b->env_id_base + i), t);
        a = actions ? actions[i] : (int)(((w >> 16) * (uint32_t)b->num_actions) >> 16);
        k = step_size ? step_size[i] : 1 + (int)(((w & 0xffffu) * 3u) >> 16);
        if (orc_transition(&b->envs[s], &tmp[i], a, k, -1, obs ? obs + (size_t)i * b->obs_dim : NULL, &r, &d)) bad = 1;
        ep[i] = b->episode[s];
        ret[i] = b->ep_return[s] + (int32_t)r;
        nr[i] = (uint8_t)d;
        if (reward) reward[i] = (float)r;
        if (done) done[i] = (uint8_t)d;
    }
    for (int i = 0; i < m; ++i) {
        int d = dst_index ? dst_index[i] : i;
        b->envs[d] = tmp[i]; b->episode[d] = ep[i]; b->ep_return[d] = ret[i]; b->need_reset[d] = nr[i];
    }
    free(tmp); free(ep); free(ret); free(nr);
    return bad ? -1 : 0;
}


/* ------------------------------------------------------------------------------------------- */
/* plan generators (include/snac_hip.h "plan generators"): CPU restatement of the build's own specification of the
 * reference's create_plan()s -- random triangles (Env/2D/DMP_Env_2D_dynamic_hindsight_replay_usedata.py:37-59, the 3D
 * bound of script/HumanPlayerGUI/env/Env3D.py:360-364) and random sine curves
 * (Env/1D/DMP_Env_1D_dynamic_hindsight_replay.py:29-42).  cv2 is absent here: the rasteriser is the specification's
 * (8-connected Bresenham outline, centre-inside fill), not cv2's -- parity with cv2 is UNPINNED. */
static void plot(int img[20][20], int x, int y) { if (x >= 0 && x < 20 && y >= 0 && y < 20) img[y][x] = 1; }

/* one edge as cv2 draws it (cv2.polylines / the boundary pass of cv2.fillPoly with thickness 1, LINE_8, shift 0 both end
 * in LineIterator(pt1, pt2, 8, leftToRight = true)): start at the LEFT end point, one pixel per step along the longer
 * axis, a diagonal step whenever the running error dx - 2 dy has gone negative -- so an exact tie stays on the row */
static void cv_line(int img[20][20], int x1, int y1, int x2, int y2) {
    if (x2 < x1) { int t = x1; x1 = x2; x2 = t; t = y1; y1 = y2; y2 = t; }
    int dx = x2 - x1, dy = y2 - y1, sy = dy < 0 ? -1 : 1;
    int majx = 1, majy = 0, minx = 0, miny = sy;
    if (dy < 0) dy = -dy;
    if (dy > dx) { int t = dx; dx = dy; dy = t; majx = 0; majy = sy; minx = 1; miny = 0; }
    int err = dx - 2 * dy, x = x1, y = y1;
    for (int i = 0; i <= dx; ++i) {
        plot(img, x, y);
        if (err < 0) { err += 2 * dx; x += minx; y += miny; }
        err -= 2 * dy;
        x += majx; y += majy;
    }
}

/* the interior as cv2.fillPoly fills it (FillEdgeCollection): every non-horizontal edge runs from its upper end (x0, y0) in
 * 16.16 fixed point with slope ((x1 - x0) << 16) / (y1 - y0) (C division: towards zero); scanline y in [y_min, y_max) takes
 * the two edges with y0 <= y < y1 and fills ceil(left) .. floor(right) */
static void cv_fill(int img[20][20], const int* vx, const int* vy) {
    long ex[3], edx[3];
    int ey0[3], ey1[3], ne = 0, ymin = 99, ymax = -99;
    for (int e = 0; e < 3; ++e) {
        int ax = vx[(e + 2) % 3], ay = vy[(e + 2) % 3], bx = vx[e], by = vy[e];   /* previous vertex -> this vertex */
        if (ay == by) continue;
        if (ay > by) { int t = ax; ax = bx; bx = t; t = ay; ay = by; by = t; }
        ex[ne] = (long)ax << 16; edx[ne] = (((long)(bx - ax)) * 65536) / (by - ay); ey0[ne] = ay; ey1[ne] = by; ++ne;
        if (ay < ymin) ymin = ay;
        if (by > ymax) ymax = by;
    }
    for (int y = ymin; y < ymax && ne >= 2; ++y) {
        long xs[3];
        int k = 0;
        for (int e = 0; e < ne; ++e)
            if (ey0[e] <= y && y < ey1[e]) xs[k++] = ex[e] + (long)(y - ey0[e]) * edx[e];
        if (k < 2) continue;
        long lo = xs[0] < xs[1] ? xs[0] : xs[1], hi = xs[0] < xs[1] ? xs[1] : xs[0];
        for (long x = (lo + 65535) >> 16; x <= (hi >> 16); ++x) plot(img, (int)x, y);
    }
}

/* one rasterisation: vertices (x, y) = (column, row) as cv2 takes them; returns the number of cells set */
int orc_raster_triangle(const int* vx, const int* vy, int sparse, int32_t* img400) {
    int img[20][20];
    int area = 0;
    memset(img, 0, sizeof(img));
    for (int e = 0; e < 3; ++e) cv_line(img, vx[(e + 2) % 3], vy[(e + 2) % 3], vx[e], vy[e]);
    if (!sparse) cv_fill(img, vx, vy);
    for (int y = 0; y < 20; ++y)
        for (int x = 0; x < 20; ++x) { img400[y * 20 + x] = img[y][x]; area += img[y][x]; }
    return area;
}

/* the specified sine (snac_hip.hip spec_sin): quadrant reduction with a two-part pi/2, fdlibm kernel polynomials, fused
 * multiply-adds throughout -- compile with -ffp-contract=off so that nothing else is fused */
static double spec_sin(double x) {
    static const double S[6] = { -1.66666666666666324348e-01, 8.33333333332248946124e-03, -1.98412698298579493134e-04,
                                 2.75573137070700676789e-06, -2.50507602534068634195e-08, 1.58969099521155010221e-10 };
    static const double Cc[6] = { 4.16666666666666019037e-02, -1.38888888888741095749e-03, 2.48015872894767294178e-05,
                                  -2.75573143513906633035e-07, 2.08757232129817482790e-09, -1.13596475577881948265e-11 };
    double n = rint(x * 0.63661977236758134308);
    double r = fma(-n, 6.07710050650619224932e-11, fma(-n, 1.57079632673412561417e+00, x));
    double z = r * r, ps = S[5], pc = Cc[5];
    for (int i = 4; i >= 0; --i) { ps = fma(z, ps, S[i]); pc = fma(z, pc, Cc[i]); }
    double sn = fma(z * r, ps, r), cs = fma(z * z, pc, fma(z, -0.5, 1.0));
    int q = (int)n & 3;
    double v = (q & 1) ? cs : sn;
    return (q & 2) ? -v : v;
}

/* plan row `plan_id` of the generator: out = 30 heights (1D) or the bordered 26x26 plan (2D: 0/1, 3D: 0/6) as the
 * reference holds it; *tb = total_brick as reset() would compute it (2D: floored at 30); returns the cells set / sum */
int orc_make_plan(int dim, int sparse, uint64_t seed, int64_t plan_id, int32_t* out, int32_t* tb) {
    if (dim == 1) {
        double u1 = (double)orc_rng_word(seed, 2u, (uint64_t)plan_id, 0u) * 2.3283064365386963e-10;
        double u2 = (double)orc_rng_word(seed, 2u, (uint64_t)plan_id, 2u) * 2.3283064365386963e-10;
        int k2 = 1 + (int)(((uint64_t)orc_rng_word(seed, 2u, (uint64_t)plan_id, 1u) * 3u) >> 32);
        double k1 = fma(9.0, u1, 3.0), phase = fma(2.0, u2, -1.0) * 3.14159265358979311600;
        int sum = 0;
        for (int x = 0; x < 30; ++x) {
            double arg = 0.20943951023931953 * fma((double)k2, (double)x, phase);
            out[x] = (int32_t)rint(fma(k1, spec_sin(arg), 20.0));
            sum += out[x];
        }
        *tb = sum;
        return sum;
    }
    {
        int32_t img[400];
        int area = 0, thr = sparse ? 20 : 50, amax = dim == 3 ? 110 : 401;
        for (int attempt = 0; attempt < 64; ++attempt) {
            int vx[3], vy[3];
            for (int v = 0; v < 3; ++v) {
                uint32_t w = orc_rng_word(seed, 2u, (uint64_t)plan_id, (uint32_t)(attempt * 4 + v));
                vx[v] = (int)(((w & 0xffffu) * 20u) >> 16);
                vy[v] = (int)(((w >> 16) * 20u) >> 16);
            }
            area = orc_raster_triangle(vx, vy, sparse, img);
            if (area > thr && area < amax) break;
        }
        for (int i = 0; i < 676; ++i) out[i] = 0;
        for (int y = 0; y < 20; ++y)
            for (int x = 0; x < 20; ++x) out[(y + 3) * 26 + x + 3] = img[y * 20 + x] * (dim == 3 ? 6 : 1);
        *tb = dim == 3 ? area * 6 : (area < 30 ? 30 : area);
        return area;
    }
}
