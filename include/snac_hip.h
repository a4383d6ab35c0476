/*
 * snac_hip.h -- C ABI of libsnac_hip.so: the MI355X (gfx950) batched mobile-construction simulator.
 *
 * Drop-in boundary for the env hot path of ai4ce/SNAC.  The reference has no FFI or plugin registry:
 * its "operator API" for this path is the Python class surface
 *     deep_mobile_printing_{1d1r,2d1r,3d1r}.reset()/step()/iou()
 *       Env/1D/DMP_Env_1D_static.py:66-151            Env/1D/DMP_Env_1D_dynamic_usedata_plan.py:40-133
 *       Env/2D/DMP_Env_2D_static.py:54-154            Env/2D/DMP_Env_2D_dynamic_usedata_plan.py:34-147
 *       Env/3D/DMP_simulator_3d_static_circle.py:67-276
 *       Env/3D/DMP_simulator_3d_dynamic_triangle_usedata.py:45-277
 *     VectorizedEnvWrapper.reset()/step()  multiprocess.py:15-32   and its driver loop multiprocess.py:78-84
 * which snac_amd/ re-exports under the same module and class names on top of the entry points below
 * (binding stubs: INTEGRATION.md).  Each entry point names the reference code it replaces.
 *
 * Conventions
 *   - Plain C types only.  Every pointer inside snac_state and every array argument is a DEVICE pointer
 *     owned by the caller (e.g. torch tensors).  The env entry points allocate nothing and keep no state of their own
 *     beyond a thread-local error string; the one exception is the optional trajectory-memory allocator
 *     (snac_traj_alloc / snac_traj_free below), which keeps a mutex-protected table of the blocks it has handed out.
 *     Input and output ARRAYS (actions, step sizes, obs, reward, done) may also lie in
 *     page-locked host memory, which is mapped into the device's address space: the kernels then read / write them over the
 *     bus themselves and a host-side caller only waits (snac_stream_sync) -- no copy command.
 *   - All work is enqueued on the caller's hipStream_t (`stream`, passed as void*; NULL = default
 *     stream), asynchronously, without host synchronisation.
 *   - Return value: SNAC_OK or a negative snac_status; snac_last_error() describes the last failure on
 *     the calling thread.
 *   - One process per GPU; envs are independent, so multi-GPU use shards envs by `env_id_base`.
 *
 * State layout in HBM (N = num_envs, all arrays env-major so one wavefront reads one env's record with
 * unit-stride lanes):
 *   hdr      snac_env_hdr[N]      16-byte packed scalars (one dwordx4 per env)
 *   episode  int32[N]             number of resets performed - 1
 *   grid     1D: int16[N][32]     heights of the 30 interior cells (2 pad); frame cells are implicit -1
 *            2D: uint32[N][20]    occupancy bit-board, row i = interior row i, bit j = interior col j
 *            3D: int16[N][400]    heights of the 20x20 interior, row-major
 *   plans    1D: int16[P][32]  2D: uint32[P][20]  3D: int16[P][400]   (interior cells, same indexing)
 *   plan_tb  int16[P]             total_brick of each plan (2D: after the floor of 30)
 *   stats    int64[N] x 3         finished episodes, sum of their integer returns, sum of
 *                                 llrint(IoU * 2^40) at episode end
 * The -1 frame of the reference's environment_memory is a pure function of the coordinates and is
 * never stored.  Interior values are exactly the reference's (2D cells are {0,1} after every step).
 *
 * Counter RNG (used when `actions` / `step_size` / plan indices are not supplied explicitly)
 *   mix32(x): x^=x>>16; x*=0x7feb352d; x^=x>>15; x*=0x846ca68b; x^=x>>16          (32-bit wrap-around)
 *   key(seed,stream) = mix32(lo32(seed) ^ mix32(hi32(seed) + 0x9E3779B9*(stream+1)))
 *   e0 = mix32(key ^ mix32(lo32(env) + 0x85EBCA6B*hi32(env) + 0x1B873593))
 *   e1 = mix32((key + 0x27D4EB2F) ^ mix32((lo32(env) ^ 0x165667B1) + 0xC2B2AE35*hi32(env)))
 *   word(seed,stream,env,t) = mix32(mix32(e0 ^ (0x9E3779B9*t)) + e1)
 *   stream 0 (per env, tick t):     action = ((word>>16) * num_actions) >> 16
 *                                   step_size = 1 + (((word & 0xffff) * 3) >> 16)
 *   stream 1 (per env, episode e):  plan_idx = (word * num_plans) >> 32
 *   env = env_id_base + local index, so results do not depend on how envs are sharded over GPUs.
 */
#ifndef SNAC_HIP_H
#define SNAC_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SNAC_ABI_VERSION 12

typedef enum snac_status {
    SNAC_OK = 0,
    SNAC_ERR_ARG = -1,         /* bad argument (null pointer, unknown kind, size mismatch) */
    SNAC_ERR_HIP = -2,         /* a HIP call failed; message holds hipGetErrorString */
    SNAC_ERR_UNSUPPORTED = -3
} snac_status;

enum { SNAC_ENV_1D = 1, SNAC_ENV_2D = 2, SNAC_ENV_3D = 3 };
enum { SNAC_OBS_F64 = 0, SNAC_OBS_F32 = 1 };
enum { SNAC_OBS_NONE = 0, SNAC_OBS_ALL = 1, SNAC_OBS_LAST = 2, SNAC_OBS_TILED = 3 };
enum { SNAC_FLAG_NEED_RESET = 1 };   /* snac_env_hdr.flags: the last step returned done */
/* snac_env_desc.rules: the termination tests of the env copies under script/PPO (and script/Rainbow/env/Env2D.py:166), which
 * write `>` where the canonical classes write `>=`:
 *   SNAC_RULE_BRICK_GT  done when count_brick > total_brick   (script/PPO/1d_dynamic/DMP_Env_1D_dynamic_usedata_plan.py:93,
 *                       script/PPO/2d_static/DMP_Env_2D_static.py:137, script/PPO/3d_static/DMP_simulator_3d_static_circle.py:205)
 *   SNAC_RULE_TIME_GT   done when count_step > total_step     (script/PPO/3d_static/DMP_simulator_3d_static_circle.py:221) */
enum { SNAC_RULE_BRICK_GT = 1, SNAC_RULE_TIME_GT = 2 };
/* Observation-layout variants of the reference's env copies, as flags of the same kernels (snac_env_desc.frame_value /
 * obs_scalars / obs_tail).  A row of `obs` is then   [window, scalar, scalar | position | plan | record]   with
 * snac_obs_dim(desc) values; every kernel that writes observations (reset, step, rollout, transition, observe) honours them.
 *   frame_value   value shown for the frame cells of the window (and by snac_export_grid): -1 (0 is read as -1), or 2 --
 *                 Env/2D/DMP_Env_2D_static_Lnet.py:61-64 fills the frame with 2.  1D / 2D only (the 3D rules test -1).
 *   obs_scalars   SNAC_SCALARS_RAW: count_brick, count_step; SNAC_SCALARS_NORM: count_brick/total_brick,
 *                 count_step/total_step; SNAC_SCALARS_DEFAULT: raw for dynamic == 0, normalised for dynamic == 1 (the
 *                 canonical classes).  The L-Net 2D class normalises with a static plan (DMP_Env_2D_static_Lnet.py:75);
 *                 the env copies under script/PPO return raw counters with dataset plans
 *                 (script/PPO/2d_dynamic/DMP_Env_2d_dynamic_usedata_plan.py:70-71).
 *   obs_tail      bit set, appended in this order:
 *     SNAC_TAIL_POSITION  1D: position (Env/1D/DMP_Env_1D_static_Lnet.py:83 -> 8 values); 2D / 3D: row, col
 *     SNAC_TAIL_PLAN      the env's plan, 1D: 30 heights, 2D / 3D: input_plan 20x20 row-major -- the flat observation of
 *                         script/PPO/{1d,2d,3d}_dynamic (37 / 451 values)
 *     SNAC_TAIL_RECORD    8 values: reward, done, pos_r, pos_c, count_brick, count_step, total_brick, plan_idx of the env
 *                         after the step -- everything a single-env caller reads back, in ONE row (one D2H copy).  A row
 *                         written outside a step (snac_reset, also for the envs its mask leaves alone; snac_observe)
 *                         reports reward 0 and done = the env's pending-reset flag */
enum { SNAC_SCALARS_DEFAULT = 0, SNAC_SCALARS_RAW = 1, SNAC_SCALARS_NORM = 2 };
enum { SNAC_TAIL_POSITION = 1, SNAC_TAIL_PLAN = 2, SNAC_TAIL_RECORD = 4 };

/* constants of one env kind: the reference's __init__ blocks (Env/1D/DMP_Env_1D_static.py:7-29,
 * Env/2D/DMP_Env_2D_dynamic_usedata_plan.py:7-32, Env/3D/DMP_simulator_3d_static_circle.py:8-40,
 * Env/3D/DMP_simulator_3d_dynamic_triangle_usedata.py:7-43) */
typedef struct snac_sizes {
    int32_t obs_dim;          /* state_dim: 7 / 51 / 51 */
    int32_t num_actions;      /* action_dim: 3 / 5 / 8 */
    int32_t total_step;       /* 750 / 600 / 1300 (3D static) or 1000 (3D dynamic) */
    int32_t half_window;      /* HALF_WINDOW_SIZE: 2 / 3 / 3 */
    int32_t env_height, env_width;     /* 1 x 34 / 26 x 26 */
    int32_t plan_height, plan_width;   /* 1 x 30 / 20 x 20 */
    int32_t grid_elems, grid_elem_bytes;   /* per-env record of snac_state.grid */
    int32_t plan_elems, plan_elem_bytes;   /* per-plan record of snac_state.plans */
} snac_sizes;

typedef struct snac_env_hdr {   /* 16 bytes, 16-byte aligned */
    int8_t  pos_r, pos_c;       /* position_memory[-1] in bordered coordinates (1D: pos_r, pos_c = 0) */
    uint8_t flags;              /* SNAC_FLAG_* */
    uint8_t reserved;
    int16_t count_brick, count_step, total_brick, plan_idx;
    int16_t ep_return;          /* integer return of the running episode (rewards are integers) */
    int16_t cross;              /* 3D: running sum of min(height, plan) over the interior (numerator of iou()) */
} snac_env_hdr;

typedef struct snac_env_desc {
    int32_t kind;               /* SNAC_ENV_* */
    int32_t dynamic;            /* 0: static-plan class (obs scalars cb, cs); 1: *_usedata class (cb/tb, cs/T) */
    int32_t num_envs;           /* N on this GPU */
    int32_t num_plans;          /* P rows in plans / plan_tb */
    int32_t obs_dtype;          /* SNAC_OBS_F64 (reference dtype) or SNAC_OBS_F32 (= (float) of the f64 value) */
    int32_t static_plan;        /* plan row used by resets when dynamic == 0 or no plan index is supplied */
    uint64_t seed;              /* counter-RNG seed */
    int64_t env_id_base;        /* global id of local env 0 */
    int32_t total_step;         /* time limit, at most 3000; 0 = the class constant (750 / 600 / 1300 static 3D / 1000 dynamic 3D).
                                   The 3D L-Net variant runs the dynamic rules with 1300
                                   (Env/3D/DMP_simulator_3d_static_circle_Lnet.py:28) */
    int32_t rules;              /* SNAC_RULE_* bits; 0 = the canonical classes */
    int32_t frame_value;        /* 0 / -1: the canonical -1; 2: the 2D L-Net frame (1D / 2D only) */
    int32_t obs_scalars;        /* SNAC_SCALARS_* */
    int32_t obs_tail;           /* SNAC_TAIL_* bits */
    int32_t reserved;           /* 0 */
} snac_env_desc;

typedef struct snac_state {
    snac_env_hdr* hdr;          /* [N] */
    int32_t* episode;           /* [N] */
    void* grid;                 /* [N][grid_elems] */
    const void* plans;          /* [P][plan_elems] */
    const int16_t* plan_tb;     /* [P] */
    int64_t* stat_episodes;     /* [N] */
    int64_t* stat_return;       /* [N] */
    int64_t* stat_iou_fx;       /* [N] */
} snac_state;

int snac_version(void);
const char* snac_last_error(void);
/* name of the kernel the calling thread's last launch through this library went to ("k_rollout2d", "k_step3d", "k_rollout" for
 * the tile kernels, ...): diagnostics -- which of the specialised kernels a call took depends on batch size, alignment, layout and
 * the tuning switches, and a measurement should name what it measured (bench.py's roofline.kernel) */
const char* snac_last_kernel(void);
/* the dispatch table: one line "ENV_VARIABLE=value  # what it decides" per batch-size threshold / switch that selects a kernel
 * (effective values: the defaults measured on the build pool, or their environment overrides); tools/retune.py re-measures them */
int snac_tuning(char* out, int32_t cap);

/* constants of (kind, dynamic); replaces the attribute reads of the reference constructors */
int snac_env_sizes(int kind, int dynamic, snac_sizes* out);

/* values per observation row for this descriptor: obs_dim of the kind plus its obs_tail (8 for the 1D L-Net class,
 * 451 for the PPO 2D / 3D dataset classes); negative snac_status on a bad descriptor */
int snac_obs_dim(const snac_env_desc* desc);

/* reset(): Env/2D/DMP_Env_2D_dynamic_usedata_plan.py:34-66 and the five sibling reset()s;
 * VectorizedEnvWrapper.reset / reset_at (multiprocess.py:20-23).
 *   mask        uint8[N] or NULL: reset env i iff mask[i] != 0 (NULL = all)
 *   plan_idx_in int16[N] or NULL: plan row per env (the reference's index_random / sequential index);
 *               NULL = counter RNG stream 1 when dynamic, desc->static_plan otherwise
 *   obs         [N][obs_dim] of obs_dtype or NULL: observation of every env after the call */
int snac_reset(const snac_env_desc* desc, const snac_state* st, const uint8_t* mask, const int16_t* plan_idx_in,
               void* obs, void* stream);

/* step(): Env/2D/DMP_Env_2D_dynamic_usedata_plan.py:85-147 and siblings, for all N envs
 * (VectorizedEnvWrapper.step, multiprocess.py:24-32), fused with observation_/_get_obs and the reward.
 *   t           tick: index of this vector step (keys the counter RNG)
 *   actions     int8[N] or NULL (counter RNG);  step_size int8[N] in {1,2,3} or NULL (counter RNG) --
 *               the value the reference draws with np.random.randint(1, 4) at the top of step()
 *   auto_reset  != 0: an env whose previous step returned done is reset first (plan from the counter RNG)
 *   obs [N][obs_dim] or NULL, reward float[N] or NULL, done uint8[N] or NULL
 * Actions outside [0, num_actions) only advance count_step (the reference raises).  Explicit step sizes are clamped into
 * {1,2,3} -- the only values the reference's randint(1, 4) produces -- so that no input can move an agent off the plan area. */
int snac_step(const snac_env_desc* desc, const snac_state* st, uint32_t t, const int8_t* actions,
              const int8_t* step_size, int auto_reset, void* obs, float* reward, uint8_t* done, void* stream);

/* snac_step with ONE action and ONE step size for every env, passed by value -- the call of a single-env caller
 * (the drop-in classes: env.step(action) with the step size the host drew from np.random, N = 1): no host-to-device
 * copy precedes the launch; with SNAC_TAIL_RECORD the whole result comes back in one row.  action: any int (values outside
 * [0, num_actions) only advance count_step); step_size is clamped into {1,2,3}. */
int snac_step_scalar(const snac_env_desc* desc, const snac_state* st, uint32_t t, int32_t action, int32_t step_size,
                     int auto_reset, void* obs, float* reward, uint8_t* done, void* stream);

/* snac_reset of every env onto plan row `plan_idx`, passed by value (the single-env caller's reset()) */
int snac_reset_scalar(const snac_env_desc* desc, const snac_state* st, int32_t plan_idx, void* obs, void* stream);

/* Block the calling thread until everything enqueued on `stream` has finished (hipStreamSynchronize).  The single-env caller's
 * read-back: `obs` of snac_step_scalar / snac_reset_scalar may point into page-locked host memory (hipHostMalloc, a pinned
 * torch tensor: mapped into the GPU's address space), the kernel then stores its row there itself and this wait is all that
 * separates the launch from reading it -- env.step(action) -> (obs, reward, done) of the reference classes
 * (Env/2D/DMP_Env_2D_dynamic_usedata_plan.py:85-147) as one launch and one wait, no copy command. */
int snac_stream_sync(void* stream);

/* Trajectory memory (optional; every entry point takes any device pointer).  The reference has no counterpart: its driver
 * loop (multiprocess.py:78-84) drops each step's arrays.  On MI355X the physical address space behaves as slices of 32 GiB:
 * write streams confined to one slice reach ~5.7 TB/s, spread over two or more ~7.1 (tools/wr_blocks.hip).
 * A hipMalloc block of 16 GB is one physical run, inside one slice unless it happens to straddle a boundary.  A block from here
 * is ONE contiguous virtual range backed -- through the HIP virtual-memory API -- by 32 MB chunks of physical memory from
 * different slices taking turns, so the [T][N][obs_dim] output of snac_rollout keeps two slices busy at any time: the headline
 * pass takes 2.33-2.40 ms instead of 2.75-2.95 (tools/mem_ab.py).  Which slice a chunk lies in cannot be asked of the driver; it
 * is measured: groups of 16 handles are timed together with a reference group under the rollout's own store pattern -- partners
 * in another slice run 20 % faster -- and the block alternates chunks of the two kinds.
 *   snac_traj_alloc_ex  `bytes` (rounded up to whole 32 MB handles, or 2 MB pages below that; blocks under 1 GiB are one plain
 *                     run) on `device`, read / write for that device; *out is an ordinary device pointer, contiguous, 2 MB-
 *                     aligned.  pool_cap_bytes: how much device memory the measurement may hold BEYOND the block while it runs
 *                     (0 = 64 GiB).  The pool starts at the block + 8 GiB and grows in steps of 8 GiB only until both kinds of
 *                     chunk are there in sufficient number; it never takes more than half of what is free next to the block nor
 *                     the last 4 GiB, and is released before the call returns.  With too little room, or without a usable
 *                     measurement, the block falls back to three runs created 32 GiB apart.  stream: the probe kernels, the
 *                     check and their waits run on this stream (NULL = the default stream); the call returns when they are done.
 *                     How a measured block is built (round 4): the probe writes the headline rollout's own store shape (a wave's
 *                     26 112-byte tile per step, 16 bytes per lane, 1 KiB per store instruction, 1024 waves); the block is a
 *                     sequence of 1 GiB WINDOWS, each the 16 chunks of a group from the reference's slice and the 16 chunks of a
 *                     group from another slice taking turns, and every window is timed as that pair BEFORE it is used (a pair
 *                     that misses the fast level is taken apart); the mapped block is then timed again, EVERY window and once as
 *                     a whole, and rebuilt from a larger pool (twice at most) if a window runs like a single slice.  What was
 *                     measured stays with the block: snac_traj_describe.
 *                     Every block is also CHECKED before it is handed out: a pattern written by one kernel is read back by
 *                     another and, one word per chunk, by a copy.  A block that fails its check is SNAC_ERR_HIP, never a silent
 *                     retry.  Typically 0.1-1 s for the headline's 16 GB.  SNAC_ERR_HIP when memory runs out.
 *                     SNAC_TRAJ_PROBE=0 skips the measurement, SNAC_TRAJ_DEBUG=1 prints it.
 *   snac_traj_alloc   the same with the default pool cap on the default stream.
 *   snac_traj_free    waits for the whole device to go idle (hipDeviceSynchronize: no kernel may still be writing the block),
 *                     unmaps and releases the block's memory.  Its address range stays reserved and is never handed out again (a
 *                     recycled range has been seen to serve stale translations, tools/vmm_stale.hip; a stale pointer faults instead
 *                     of hitting someone else's data): every block of 1 GiB or more costs its own size plus its probe ranges in
 *                     ADDRESS SPACE for the life of the process -- no memory; 47 bits last for more than a thousand headline-sized
 *                     blocks; snac_traj_reserved_bytes() says how much is held that way.  NULL is a no-op; a pointer that did not
 *                     come from snac_traj_alloc is SNAC_ERR_ARG.  (The Python wrapper frees a block when the last tensor viewing
 *                     it dies, so the device-wide wait can come from a garbage collection.)
 * The caller owns the block; the library keeps only what it needs to unmap and to describe it. */
int snac_traj_alloc_ex(size_t bytes, int device, size_t pool_cap_bytes, void* stream, void** out);
int snac_traj_alloc(size_t bytes, int device, void** out);
int snac_traj_free(void* ptr);
/* how a live block of snac_traj_alloc is backed (diagnostics): one of the values below, or SNAC_ERR_ARG for any other pointer */
#define SNAC_TRAJ_ONE_RUN 1      /* below 1 GiB: handles in creation order */
#define SNAC_TRAJ_THREE_RUNS 2   /* the fallback: three runs created 32 GiB apart, chunk j -> run j % 3 */
#define SNAC_TRAJ_MEASURED 3     /* windows of 1 GiB: chunks of the reference group's slice and of another slice in turn, every
                                    window timed as the pair it is */
int snac_traj_layout(const void* ptr);
/* what the allocator measured while it built a live block (all times in microseconds per GiB written under the rollout's store
 * shape; zeros for the layouts that are not measured).  A caller -- bench.py's `placement` -- can tell from this alone whether the
 * block it was given runs at the two-slice level: windows_slow == 0 and block_us_per_gib close to fast_us_per_gib. */
#define SNAC_TRAJ_INFO_WINDOWS 64
typedef struct snac_traj_info {
    int32_t layout;                  /* SNAC_TRAJ_* */
    int32_t rebuilds;                /* measured blocks built and thrown away before this one (0 .. 2) */
    int32_t pool_groups;             /* 512 MB groups the pool held when the block was assembled */
    int32_t probe_launches;          /* launches of the probe kernel for this block (the last attempt) */
    int32_t windows;                 /* 1 GiB windows timed in the finished block */
    int32_t windows_slow;            /* of those: above 1.08 x fast_us_per_gib (0 unless the last rebuild still had one) */
    float self_us_per_gib;           /* the reference group written as a pair with itself: the scale the classes are judged on */
    float fast_us_per_gib;           /* median over the partners in another slice than the reference */
    float slow_us_per_gib;           /* median over the partners in the reference's slice */
    float window_max_us_per_gib;     /* the finished block: its slowest window ... */
    float window_mean_us_per_gib;    /* ... the mean over its windows ... */
    float block_us_per_gib;          /* ... and all of it in one launch */
    float build_ms;                  /* wall time of the whole snac_traj_alloc call */
    uint64_t bytes;                  /* mapped size */
    float window_us[SNAC_TRAJ_INFO_WINDOWS];   /* the first 64 windows, in address order */
} snac_traj_info;
int snac_traj_describe(const void* ptr, snac_traj_info* out);
/* address space (bytes) of ranges this process has unmapped and keeps reserved (see snac_traj_free) */
uint64_t snac_traj_reserved_bytes(void);

/* the driver loop of multiprocess.py:78-84 -- T vector steps with auto-reset, fused in one launch with the
 * env state held on chip.
 *   actions / step_size   int8[T][N] or NULL (counter RNG, ticks t0 .. t0+T-1)
 *   obs_mode              SNAC_OBS_ALL: obs is [T][N][obs_dim]; SNAC_OBS_LAST: obs is [N][obs_dim] and
 *                         receives the last step only; SNAC_OBS_NONE: obs ignored; SNAC_OBS_TILED: every observation, tile-major --
 *                         obs is [ceil(N / 64)][T][64][obs_dim], the row of (t, env) at ((env / 64) * T + t) * 64 + env % 64: each
 *                         tile of 64 envs streams through its own contiguous region instead of jumping N rows per step
 *                         (the build's own layout for trajectories that stay on the GPU: up to 7.1 instead of 6.0 TB/s of
 *                         writes where the tensor lies well; the envs of a ragged last tile beyond N
 *                         are not written)
 *   reward float[T][N] or NULL, done uint8[T][N] or NULL */
int snac_rollout(const snac_env_desc* desc, const snac_state* st, int32_t T, uint32_t t0, const int8_t* actions,
                 const int8_t* step_size, int obs_mode, void* obs, float* reward, uint8_t* done, void* stream);

/* snac_rollout that also records, per env-step, what a replay memory needs besides obs / reward / done (SURVEY.md
 * section 8 row f1): the action taken and the step size used (useful when they come from the counter RNG), the plan
 * row in effect, and whether the step was the first of its episode.  Every member is [T][N] or NULL. */
typedef struct snac_rollout_record {
    int8_t* actions;
    int8_t* step_size;
    int16_t* plan_idx;
    uint8_t* first;
} snac_rollout_record;
int snac_rollout_rec(const snac_env_desc* desc, const snac_state* st, int32_t T, uint32_t t0, const int8_t* actions,
                     const int8_t* step_size, int obs_mode, void* obs, float* reward, uint8_t* done,
                     const snac_rollout_record* rec, void* stream);

/* snac_rollout_rec with SNAC_OBS_TILED into a RING of `ring_ticks` steps: obs is [ceil(N / 64)][ring_ticks][64][obs_dim] and this
 * launch writes the steps first_tick .. first_tick + T - 1 of it (first_tick + T <= ring_ticks; the caller splits a wrap into
 * two launches).  reward / done / rec stay [T][N] (the caller passes the ring's own slices).  The tile-major replay ring of
 * snac_amd.ReplayRing(layout="tiled"): a tile of 64 envs streams through its own contiguous region of the ring. */
int snac_rollout_tiled(const snac_env_desc* desc, const snac_state* st, int32_t T, uint32_t t0, const int8_t* actions,
                       const int8_t* step_size, int32_t ring_ticks, int32_t first_tick, void* obs, float* reward, uint8_t* done,
                       const snac_rollout_record* rec, void* stream);

/* Minibatch assembly for the replay memory of the DQN / DRQN scripts (store_memory / learning_process,
 * script/DQN/2d/DQN_2d_dynamic.py:122-124,145-166): the tuples (s, a, r, s', plan) are not stored, they are gathered from
 * the rollout output ring  obs_ring[cap][N][obs_dim] (obs_dtype)  filled by snac_rollout(_rec) with SNAC_OBS_ALL:
 *   s'   = obs_ring[tick][env]
 *   s    = obs_ring[tick-1 mod cap][env], or the reset observation when first_ring[tick][env] != 0
 *   plan = the env's input_plan (2D / 3D: 20x20, 1D: 30 heights) expanded from the plan table
 * for `batch` samples (tick_idx[b], env_idx[b]); outputs are float32 as the scripts feed them to the networks:
 * s_out / s_next_out [batch][obs_dim], plan_out [batch][400 | 30] or NULL (2D / 3D: 16-byte aligned, it is written four cells at a time).  a, r, done are plain gathers of the [cap][N]
 * arrays and stay with the caller. */
int snac_replay_gather(const snac_env_desc* desc, const snac_state* st, int32_t cap, const void* obs_ring,
                       const uint8_t* first_ring, const int16_t* plan_idx_ring, const int32_t* tick_idx,
                       const int32_t* env_idx, int32_t batch, float* s_out, float* s_next_out, float* plan_out,
                       void* stream);
/* the same from a tile-major ring obs_ring[ceil(N / 64)][cap][64][obs_dim] (filled by snac_rollout_tiled); first_ring and
 * plan_idx_ring stay [cap][N] */
int snac_replay_gather_tiled(const snac_env_desc* desc, const snac_state* st, int32_t cap, const void* obs_ring,
                             const uint8_t* first_ring, const int16_t* plan_idx_ring, const int32_t* tick_idx,
                             const int32_t* env_idx, int32_t batch, float* s_out, float* s_next_out, float* plan_out,
                             void* stream);

/* ---- plan generators: the hindsight classes of the reference draw a fresh random plan per reset --
 * random triangles in 2D / 3D (create_plan, Env/2D/DMP_Env_2D_dynamic_hindsight_replay_usedata.py:37-59: three vertices from
 * np.random.randint(0, 20, size=3) twice, cv2.polylines (+ cv2.fillPoly when dense), redrawn until more than 50 (dense) / 20
 * (sparse) cells are set; the datasets of 400 / 50 / 50 plans were made this way) and random sine curves in 1D (create_plan,
 * Env/1D/DMP_Env_1D_dynamic_hindsight_replay.py:29-42: k1 = uniform(3, 12), k2 = randint(1, 4), phase = uniform(-1, 1) pi,
 * y = round(k1 sin(2 pi / 30 (k2 x + phase)) + 20)).  snac_make_plans writes rows [first, first + count) of st->plans and
 * st->plan_tb (the caller's tables, in the layout of desc->kind) on the device, one wavefront per plan:
 *   vertices NULL   counter RNG stream 2 keyed by (seed, plan_id_base + row): attempt a uses words 4a, 4a+1, 4a+2 (vertex v:
 *                   x = ((word & 0xffff) * 20) >> 16, y = ((word >> 16) * 20) >> 16), redrawn (at most 64 times) until the
 *                   area threshold is passed; 1D: words 0, 1, 2 -> k1 = 3 + 9 u, k2 = 1 + (word * 3 >> 32), phase = (2 u - 1) pi
 *                   with u = word * 2^-32, y = rint(fma(k1, sin(..), 20)) with the sine specified in snac_hip.hip (spec_sin)
 *   vertices        int8[count][6] = x0 y0 x1 y1 x2 y2 (2D / 3D): ONE rasterisation per row, no redraw -- the caller draws the
 *                   vertices (the drop-in classes take them from np.random like the reference) and loops on area_out
 *   sparse          0: outline + interior, threshold 50; 1: outline only, threshold 20; 3D plans also need fewer than 110 cells
 *                   (script/HumanPlayerGUI/env/Env3D.py:360-364, how the 3D datasets were drawn)
 *   area_out        int32[count] or NULL: number of cells set (1D: total_brick); NEGATIVE (-cells) for a row whose 64 redraws were
 *                   all rejected (probability ~ 0): the last triangle stands, with total_brick >= 1
 * Rasteriser: cv2's own rules for this call restated (LineIterator with leftToRight for the outline, the 16.16 fixed-point
 * scanline fill of FillEdgeCollection for dense plans; snac_hip.hip tri_row).  cv2 itself is not available where this was
 * built, but its OUTPUT is: all 1000 2D dataset plans the reference ships, drawn by its authors with this code, are
 * reproduced bit for bit from their vertices (tests/test_plan_generators.py, tests/golden/dataset_triangles.npz).  What stays
 * unpinned is the np.random stream a seeded script sees (the counter RNG draws the vertices here; the drop-in class passes
 * np.random's vertices in).  2D total_brick = max(area, 30); 3D plan = mask * 6, total_brick = 6 area. */
int snac_make_plans(const snac_env_desc* desc, const snac_state* st, int32_t first, int32_t count, int32_t sparse,
                    uint64_t seed, int64_t plan_id_base, const int8_t* vertices, int32_t* area_out, void* stream);

/* ---- hindsight relabelling on the device: the DRQN_hindsight scripts replay a finished episode on a
 * second env whose plan has been overwritten with the episode's own final grid
 * (script/DRQN_hindsight/2d/DRQN_hindsight_2D_dynamic.py:270-282: env_hindsight.plan[3:23, 3:23] =
 *  env_train.environment_memory[3:23, 3:23]; 1D: script/DRQN_hindsight/1d/DRQN_hindsight_1D_static.py:243).  That overwrite as one
 * launch: for i in [0, m)
 *     plan row first + i of st  <-  max(interior of grid i, 0)          (2D: the 0 / 1 board; 1D / 3D: the heights)
 *     st->plan_tb[first + i]    <-  total_brick[i], or (NULL) the total_brick in the header of source env i
 * The grids come EITHER from the packed records of a batch -- src (its grid and hdr; src_envs rows; src_rows int32[m] picks the
 * envs, NULL: env i) -- OR from environment_memory double[m][env_height][env_width] in the reference's own format (frame values
 * are ignored; total_brick is then required).  st and src may be the same batch.  The relabel rollout is then snac_rollout on a
 * batch whose env i was reset onto row first + i with the recorded actions and step sizes (snac_amd/hindsight.py). */
int snac_plans_from_grids(const snac_env_desc* desc, const snac_state* st, int32_t m, int32_t first, const snac_state* src,
                          int32_t src_envs, const int32_t* src_rows, const double* environment_memory,
                          const int32_t* total_brick, void* stream);

/* current observation of every env without stepping: observation_() + the hstack of
 * Env/2D/DMP_Env_2D_dynamic_usedata_plan.py:64-72 */
int snac_observe(const snac_env_desc* desc, const snac_state* st, void* obs, void* stream);

/* iou(): Env/1D/DMP_Env_1D_static.py:138-151, Env/3D/DMP_simulator_3d_static_circle.py:257-276, and the
 * caller-side boolean IoU of the 2D scripts (script/DQN/2d/DQN_2d_dynamic.py:63-71).  out: double[N] */
int snac_iou(const snac_env_desc* desc, const snac_state* st, double* out, void* stream);

/* environment_memory as the reference holds it: out is double[N][env_height][env_width] with the -1 frame */
int snac_export_grid(const snac_env_desc* desc, const snac_state* st, double* out, void* stream);

/* ---- the resident single-env stepper (round 5): what the drop-in classes step through.
 * The reference's scripts drive ONE env, one env.step(action) per loop turn (script/DQN/2d/DQN_2d_dynamic.py:214;
 * Env/2D/DMP_Env_2D_dynamic_usedata_plan.py:85-147 is 9 us of Python per step).  Through snac_step_scalar such a step is one launch and
 * one stream wait (15 us).  A mailbox keeps ONE wavefront resident instead: it polls a doorbell in coherent page-locked host memory,
 * steps the envs of a small batch (N = 1: the drop-in classes; up to 256: an env per lane, 64 envs per wavefront, a launch per wavefront) with the kind's own step rules, writes the observation row (obs_dim values of obs_dtype, layout of desc,
 * tails included) into the mailbox over the bus, acknowledges, and writes the env's state through to st behind the acknowledgement
 * (snac_mailbox_settle waits for that: call it before any other entry point reads or changes st).  The wave leaves by itself after idle_us microseconds without a command
 * (0 = 1000) and on snac_mailbox_quit / _destroy; snac_mailbox_step arms (launches) one when none is resident.
 *   snac_mailbox_touch   the caller has changed st through another entry point (reset, plan row, import ...) and has waited for it:
 *                        the wave reloads the records before its next step
 *   snac_mailbox_step    semantics of snac_step_scalar(desc, st, t, action, step_size, auto_reset = 0, row, NULL, NULL) + a wait;
 *                        episodic sums are updated like snac_step's.  A wave whose launch is still QUEUED (behind kernels that fill the
 *                        device) is waited for, up to SNAC_MAILBOX_TIMEOUT_S seconds (default 120); then the command is WITHDRAWN
 *                        (replaced by a quit of the same sequence number: a wave that starts later leaves without stepping) and the
 *                        call returns SNAC_ERR_HIP with st as the last acknowledged step left it -- or SNAC_OK if the step was served
 *                        while it was being withdrawn.  (A batch of several waves whose launch the limit cut in two: the error string
 *                        names the waves, as a bit mask, whose 64 envs each HAVE taken the step; the others have not.)
 *                        The mailbox belongs to the device that was current at snac_mailbox_create:
 *                        its waves are launched there whatever is current later.  While a thread keeps stepping, a wave stays
 *                        resident: a device-wide synchronisation (hipDeviceSynchronize, hipFree) in ANOTHER thread returns only once the
 *                        stepping pauses for idle_us -- synchronise streams or events there instead
 *   snac_mailbox_row     the row (host pointer, valid until destroy; also a device pointer: the launch path may write it too)
 *   snac_mailbox_stats   out[0] launches, out[1] steps served, out[2] a wave is resident, out[3] idle_us, out[4..7] the last step in ticks of
 *                        the GPU's 100 MHz clock: transition, row stores issued, fence before the acknowledgement, write-through behind it */
typedef struct snac_mailbox snac_mailbox;
int snac_mailbox_create(const snac_env_desc* desc, uint32_t idle_us, snac_mailbox** out);
double* snac_mailbox_row(snac_mailbox* mb);
int snac_mailbox_touch(snac_mailbox* mb);
int snac_mailbox_step(snac_mailbox* mb, const snac_env_desc* desc, const snac_state* st, int32_t action, int32_t step_size);
/* the same for a batch of up to 256 envs (wavefront w takes envs [64 w, 64 w + 64), env 64 w + e on lane e; every wavefront is a launch
 * of its own on a stream of its own and polls the same doorbell) -- the reference's VectorizedEnvWrapper
 * (multiprocess.py:15-32, default --num_envs 3): actions / step_size int8[num_envs] in host memory; rows [num_envs][obs_dim] in
 * snac_mailbox_row, rewards float[num_envs] in snac_mailbox_reward, done flags uint8[num_envs] in snac_mailbox_done */
int snac_mailbox_step_n(snac_mailbox* mb, const snac_env_desc* desc, const snac_state* st, const int8_t* actions, const int8_t* step_size);
float* snac_mailbox_reward(snac_mailbox* mb);
uint8_t* snac_mailbox_done(snac_mailbox* mb);
int snac_mailbox_settle(snac_mailbox* mb);   /* wait until st holds the last acknowledged step (the write-through trails the acknowledgement) */
int snac_mailbox_quit(snac_mailbox* mb);
int snac_mailbox_destroy(snac_mailbox* mb);
int snac_mailbox_stats(const snac_mailbox* mb, uint32_t out[8]);

/* ---- tree search: the MCTS variants of the reference (Env/1D/DMP_Env_1D_{static,dynamic}_MCTS*.py,
 * Env/2D/DMP_ENV_2D_{static,dynamic}_MCTS*.py, Env/3D/DMP_simulator_3d_*_MCTS*.py; nine files) ----
 *
 * transition(state, action, is_model_dynamic) -> (state', obs, reward, done)
 * (Env/2D/DMP_ENV_2D_dynamic_MCTS.py:117-175, Env/1D/DMP_Env_1D_dynamic_MCTS.py:82-139,
 *  Env/3D/DMP_simulator_3d_static_circle_MCTS.py:215-288, Env/3D/DMP_simulator_3d_dynamic_triangle_MCTS.py:195-277;
 *  called once per tree edge by script/MCTS/utils/mcts_Qvalue_dynamic.py:88,118), batched over m edges.
 * The state arrays of `st` are used as a NODE POOL of desc->num_envs rows.  For i in [0, m):
 *     row dst_index[i]  <-  step(row src_index[i], actions[i], step size i)        (index NULL: row i)
 * with the step rules of (kind, dynamic), no auto-reset, and no change to the episodic sums: the running return, episode
 * counter, plan row and total_brick travel with the state; SNAC_FLAG_NEED_RESET records `done`.
 *   actions / step_size   int8[m] or NULL = counter RNG stream 0 keyed by (env_id_base + i, t)
 *   obs                   [m][obs_dim] (obs_dtype) or NULL;  reward float[m] / done uint8[m] or NULL
 * Indices are clamped into the pool.  A destination row must not be the source row of a DIFFERENT edge of the same call
 * (dst_index[i] == src_index[i], i.e. in place, is fine); several edges may share a source. */
int snac_transition(const snac_env_desc* desc, const snac_state* st, int32_t m, const int32_t* src_index,
                    const int32_t* dst_index, uint32_t t, const int8_t* actions, const int8_t* step_size, void* obs,
                    float* reward, uint8_t* done, void* stream);

/* ---- 2D node pools with ONE record per node (round 6).  A tree edge reads its parent at a random row; in the arrays of snac_state
 * that is three lines of memory (header, episode counter, board: the memory side reads whole 128-byte lines) for 100 bytes.  A
 * snac_node2d holds the three in one line: an edge reads one line and writes one.  The pool is caller-owned, 128-byte aligned.
 *   snac_nodes2d_pack        node record node_rows[i] (NULL: i)  <-  batch row rows[i] (NULL: i) of st, i in [0, m)
 *   snac_nodes2d_unpack      the inverse: batch row rows[i] of st  <-  node record node_rows[i]
 *   snac_transition_nodes2d  snac_transition on the pool: record dst_index[i] <- step(record src_index[i], actions[i], step size i);
 *                            st supplies the plan table (plans, plan_tb) only; same rules, outputs and index conventions; the
 *                            canonical observation layout (variants: snac_transition); 2D kinds only */
typedef struct snac_node2d {    /* 128 bytes, 128-byte aligned */
    snac_env_hdr hdr;
    int32_t episode;
    int32_t zero0[3];
    uint32_t board[20];         /* the grid record of the 2D kinds: row word q, bit j = interior cell (q, j) */
    uint32_t zero1[4];
} snac_node2d;
int snac_nodes2d_pack(const snac_env_desc* desc, const snac_state* st, const int32_t* rows, int32_t m, snac_node2d* nodes, int32_t pool_rows,
                      const int32_t* node_rows, void* stream);
int snac_nodes2d_unpack(const snac_env_desc* desc, const snac_node2d* nodes, int32_t pool_rows, const int32_t* node_rows, int32_t m, snac_state* st,
                        const int32_t* rows, void* stream);
int snac_transition_nodes2d(const snac_env_desc* desc, const snac_state* st, snac_node2d* nodes, int32_t pool_rows, int32_t m,
                            const int32_t* src_index, const int32_t* dst_index, uint32_t t, const int8_t* actions, const int8_t* step_size,
                            void* obs, float* reward, uint8_t* done, void* stream);

/* (position, environment_memory, count_brick, count_step) tuples of the reference (the `state` of the MCTS variants,
 * Env/2D/DMP_ENV_2D_dynamic_MCTS.py:88-91) -> pool rows dst_index[i] (NULL: row i); the inverse of snac_export_grid plus
 * the header.  position int32[m][2] (row, col; 1D: position, ignored), count_brick / count_step int32[m],
 * plan_idx int32[m] or NULL (the row keeps its plan), total_brick int32[m] or NULL (plan_tb of the plan row),
 * environment_memory double[m][env_height][env_width].  The running return restarts at 0; values are clamped into the
 * ranges the kernels index with. */
int snac_import_state(const snac_env_desc* desc, const snac_state* st, int32_t m, const int32_t* dst_index,
                      const int32_t* position, const int32_t* count_brick, const int32_t* count_step,
                      const int32_t* plan_idx, const int32_t* total_brick, const double* environment_memory, void* stream);

/* equality_operator(o1, o2) (np.array_equal of two observations, Env/2D/DMP_ENV_2D_dynamic_MCTS.py:254-258; how
 * script/MCTS/utils/mcts_Qvalue_dynamic.py:100-106 recognises an already-expanded child), for m pairs:
 *     out[i] = all(obs_a[idx_a[i]] == obs_b[idx_b[i]])        (index NULL: row i; rows of obs_dim values, obs_dtype)
 * rows_a / rows_b: number of rows of the two arrays (indices are clamped). */
int snac_obs_equal(const snac_env_desc* desc, const void* obs_a, const int32_t* idx_a, int32_t rows_a, const void* obs_b,
                   const int32_t* idx_b, int32_t rows_b, int32_t m, uint8_t* out, void* stream);

/* The "Evaluation" block of the vanilla MCTS procedure (script/MCTS/utils/mcts.py:100-110): from a leaf, default-policy steps until
 * `terminal` or the horizon, `estimate += reward * gamma**t` in that order.  The steps are a snac_rollout of the forked leaves without
 * observation rows (reward / done [H][m]); this call does the sums on the device, one leaf per lane, sequentially in t, each product
 * and each sum rounded to float64 (no fused multiply-add) -- what the reference's python floats do:
 *     alive = !terminal[i];  for t in 0 .. H - 1 while alive:  est[i] += (double)reward[t][i] * gpow[t];  steps[i] += 1;  alive = !done[t][i]
 * est [m]: in = the leaf's first reward (the reference's `estimate = reward`), out = the estimate; steps [m] out (may be NULL);
 * terminal [m] may be NULL (no leaf is terminal); gpow [H] = gamma**t as the CALLER's pow computes it (device memory, like the rest). */
int snac_discounted_return(int32_t H, int32_t m, const float* reward, const uint8_t* done, const uint8_t* terminal, const double* gpow,
                           double* est, int64_t* steps, void* stream);

#ifdef __cplusplus
}
#endif
#endif
