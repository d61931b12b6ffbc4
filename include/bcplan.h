/*
 * bcplan.h -- C ABI of libbcplan.so: batched, MI355X-native PlanEnv.step() (gfx950 HIP kernels).
 *
 * This is the drop-in boundary for the ONE hot path of braincorp/bc-gym-planning-env:
 *     PlanEnv.step()                         envs/base/env.py:334-361
 *       -> _env_step / pose_collides         envs/base/env.py:442-489
 *       -> TricycleRobot.step                robot_models/tricycle_model.py:478-538
 *          DiffDriveRobot.step               robot_models/differential_drive.py:236-265
 *       -> get_pixel_footprint (+fillPoly)   utilities/path_tools.py:122-162
 *       -> ContinuousRewardProvider.reward   envs/base/reward.py:214-259
 * The reference has no FFI layer of its own; what it does have is a set of optional native hooks
 * (`try: from brain.shining_utils... import *_impl`, see below).  Every entry point here cites the
 * reference interface it replaces.  INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - plain C symbols, no exceptions, no torch types.  Return 0 on success, a negative BCP_E_* code
 *     otherwise; bcp_last_error() returns a thread-local message for the last failure.
 *   - Unless a parameter says "host", every pointer is a DEVICE pointer on the handle's GPU
 *     (e.g. torch_tensor.data_ptr()); buffers are caller-owned and must outlive their use.
 *   - All launches are asynchronous on the `stream` argument (a hipStream_t, NULL = default stream).
 *     bcp_step()/bcp_reset_masked() never synchronise or copy to the host and allocate nothing once the first step
 *     has run (the first call after a re-bind uploads a 2 KB parameter block; bcp_egocentric_costmaps() allocates its
 *     scratch on first use).  The step counter (noise stream key, parity of the alternating counter sets) and the
 *     noise seed live on the device and are advanced by the kernels themselves: a step's launch arguments depend only
 *     on the pointers in bcp_step_io and the flags, so steps captured into a hipGraph (after one ordinary step has
 *     uploaded the parameter block) can be replayed.
 *   - One handle per GPU / process rank.  A handle is not thread-safe; different handles are independent.
 *   - There is NO CPU fallback in this library: without a GPU bcp_create() fails with BCP_E_NO_DEVICE.
 */
#ifndef BCPLAN_H
#define BCPLAN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BCP_ABI_VERSION 2
#define BCP_MAX_VERTS 32
#define BCP_LETHAL 254 /* CostMap2D.LETHAL_OBSTACLE, utilities/costmap_2d.py:20-22 */
#define BCP_MAX_KERNEL_HALF 127 /* footprint mask is at most 255x255 px (circumscribed radius / resolution) */

enum { BCP_MODEL_TRICYCLE = 0, BCP_MODEL_DIFFDRIVE = 1 };
/* EnvParams.reward_provider_name (envs/base/reward_provider_examples.py:13-16) */
enum { BCP_REWARD_CONTINUOUS = 0, BCP_REWARD_PURE_PURSUIT = 1 };

enum {
    BCP_OK = 0,
    BCP_E_INVALID = -1,   /* bad argument / unsupported configuration */
    BCP_E_NO_DEVICE = -2, /* no usable GPU: the product has no CPU path */
    BCP_E_HIP = -3,       /* a HIP runtime call failed (message has the hipError string) */
    BCP_E_STATE = -4,     /* call order violated (e.g. step before costmaps/paths/state were bound) */
    BCP_E_INTERNAL = -5   /* bcp_step's watchdog: a bounded hand-off wait inside an EARLIER step launch gave up (see
                             bcp_expired_waits); the steps since then ran, but results of the affected workgroups are
                             unreliable -- a defect of the library, never of the data */
};

/* per-env error bits written to bcp_step_io.err (mirror of the reference's Python exceptions) */
enum {
    BCP_ERR_ANGLE_JUMP = 1, /* path_velocity raises when |dtheta| >= pi, utilities/path_tools.py:319-322 */
    BCP_ERR_TIME_ORDER = 2, /* path_velocity's `assert (dt > 0).all()`, utilities/path_tools.py:307 (bcp_path_velocity only) */
    BCP_ERR_INTERNAL = 4    /* no counterpart in the reference: a wait inside the step kernel ran into its iteration limit (a hand-off
                               between wavefronts never arrived); the step finished, its results for this workgroup are unreliable */
};

/* bcp_step flags */
enum {
    BCP_STEP_AUTO_RESET = 1, /* after outputs are written, envs with done=1 are restored to the bound initial
                                state (what a VecEnv does: scripts/rl_runners/ppo_runner.py:35-36) */
    BCP_STEP_ACTIONS_F32 = 2 /* actions are float32[N,2] (action_space dtype, envs/base/env.py:237-240);
                                otherwise float64[N,2].  float32 is widened to float64 before any arithmetic */
};

/* bcp_params.options */
enum {
    BCP_OPT_DIFFDRIVE_NOISE = 1 /* DiffDriveRobot with noise_parameters raises IndexError in the reference itself
                                   (robot_models/differential_drive.py:73: new_pose[:, 2] on a 1-D pose), so there is no
                                   reference behaviour to match: bcp_create refuses model = BCP_MODEL_DIFFDRIVE with noise_on
                                   (BCP_E_INVALID) unless this bit opts in to the UNPINNED 1-pose analogue of
                                   kinematic_body_pose_motion_step_with_noise */
};

/* POD flattening of EnvParams (envs/base/params.py:14-42), RewardParams (envs/base/reward.py:162-171),
 * the TricycleRobot switches (robot_models/tricycle_model.py:296-300) and the robot constants
 * (robot_models/robot_dimensions_examples.py:108-188).  host struct. */
typedef struct bcp_params {
    int32_t abi_version;            /* = BCP_ABI_VERSION */
    int32_t model;                  /* BCP_MODEL_* */
    int32_t n_verts;                /* footprint vertices, <= BCP_MAX_VERTS */
    int32_t dynamic_model;          /* TricycleRobot._dynamic_model */
    int32_t model_front_column_pid; /* TricycleRobot._model_front_column_pid */
    int32_t noise_on;               /* noise_parameters is not None (env.py:226-232) */
    int32_t iteration_timeout;      /* EnvParams.iteration_timeout */
    int32_t options;                /* BCP_OPT_* bits, 0 = none */
    double verts[BCP_MAX_VERTS][2]; /* metres, robot frame, already multiplied by footprint_scale */
    double dt;
    double front_wheel_from_axis;
    double max_front_wheel_angle;
    double max_front_wheel_speed;
    double max_linear_acceleration;
    double max_angular_acceleration;
    double front_column_p_gain;
    double alpha[6];                /* alpha1..alpha6 of the odometry noise model (differential_drive.py:55-74) */
    double spatial_precision;       /* RewardParams */
    double angular_precision;
    double spatial_progress_multiplier;
    int32_t reward_provider;        /* BCP_REWARD_*: ContinuousRewardProvider (reward.py:174-288) or
                                       ContinuousRewardPurePursuitProvider (reward.py:291-371) */
    int32_t control_delay;          /* EnvParams.control_delay / pose_delay / state_delay (params.py:28-30): FIFO delays */
    int32_t pose_delay;             /* of the action, of State.pose and of State.robot_state (env.py:27-49, 363-398) */
    int32_t state_delay;
} bcp_params;

/* Struct-of-arrays env state, caller-owned device memory, n_envs elements per array.
 * Field names follow State / TricycleRobotState / ContinuousRewardProviderState
 * (envs/base/env.py:52-68, robot_models/tricycle_model.py:234-244, envs/base/reward.py:12-23).
 * `current_time` is not stored: it is the running float64 sum of dt over current_iter steps (env.py:382) and the
 * host layer rebuilds it bit-exactly from current_iter. */
typedef struct bcp_state {
    double *x, *y, *angle;            /* robot pose == State.pose (delays 0) */
    double *v, *w;                    /* measured velocities */
    double *steering_motor_command;   /* tricycle only (may alias a dummy array for diff-drive) */
    double *wheel_angle;              /* tricycle only */
    double *min_spat_dist_so_far;     /* reward provider */
    int32_t *target_idx;              /* reward provider */
    int32_t *current_iter;
    uint8_t *robot_collided;          /* sticky */
    /* Only with delays > 0 (NULL otherwise).  x/y/angle/... above are always the robot's TRUE state.  The k-th element
     * pushed into a queue since the last reset (k = current_iter + 1) lives in slot (k - 1) % delay. */
    double *pose_seen;                /* [3][N] State.pose when pose_delay > 0: what the reward provider and the
                                         observation see (env.py:377-394) */
    double *robot_state_seen;         /* [7][N] State.robot_state when state_delay > 0 (field order as above) */
    double *control_queue;            /* [control_delay][2][N]  State.control_queue */
    double *poses_queue;              /* [pose_delay][3][N]     State.poses_queue */
    double *robot_state_queue;        /* [state_delay][7][N]    State.robot_state_queue */
} bcp_state;

/* per-step inputs/outputs, device pointers, N = n_envs */
typedef struct bcp_step_io {
    const void *actions;     /* [N,2] (wheel_v, wheel_angle) tricycle / (v, w) diff-drive; dtype by flag */
    const double *noise_z;   /* [N,3] standard normals in slot order, or NULL: then, if params.noise_on, normals
                                come from the on-device Philox4x32-10 stream keyed (seed, env, step counter) */
    double *noise_z_out;     /* optional [N,3]: the normals this step used (NaN where no draw happened) */
    double *reward;          /* [N] */
    uint8_t *done;           /* [N] goal reached | timed out | robot_collided (env.py:407-419) */
    uint8_t *collided_now;   /* optional [N]: this step's pose_collides verdict (return of _env_step) */
    int32_t *err;            /* optional [N]: BCP_ERR_* bits */
} bcp_step_io;

typedef struct bcp_handle bcp_handle;

/* ---- lifetime ----------------------------------------------------------------------------------------- */
const char *bcp_last_error(void);
int bcp_abi_version(void);
/* replaces PlanEnv.__init__ (env.py:219-249) for a batch of n_envs envs on GPU `device`.
 * env_id_base: global index of this handle's env 0 (rank * n_envs when sharded); keys the RNG stream. */
int bcp_create(const bcp_params *params /*host*/, int64_t n_envs, int device, int64_t env_id_base, bcp_handle **out);
int bcp_destroy(bcp_handle *h);
/* PlanEnv.seed / np.random.seed for the noise stream (differential_drive.py:50 uses the global numpy RNG) */
int bcp_seed(bcp_handle *h, uint64_t seed);

/* Execution knobs; none of them changes any result (tests run every combination against the oracle).
 *   BCP_TUNE_EXACT_MODE       0 = auto, 1 = always the wave-cooperative exact rasteriser, 2 = always the per-thread one,
 *                             3 = wave-cooperative, cell by cell (the lethal cells under the image tested one by one)
 *   BCP_TUNE_DENSE_THRESHOLD  auto mode: more undecided poses than this in one wavefront -> settle them inside the
 *                             step kernel (cooperatively with a distance field, per thread without one)
 *   BCP_TUNE_CULL             0 = skip the distance-field pre-classification (every in-map pose is rasterised)
 *   BCP_TUNE_DEFER            0 = settle undecided poses inside the step kernel instead of the second, load-balanced
 *                             kernel (only relevant with a distance field and exact mode auto)
 *   BCP_TUNE_EDT_LDS          0 = build distance fields with the two-pass global-memory kernels even where a map fits
 *                             into LDS (takes effect at the next bcp_set_costmaps)
 *   BCP_TUNE_FUSED            0 = settle the parked poses in a second launch (step_fast_pair_kernel + step_pending_kernel)
 *                             instead of inside the step launch itself (step_local_kernel, the default)
 *   BCP_TUNE_EGO_SPARSE       0 = egocentric views always sample the costmap pixel by pixel; 1 (default) = sparse maps with
 *                             border value 0 are drawn as a zero fill plus one patch per non-zero source cell, "sparse"
 *                             decided by a cost model from the largest number of non-zero cells any map entry holds;
 *                             >= 2 = the same with this explicit limit of cells per map (tests, sweeps)
 *   BCP_TUNE_NEAR_DILATE      how the 1-bit tiles of the distance field (bcp_get_near_field) are made: 0 = always by
 *                             thresholding the uint8 field; 1 (default) = a pool refresh (bcp_refresh_mini_worlds) under the
 *                             single-launch step dilates the lethal mask by the sample disc instead and leaves the uint8
 *                             fields of those entries to be computed when something asks for them (the other step forms,
 *                             bcp_pose_collides, bcp_get_distance_field); 2 = bcp_set_costmaps also overwrites its
 *                             thresholded tiles with dilated ones (tests: the two must agree bit for bit)
 *   BCP_TUNE_LOCAL_PAIRS      workgroup size of the single-launch step: 4 = 16 wavefronts / 256 envs (one workgroup per CU),
 *                             2 = 8 wavefronts / 128 envs (two per CU), 1 = 4 wavefronts / 64 envs (four per CU); 0 (default)
 *                             = the library's choice for the configuration.  Same results bit for bit in every size.
 *                             (The environment variable BCP_LOCAL_PAIRS = 1 | 2 | 4, read by bcp_create, sets this knob's
 *                             initial value for every handle of the process: the whole test suite runs under each size.)
 *   BCP_TUNE_EGO_LIST_STRIDE  0 (default) = the cell lists of the sparse egocentric route are sized from the counted cells;
 *                             a multiple of 64 = this many cells per map entry: an entry with more is drawn pixel by pixel
 *                             inside the same launch (what happens to a pool entry that is re-sampled with more cells than
 *                             the lists were sized for; the knob lets tests reach that path)
 *   BCP_TUNE_NEAR_SHIFT       resolution of the 1-bit tiles the single-launch step classifies poses on when the costmaps are
 *                             private: 0 = the tiles of bcp_get_near_field, 1 / 2 / 3 = a copy at 1/2, 1/4, 1/8 of the resolution (a
 *                             bit is the OR of the 2 x 2 / 4 x 4 / 8 x 8 cells it stands for: fewer lines of memory per pose, a few more
 *                             poses left to the exact test); -1 (default) = the library's choice.  In force from the next
 *                             bcp_set_costmaps on; results are the same bit for bit.  (Environment variable BCP_NEAR_SHIFT,
 *                             read by bcp_create: the knob's initial value.) */
enum { BCP_TUNE_EXACT_MODE = 0, BCP_TUNE_DENSE_THRESHOLD = 1, BCP_TUNE_CULL = 2, BCP_TUNE_DEFER = 3, BCP_TUNE_EDT_LDS = 4,
       BCP_TUNE_FUSED = 5, BCP_TUNE_EGO_SPARSE = 6, BCP_TUNE_NEAR_DILATE = 7, BCP_TUNE_LOCAL_PAIRS = 8,
       BCP_TUNE_EGO_LIST_STRIDE = 9, BCP_TUNE_NEAR_SHIFT = 10 };
int bcp_set_tuning(bcp_handle *h, int32_t key, int32_t value);

/* ---- static per-episode inputs ------------------------------------------------------------------------ */
/* Geometry pool: RandomMiniEnv.reset() with draw_new_turn_on_reset (envs/mini_env.py:469-481) as a batched, masked
 * re-initialisation from n_geoms pre-generated (costmap, path, initial state) entries -- see
 * bc_gym_planning_env_amd/mini_env.py for the sampler (mini_env.py:269-361).  With a pool in effect every
 * NON-shared array given to bcp_set_costmaps / bcp_set_paths / bcp_bind_initial_state has n_geoms entries instead of
 * n_envs, and env i uses entry geom_of_env[i].
 *   geom_of_env  int32 [N], caller-owned device memory, values in [0, n_geoms); read by every step and REWRITTEN by
 *                resets (BCP_STEP_AUTO_RESET and bcp_reset_masked): a reset env moves to next_geom[geom_of_env[i]]
 *                and takes that entry's initial state.
 *   next_geom    int32 [n_geoms] successor table (e.g. g+1 within an env's chain of sampled geometries, or any
 *                permutation), device memory; NULL = stay on the same entry (draw_new_turn_on_reset=False).
 * Call before the arrays are set (changing the entry count invalidates them); n_geoms = 0 turns the pool off. */
int bcp_set_geometry_pool(bcp_handle *h, int32_t n_geoms, int32_t *geom_of_env, const int32_t *next_geom);
/* (bcp_plan_mini_worlds / bcp_release_mini_worlds are the only calls that write next_geom) */
/* (below, N = n_envs, or n_geoms when a geometry pool is in effect)
 * CostMap2D (utilities/costmap_2d.py:13-37).  data: uint8 [rows, cols] when shared, else [N, rows, cols]
 * (row-major, `rows`/`cols` is the padded allocation).  valid_rows/valid_cols (optional, [N] int32) give each
 * env's true map shape for the bounds test of env.py:483-484; NULL => rows/cols.  origins: host double[2] when
 * origins_per_env == 0, else device double [N,2].  Builds the library-owned 1-bit lethal mask
 * (cell == 254) that the step kernel reads and, for a shared map, the distance transform of the lethal cells used
 * to settle most poses without rasterising; call again whenever the costmap content changes. */
int bcp_set_costmaps(bcp_handle *h, const uint8_t *data, int32_t rows, int32_t cols, int32_t shared,
                     const int32_t *valid_rows, const int32_t *valid_cols, const double *origins,
                     int32_t origins_per_env, double resolution, void *stream);

/* Introspection (tests, debugging): the distance field bcp_set_costmaps derived for the pre-classification of poses --
 * floor(min(clamp, Euclidean distance in cells to the nearest lethal cell)) over the map padded by `pad` cells.
 * shape (host int32 [4]) receives {rows + 2 pad, cols + 2 pad, pad, clamp}; out (device uint8 [n_entries][shape0][shape1],
 * or NULL to ask for the shape only) the fields of entries first_entry .. first_entry + n_entries - 1 (entry 0 of a
 * shared map). */
int bcp_get_distance_field(bcp_handle *h, int64_t first_entry, int64_t n_entries, uint8_t *out, int32_t *shape /*host*/,
                           void *stream);
/* The same field as the step's outer test reads it: one bit per cell, set where the field value is < t_out (a lethal
 * cell may touch a footprint whose sample disc is centred there), in tiles of 32 x 32 cells -- bit (x & 31) of word
 * ((y >> 5) * tiles_x + (x >> 5)) * 32 + (y & 31) is cell (y, x) of the padded field.  shape (host int32 [3]) receives
 * {tiles_y, tiles_x, t_out}; out (device uint32 [n_entries][tiles_y * tiles_x * 32], or NULL for the shape only). */
int bcp_get_near_field(bcp_handle *h, int64_t first_entry, int64_t n_entries, uint32_t *out, int32_t *shape /*host*/,
                       void *stream);
/* Static path of the reward provider (ContinuousRewardProviderState.path, reward.py:17-18), already refined.
 * xytheta: double [max_len,3] when shared, else [N,max_len,3]; lens: NULL when shared (then len = max_len) else
 * int32 [N].  Precomputes cos/sin of the waypoint headings (path_tools.py:405) on the device. */
int bcp_set_paths(bcp_handle *h, const double *xytheta, const int32_t *lens, int32_t max_len, int32_t shared,
                  void *stream);

/* ---- state -------------------------------------------------------------------------------------------- */
/* PlanEnv.set_state / get_state (env.py:278-291): the library reads and writes the caller's SoA arrays in
 * place, so "get_state" is reading these tensors and "set_state" is writing them. */
int bcp_bind_state(bcp_handle *h, const bcp_state *state /*host struct of device pointers*/);
/* PlanEnv._initial_state (env.py:247): the snapshot reset()/auto-reset restores (one entry per env, or per pool
 * geometry). */
int bcp_bind_initial_state(bcp_handle *h, const bcp_state *initial /*host struct of device pointers*/);
/* PlanEnv.reset (env.py:293-303) for every env with mask[i] != 0 (mask NULL = all). */
int bcp_reset_masked(bcp_handle *h, const uint8_t *mask, void *stream);

/* Monte-Carlo fan-out ("you need to run many rollouts from one state", README.md:45-61): every env with mask[i] != 0
 * (mask NULL = all) takes over env `src`'s complete state -- robot, reward provider, iteration counter, collision
 * flag, geometry-pool entry, delay queues.  The batched form of `s = env.get_state(); other.set_state(s)`. */
int bcp_broadcast_state(bcp_handle *h, int64_t src, const uint8_t *mask, void *stream);

/* ---- the hot path ------------------------------------------------------------------------------------- */
/* PlanEnv.step (env.py:334-361) for all envs: one fused kernel launch. */
int bcp_step(bcp_handle *h, const bcp_step_io *io /*host struct*/, uint32_t flags, void *stream);

/* ---- operator-level seams (the reference's optional native hooks) -------------------------------------- */
/* IRobot.step for n robots (tricycle_model.py:478 / differential_drive.py:236): kinematics only, no collision.
 * state7_io: [7][n] SoA {x,y,angle,v,w,steering_motor_command,wheel_angle}; actions float64 [n,2]. */
int bcp_robot_step(bcp_handle *h, double *state7_io, int64_t n, const double *actions, const double *noise_z,
                   int32_t *err, void *stream);
/* pose_collides(x, y, angle, robot, costmap) (env.py:464-489; twin costmap_utils.py:178-203) for n poses
 * [n,3]; pose i is tested against env (i % n_envs)'s current costmap.  out: uint8 [n]. */
int bcp_pose_collides(bcp_handle *h, const double *poses, int64_t n, uint8_t *out, void *stream);
/* is_robot_colliding(robot_pose, footprint, costmap_data, origin, resolution) (utilities/costmap_utils.py:106-164):
 * pose_collides, except that a robot whose own pixel lies outside the map never collides (in_costmap_bounds, :127-130).
 * Same arguments as bcp_pose_collides. */
int bcp_is_robot_colliding(bcp_handle *h, const double *poses, int64_t n, uint8_t *out, void *stream);
/* is_footprint_colliding_impl(image_slice, blit_mask, lethal) (the native hook of utilities/costmap_utils.py:106-136;
 * Python fallback :163-164: np.any(values == LETHAL)) for n (slice, mask) pairs of one shape: image_slices, blit_masks
 * uint8 [n, rows, cols] (mask: non-zero = footprint cell); out uint8 [n] = any(slice[mask] == lethal). */
int bcp_is_footprint_colliding(bcp_handle *h, const uint8_t *image_slices, const uint8_t *blit_masks, int64_t n,
                               int32_t rows, int32_t cols, uint8_t lethal, uint8_t *out, void *stream);
/* The reward-provider seam (envs/base/reward.py:184-259): reward_provider.reward(state) and .done(state) for n
 * (pose, provider state) pairs against the paths given to bcp_set_paths (pose i: the path of env i % n_envs), with this
 * handle's provider (RewardParams / BCP_REWARD_*).  poses double [n,3]; min_spat_dist_so_far double [n] and target_idx
 * int32 [n] are the provider state, updated in place; robot_collided uint8 [n] or NULL (only the pure-pursuit provider
 * reads it); reward double [n]; goal_reached uint8 [n] or NULL = provider.done(). */
int bcp_reward(bcp_handle *h, const double *poses, int64_t n, double *min_spat_dist_so_far, int32_t *target_idx,
               const uint8_t *robot_collided, double *reward, uint8_t *goal_reached, void *stream);
/* find_last_reached(pose, segment, spatial_precision, angular_precision) (utilities/path_tools.py:432-448) for n poses
 * against the bound paths: out int32 [n] = index of the last way point reached, -1 for None. */
int bcp_find_last_reached(bcp_handle *h, const double *poses, int64_t n, int32_t *out, void *stream);
/* path_velocity(path) (utilities/path_tools.py:298-323): path_txyth double [n_rows,4] rows of (t, x, y, angle) ->
 * v, w double [n_rows - 1]; err int32 [n_rows - 1] or NULL: BCP_ERR_* bits where the reference raises / asserts. */
int bcp_path_velocity(bcp_handle *h, const double *path_txyth, int64_t n_rows, double *v, double *w, int32_t *err,
                      void *stream);
/* get_pixel_footprint_impl(angle, footprint, resolution, fill=True) (path_tools.py:101-162) for n angles.
 * masks: uint8 [n, side, side] (side >= 2*half+1 for every angle), zero-filled then 255 inside; the kernel
 * image of angle i occupies the top-left shape_hw[i] = {2*half_y+1, 2*half_x+1} corner. */
int bcp_pixel_footprint(bcp_handle *h, const double *angles, int64_t n, double resolution, uint8_t *masks,
                        int32_t side, int32_t *shape_hw, void *stream);
/* Done masks as bits, for a job sharded over several GPUs (bc_gym_planning_env_amd/distributed.py: DoneGather(packed=True),
 * bench.py): the only data that crosses GPUs is each rank's done mask, and it crosses as one bit per env --
 * bits[w] bit b = (mask[32 w + b] != 0), n envs -> (n + 31) / 32 words -- an eighth of the bytes an RCCL all-gather has to
 * move while it shares the compute units with the steps.  Device pointers of the current device, asynchronous on `stream`,
 * no handle.  bcp_unpack_mask_bits writes zeros and ones.  No counterpart in the reference (its envs are independent objects,
 * envs/base/env.py; its only vectorised call site, scripts/rl_runners/ppo_runner.py:35-36, runs them in one process). */
int bcp_pack_mask_bits(const uint8_t *mask, int64_t n, uint32_t *bits, void *stream);
int bcp_unpack_mask_bits(const uint32_t *bits, int64_t n, uint8_t *mask, void *stream);

/* normalize_angle_impl (coordinate_transformations.py:17-36) */
int bcp_normalize_angle(bcp_handle *h, const double *in, double *out, int64_t n, void *stream);
/* world_to_pixel_impl (coordinate_transformations.py:169-205): xy [n,2] -> int64 [n,2]; origin host double[2] */
int bcp_world_to_pixel(bcp_handle *h, const double *xy, int64_t n, const double *origin, double resolution,
                       int64_t *out, void *stream);

/* ---- egocentric observation (what envs/egocentric.py:102-160 feeds a policy) ------------------------------ */
/* extract_egocentric_costmap(costmap, pose, resulting_origin, resulting_size, border_value)
 * (utilities/costmap_utils.py:25-75: cv2.getRotationMatrix2D + cv2.warpAffine, INTER_NEAREST) for all envs in one
 * launch: image i = the costmap of env (i % n_envs) seen from poses[i] ([n,3] device doubles; NULL = the bound
 * state's current poses, n = n_envs), robot at (0, 0) heading along +x.  window_origin / window_size: host double[2] in metres (both or neither; NULL = output has
 * the costmap's shape and only the rotation is applied).  out: uint8 [n, shape_hw[0], shape_hw[1]] as reported by
 * bcp_egocentric_shape.  Reads the RAW uint8 costmaps last given to bcp_set_costmaps, which must still be alive
 * (shared maps: the allocation must be readable up to the next multiple of 4 bytes). */
int bcp_egocentric_shape(bcp_handle *h, const double *window_size /*host, or NULL*/, int32_t *shape_hw /*host [2]*/);
int bcp_egocentric_costmaps(bcp_handle *h, const double *poses, int64_t n, const double *window_origin /*host*/,
                            const double *window_size /*host*/, uint8_t border_value, uint8_t *out, void *stream);
/* Which kernel the last bcp_egocentric_costmaps call of this handle ran (no reference counterpart: measurement and tests
 * name the route instead of guessing it).  info4 (host int32[4]): BCP_EGO_* kernel, largest number of non-zero cells of
 * any map entry (-1: not counted), stride of the cell lists, the limit of cells the route decision compared it with. */
enum { BCP_EGO_NONE = 0, BCP_EGO_SPARSE = 1 /* ego_sparse_kernel */, BCP_EGO_STAGED = 2 /* ego_costmap_kernel, shared map in LDS */,
       BCP_EGO_BINNED = 3 /* ego_costmap_binned_kernel */, BCP_EGO_WINDOW = 4 /* ego_costmap_window_kernel */,
       BCP_EGO_GLOBAL = 5 /* ego_costmap_kernel sampling global memory */ };
int bcp_egocentric_route(bcp_handle *h, int32_t *info4 /*host*/);
/* EgocentricCostmap.observation's `goal_n_state` (envs/egocentric.py:140-160) for all envs: the next way point in
 * the robot frame (from_global_to_egocentric, coordinate_transformations.py:341-362) with its position divided by
 * world_size (host double[2] = CostMap2D.world_size() of the egocentric map) and clipped to [-1, 1], followed by
 * robot_state.to_numpy_array().  out: float32 [N, 3 + 6] (tricycle) / [N, 3 + 5] (diff-drive); zeros for envs whose
 * path is exhausted. */
int bcp_goal_n_state(bcp_handle *h, const double *world_size /*host*/, float *out, void *stream);
/* ColoredEgoCostmapRandomAisleTurnEnv's `goal` vector (envs/synth_turn_env.py:412-420) for all envs: the LAST way
 * point in the robot frame divided by world_size and normalised to unit length, then the robot's egocentric state
 * (v, w, wheel_angle; 0 for a diff-drive robot).  out: float64 [N, 5]. */
int bcp_goal_direction_state(bcp_handle *h, const double *world_size /*host*/, double *out, void *stream);

/* ---- RandomMiniEnv worlds sampled on the device ------------------------------------------------------------ */
/* RandomMiniEnvParams (envs/mini_env.py:30-47) + the EnvParams fields the sampler reads.  host struct. */
typedef struct bcp_mini_world_params {
    double inner_h, inner_w, mid_margin, out_margin;
    double min_obstacle_angle, max_obstacle_angle;
    double lim_euc_dist, lim_ang_dist, angular_pose_noise_scale;
    double resolution;      /* EnvParams.resolution */
    double goal_spat_dist;  /* EnvParams.goal_spat_dist / goal_ang_dist: start and goal must not be this close */
    double goal_ang_dist;
} bcp_mini_world_params;
/* numpy.random.RandomState(seed) for n_chains independent streams: mt_state is uint32 [n_chains][625] device memory
 * (624 MT19937 words + the position), seeds int64 [n_chains] device memory (0 <= seed < 2^32, mt19937_seed). */
int bcp_mini_world_seed(bcp_handle *h, const int64_t *seeds, int64_t n_chains, uint32_t *mt_state, void *stream);
/* _sample_mini_env_params (envs/mini_env.py:328-359) `episodes` times in a row for every stream, one wavefront per
 * stream: same draw order as the reference, two 1-px walls (cv2.line), pose_collides of both path ends with this
 * handle's footprint, "not too close" test; the streams continue where they stopped.  Outputs (device):
 *   worlds  double [n_chains * episodes][14] = start(3), end(3), obstacle_a(2), obstacle_o(2), obstacle_b(2), h, w
 *   maps    uint8  [n_chains * episodes][rows][cols]  (rows, cols as CostMap2D.create_empty gives them for (h, w))
 *   status  int32  [n_chains]  0, or 1 where the reference would raise "the sampling space looks empty"
 * Transcendentals are the device's: a world can differ from numpy's in the last bit of a coordinate. */
int bcp_sample_mini_worlds(bcp_handle *h, const bcp_mini_world_params *p /*host*/, uint32_t *mt_state, int64_t n_chains,
                           int32_t episodes, int32_t rows, int32_t cols, double *worlds, uint8_t *maps, int32_t *status,
                           void *stream);

/* make_initial_state (envs/base/env.py:179-214) for sampled worlds, on the device: refine_path of each world's coarse
 * (start, end) path (utilities/path_tools.py:178-240) and the initial state of this handle's reward provider
 * (reward.py:261-288 / :355-371).  worlds as bcp_sample_mini_worlds writes them; paths: double [n_worlds][max_len][3];
 * lens: int32 [n_worlds]; init: double [n_worlds][2] = (min_spat_dist_so_far, target_idx); status: int32 [n_worlds],
 * 0 = ok, 1 = the refined path does not fit max_len, 2 = "Goal pose too close to initial pose". */
int bcp_mini_world_paths(bcp_handle *h, const double *worlds, int64_t n_worlds, double path_delta, int32_t max_len,
                         double *paths, int32_t *lens, double *init, int32_t *status, void *stream);

/* A pool that never runs out: RandomMiniEnv.reset() draws a NEW world every time (envs/mini_env.py:441-459), a pool
 * of `episodes` entries per stream would wrap around.  For a handle whose geometry pool has n_envs x episodes entries,
 * env c on the entries c * episodes .. + episodes - 1 of stream c (mt_state [n_envs][625], worlds / maps / paths / lens
 * / init as produced by bcp_sample_mini_worlds + bcp_mini_world_paths and handed to bcp_set_costmaps / bcp_set_paths /
 * bcp_bind_initial_state), three calls top the pool up behind the envs without a host round trip.  World number j of
 * stream c lives in entry c * episodes + j % episodes; next_geom (the array given to bcp_set_geometry_pool, WRITTEN
 * here) is a ring with one guard: the entry of the newest world points to itself, so an env that gets there before new
 * worlds are ready repeats that world instead of wrapping onto an old one (set next_geom[c * episodes + episodes - 1]
 * to itself before the first step).
 *   1. bcp_plan_mini_worlds     looks where every env is, marks the entries of the worlds it has left as free and
 *                               closes the ring behind the last of them (the next guard).
 *        generated  int64 [n_envs] device, in/out: worlds drawn from each stream so far (`episodes` after the first fill)
 *        info       int32 [4] device, out: {entries freed, envs found waiting on their guard entry, 0, 0}
 *   2. bcp_refresh_mini_worlds  samples the next worlds of each stream into the free entries: costmap, path, lens, init
 *                               and initial state are rewritten in place and the handle's derived data (lethal masks,
 *                               distance fields, path index) is rebuilt for exactly those entries.
 *        status  int32 [n_envs]: as bcp_sample_mini_worlds;  path_status int32 [n_envs * episodes]: as bcp_mini_world_paths
 *   3. bcp_release_mini_worlds  opens the previous guard: the envs can walk on into the new worlds.
 * 1 and 3 belong on the stream of the steps.  Sampling a world is a latency-bound job for one wavefront (~0.3 ms, with
 * a long tail), so 2 may run on a SIDE stream while steps go on -- they neither read nor reach an entry that is being
 * rewritten: make the side stream wait for 1 and the stream of the steps wait for 2 before 3 (events).  With one
 * plan-refresh-release round per `episodes - 1` episodes of the fastest env no world is ever repeated. */
int bcp_plan_mini_worlds(bcp_handle *h, int32_t episodes, int64_t *generated, int32_t *info, void *stream);
int bcp_refresh_mini_worlds(bcp_handle *h, const bcp_mini_world_params *p /*host*/, uint32_t *mt_state, double *worlds,
                            uint8_t *maps, double *paths, int32_t *lens, double *init, double path_delta, int32_t *status,
                            int32_t *path_status, void *stream);
int bcp_release_mini_worlds(bcp_handle *h, void *stream);

/* ---- the on-device noise stream (introspection) --------------------------------------------------------------- */
/* A stream of the handle that may only use `cu_percent` % of the device's compute units, spread evenly over the chip
 * (hipExtStreamCreateWithCUMask).  For work that runs BESIDE the steps: the single-launch step wants whole compute units
 * (one 1024-thread workgroup each), and behind a kernel of small long-running workgroups on an ordinary side stream -- the
 * world sampler of bcp_refresh_mini_worlds -- a step launch can wait until that kernel has drained (measured: 18 ms).  On a
 * masked stream such a kernel leaves the other compute units to the steps.  Measured in round 3: with 50 %
 * the endless pool ran SLOWER (0.078 against 0.064 ms per step) -- the refresh takes twice as long and the steps' own
 * workgroups still land on the masked units too -- so the Python layer defaults to an ordinary stream; the entry point is
 * kept for callers whose side work is lighter.  The stream belongs to the handle (destroyed
 * with it; asking for another share replaces it after a synchronisation).  BCP_E_HIP when the runtime refuses: use an
 * ordinary stream.  No counterpart in the reference. */
int bcp_side_stream(bcp_handle *h, int32_t cu_percent, void **stream);

/* n_steps steps in one call, for callers that hold the actions of a whole rollout (open-loop Monte-Carlo rollouts from one
 * state, the use the reference documents in its README; no counterpart function in the reference, whose PlanEnv.step takes
 * one action).  Every array of bcp_step_io has a leading [n_steps] dimension: actions [n_steps][N][2], noise_z /
 * noise_z_out [n_steps][N][3], reward / done / collided_now / err [n_steps][N]; row k is what the k-th of n_steps calls of
 * bcp_step would read and write, and the state ends where those calls would leave it -- bit for bit, in-kernel resets and
 * the on-device noise stream included.  With the single-launch step form (bcp_step_form() == 3) the steps are ONE launch:
 * the workgroups advance independently and launch, argument fetch and staging are paid once; other forms are stepped one
 * launch at a time. */
int bcp_rollout(bcp_handle *h, const bcp_step_io *io, int32_t n_steps, uint32_t flags, void *stream);

/* Health of the step kernel's internal hand-offs.  The single-launch step passes undecided poses between the wavefronts of a
 * workgroup through LDS; every wait on such a hand-off is bounded (~10^7 cycles against the ~2 * 10^4 a step lasts).  A
 * wait that runs into its limit gives up -- the step still finishes; an env whose verdict never arrived is finished as
 * "free" with BCP_ERR_INTERNAL in `err` -- and is counted here.  *count = waits that gave up since bcp_create (host
 * pointer; the call synchronises `stream`).  Anything but 0 means a defect of the library, never of the caller's data.
 * bcp_step also watches this counter by itself, without ever waiting for the GPU: every 256 calls it copies the counter to
 * pinned host memory behind the step it has just launched, and a later bcp_step that finds the copy landed and the counter
 * grown returns BCP_E_INTERNAL (once per growth; the step of that call has been launched all the same).  Steps replayed
 * from a captured hipGraph do not pass through bcp_step: such callers ask here.  No counterpart in the reference. */
int bcp_expired_waits(bcp_handle *h, int64_t *count, void *stream);
/* Poses the single-launch step (step_local_kernel) could not clear by its distance-field classification and handed to the
 * exact footprint test (pose_collides proper, envs/base/env.py:464-489), summed over all steps since bcp_create: the
 * "parked-pose fraction" of a workload is the growth of this counter over a run / (n_envs * steps).  Host pointer; the call
 * synchronises `stream`.  The other step forms do not count.  No counterpart in the reference (measurement only). */
int bcp_parked_poses(bcp_handle *h, int64_t *count, void *stream);

/* The standard normals a step draws when bcp_step_io.noise_z is NULL (the stand-in for np.random.normal of
 * robot_models/differential_drive.py:43-52): out double [n_steps][n_envs][3] = the three slots of envs first_env ..
 * first_env + n_envs - 1 of this handle at step counters first_step .. first_step + n_steps - 1, for the handle's seed
 * and env_id_base.  (A step consumes slot k only when variance_k > 0, like the reference.)  Any n_envs, not only the
 * handle's. */
int bcp_device_normals(bcp_handle *h, int64_t first_env, int64_t n_envs, uint64_t first_step, int32_t n_steps, double *out,
                       void *stream);

/* ---- measurement -------------------------------------------------------------------------------------- */
/* Which kernels a bcp_step() of this handle launches, as configured now: 0 = step_kernel alone (no distance field, or a
 * forced mode), 1 = step_fast_pair_kernel alone (every undecided pose settled in place), 2 = step_fast_pair_kernel +
 * step_pending_kernel, 3 = step_local_kernel (one launch).  Negative: BCP_E_*. */
int bcp_step_form(bcp_handle *h);
/* Runs `steps` bcp_step() launches back to back on `stream` bracketed by HIP events recorded on that stream and
 * returns the average kernel-launch duration in milliseconds (synchronises).  Used by bench.py for
 * roofline.achieved. */
int bcp_time_steps(bcp_handle *h, const bcp_step_io *io, uint32_t flags, int32_t steps, void *stream, float *avg_ms);
/* Per-kernel split of a step.  Single-launch forms (bcp_step_form 0, 1, 3 -- the default is 3): kernel_ms[0] = the average
 * step, kernel_ms[1] = 0; the envs advance by `steps` steps.  Two-launch form (2): kernel_ms[0] = average duration of
 * kernel 1 (a second loop of `steps` launches of kernel 1 alone), kernel_ms[1] = average full step minus kernel_ms[0], i.e.
 * step_pending_kernel plus the launch boundary.  Coarse: the kernel-1-only loop re-parks the same poses without settling
 * them; profiles/ (rocprofv3 kernel trace) is the reference for per-kernel durations. */
int bcp_time_step_kernels(bcp_handle *h, const bcp_step_io *io, uint32_t flags, int32_t steps, void *stream,
                          float *kernel_ms);

#ifdef __cplusplus
}
#endif
#endif /* BCPLAN_H */
