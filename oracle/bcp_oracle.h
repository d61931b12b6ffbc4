/*
 * bcp_oracle.h -- CPU ORACLE for the PlanEnv.step() hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C, float64 restatement of the reference algorithm
 * (braincorp/bc-gym-planning-env).  It exists so that tests/, __graft_entry__.smoke()
 * and bench.py's `cpu_baseline` leg can check / time the HIP path against it.
 * Nothing under bc_gym_planning_env_amd/ (the product) may include, link or call it.
 *
 * Parity status: every function except bco_fill_poly() is pinned bit-for-bit against
 * outputs of the reference's own Python code imported in the build container
 * (oracle/gen_golden.py -> tests/golden/ fixtures) and against the reference's
 * known-answer tests.  bco_fill_poly() restates the published OpenCV fillPoly
 * algorithm (opencv-python, UNPINNED in reference setup.py:20-23 / Pipfile:13, source
 * absent from /root/reference and from this image); it is pinned only by the reference
 * KATs test_path_tools.py:465-468 (493 cells) and test_costmap_utils.py:251-314
 * (20-pose collision table).  Beyond those the exact pixel set of a filled polygon is
 * "parity unpinned".  The same holds for the egocentric-observation functions
 * bco_rotation_matrix_2d() / bco_warp_affine_nearest() (cv2.getRotationMatrix2D / cv2.warpAffine,
 * INTER_NEAREST): restated from the published OpenCV algorithm (fixed-point AB_BITS = 10 coordinates)
 * and pinned only by the reference KATs test_costmap_utils.py:38-189, 192-207 (mark positions after
 * rotation / shift / cut, the 5-pixel image of a rotated 2x2 block, rotate_costmap's (29, 47)).
 *
 * Build:  make -C oracle         (gcc -O2 -ffp-contract=off, no fast-math)
 */
#ifndef BCP_ORACLE_H
#define BCP_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BCO_MAX_VERTS 32
#define BCO_LETHAL 254 /* CostMap2D.LETHAL_OBSTACLE, utilities/costmap_2d.py:20-22 */

enum { BCO_MODEL_TRICYCLE = 0, BCO_MODEL_DIFFDRIVE = 1 };

/* error bits reported per env (mirror of the reference's Python exceptions) */
enum {
    BCO_ERR_NONE = 0,
    BCO_ERR_ANGLE_JUMP = 1 /* path_velocity raises: |dtheta| >= pi, utilities/path_tools.py:319-322 */
};

/* POD flattening of EnvParams (envs/base/params.py:14-42), RewardParams (reward.py:162-171)
 * and the robot constants (robot_models/robot_dimensions_examples.py:108-188). */
typedef struct bco_params {
    int32_t model;               /* BCO_MODEL_* */
    int32_t n_verts;             /* footprint vertices (16 tricycle / 24 diffdrive) */
    double verts[BCO_MAX_VERTS][2]; /* metres, robot frame, already * footprint_scale */
    double dt;                   /* EnvParams.dt */
    double front_wheel_from_axis;
    double max_front_wheel_angle;
    double max_front_wheel_speed;
    double max_linear_acceleration;
    double max_angular_acceleration;
    double front_column_p_gain;
    int32_t dynamic_model;       /* TricycleRobot._dynamic_model (tricycle_model.py:298) */
    int32_t model_front_column_pid; /* TricycleRobot._model_front_column_pid (:299) */
    int32_t noise_on;            /* noise_parameters is not None */
    int32_t iteration_timeout;
    double alpha[6];             /* alpha1..alpha6, differential_drive.py:65-70 */
    double spatial_precision;
    double angular_precision;
    double spatial_progress_multiplier;
    /* SURVEY 8(f) row 4 */
    int32_t reward_provider;     /* BCO_REWARD_*: EnvParams.reward_provider_name (params.py:33) */
    int32_t control_delay;       /* EnvParams.control_delay / pose_delay / state_delay (params.py:28-30) */
    int32_t pose_delay;
    int32_t state_delay;
} bco_params;

enum { BCO_REWARD_CONTINUOUS = 0, BCO_REWARD_PURE_PURSUIT = 1 };

/* Per-env FIFO state of _get_element_from_list_with_delay (env.py:27-49).  The k-th element pushed since the last
 * reset (k = 1, 2, ...) lives in slot (k - 1) % delay of its queue; k is current_iter + 1 at the time of the push. */
typedef struct bco_delay_state {
    double *control_q; /* [control_delay][2] */
    double *pose_q;    /* [pose_delay][3] */
    double *state_q;   /* [state_delay][7] */
    double *obs_pose;  /* [3]  State.pose: what the reward provider and the observation see */
    double *obs_state; /* [7]  State.robot_state of the observation */
} bco_delay_state;

/* ---- scalar utilities -------------------------------------------------------------- */
double bco_normalize_angle(double z);                 /* coordinate_transformations.py:28-36 */
void bco_world_to_pixel(const double *xy, int64_t n, const double origin[2], double resolution,
                        int64_t *out);                /* coordinate_transformations.py:185-205 */
/* 2-row path_velocity, path_tools.py:298-323.  returns BCO_ERR_* */
int bco_path_velocity(const double pose0[3], const double pose1[3], double dt, double *v, double *w);
/* differential_drive.py:21-40 */
void bco_kinematic_step(const double pose[3], double v, double w, double dt, double out[3]);
/* differential_drive.py:55-74.  z[3] are standard normals in SLOT order (slot k is used only
 * when variance_k > 0).  *drawn gets the bitmask of consumed slots. */
void bco_kinematic_step_noise(const double pose[3], double v, double w, double dt, const double alpha[6],
                              const double z[3], double out[3], int *drawn);

/* tricycle_model.py:127-154 */
double bco_front_wheel_column_step(double cur, double desired, double max_angle, double max_speed,
                                   double p_gain, double dt);
/* tricycle_model.py:157-188 */
void bco_velocity_dynamic_model_step(double cur_v, double cur_w, double wheel_angle, double desired_wheel_v,
                                     double front_wheel_from_axis, double max_lin_acc, double max_ang_acc,
                                     double dt, double *new_v, double *new_w);

/* robot state layout: {x, y, angle, v, w, steering_motor_command, wheel_angle}
 * (TricycleRobotState tricycle_model.py:234-244; DiffdriveRobotState uses the first five,
 * differential_drive.py:83-87).  TricycleRobot.step :478-538 / DiffDriveRobot.step :236-265 */
int bco_robot_step(const bco_params *p, double st[7], const double cmd[2], const double z[3], int *drawn);

/* ---- footprint / collision --------------------------------------------------------- */
/* pre-fill part of get_pixel_footprint, path_tools.py:140-150: integer polygon (already shifted by
 * +half) and half sizes {half_x, half_y}.  dot_fma!=0 reproduces an FMA-contracted BLAS ddot tail. */
void bco_footprint_vertices(double angle, const double *verts, int k, double resolution,
                            int32_t *out_xy, int32_t half[2]);
/* OpenCV-compatible cv2.fillPoly(img, [pts], value) for ONE integer contour, shift 0, LINE_8 */
void bco_fill_poly(uint8_t *img, int rows, int cols, const int32_t *pts_xy, int k, uint8_t value);
/* cv2.line(img, p0, p1, value, thickness=1) stand-in (clipLine + 8-connected Bresenham). Used only to
 * draw walls when building maps for fixtures (reset path, not the step path). */
void bco_line(uint8_t *img, int rows, int cols, int64_t x0, int64_t y0, int64_t x1, int64_t y1, uint8_t value);
/* get_pixel_footprint, path_tools.py:122-162.  out must hold (2*hy+1)*(2*hx+1) bytes; returns dims */
void bco_pixel_footprint(double angle, const double *verts, int k, double resolution,
                         uint8_t *out, int out_cap, int *h, int *w);
/* pose_collides, envs/base/env.py:464-489 */
int bco_pose_collides(double x, double y, double angle, const double *verts, int k,
                      const uint8_t *map, int rows, int cols, const double origin[2], double resolution);

/* ---- reward ------------------------------------------------------------------------ */
/* find_last_reached, path_tools.py:408-448.  returns -1 for None */
int bco_find_last_reached(const double pose[3], const double *path, int m, double sp, double ap);
/* ContinuousRewardProvider.reward, reward.py:214-259 (mutates min_dist/target_idx) */
double bco_reward(const bco_params *p, const double pose[3], const double *path, int m,
                  double *min_dist, int32_t *target_idx);
/* ContinuousRewardProvider.generate_initial_state reward.py:261-288. returns 0, or -1 for ValueError */
int bco_initial_reward_state(const double *path, int m, double sp, double ap, double *min_dist, int32_t *target_idx);

/* ContinuousRewardPurePursuitProvider.reward (reward.py:330-353) incl. update_goal (:125-139); mutates the state */
double bco_reward_pure_pursuit(const double pose[3], const double *path, int m, int collided, double *min_dist,
                               int32_t *target_idx);
/* ContinuousRewardPurePursuitProvider.generate_initial_state (reward.py:355-371) */
void bco_initial_pure_pursuit_state(const double *path, int m, double *min_dist, int32_t *target_idx);

/* ---- full step --------------------------------------------------------------------- */
/* One PlanEnv.step (env.py:334-361, delays 0) for ONE env.  Returns BCO_ERR_*. */
int bco_env_step(const bco_params *p, double st[7], double *min_dist, int32_t *target_idx, int32_t *cur_iter,
                 double *cur_time, uint8_t *collided_sticky, const double cmd[2], const double z[3],
                 const uint8_t *map, int rows, int cols, const double origin[2], double resolution,
                 const double *path, int m, double *reward, uint8_t *done, uint8_t *collided_now, int *drawn);

/* The same with EnvParams' delays and reward provider honoured (p->*_delay, p->reward_provider); `d` may be NULL when
 * all delays are 0.  st is the robot's TRUE state; d->obs_pose / d->obs_state receive what State exposes. */
int bco_env_step_ex(const bco_params *p, double st[7], bco_delay_state *d, double *min_dist, int32_t *target_idx,
                    int32_t *cur_iter, double *cur_time, uint8_t *collided_sticky, const double cmd[2], const double z[3],
                    const uint8_t *map, int rows, int cols, const double origin[2], double resolution,
                    const double *path, int m, double *reward, uint8_t *done, uint8_t *collided_now, int *drawn);

/* ---- egocentric observation -------------------------------------------------------- */
/* cv2.getRotationMatrix2D(center, angle_deg, scale): center is a Point2f (float32), everything else float64. */
void bco_rotation_matrix_2d(double cx, double cy, double angle_deg, double scale, double M[6]);
/* cv2.warpAffine(src, M, (dcols, drows), flags=INTER_NEAREST, borderMode=BORDER_CONSTANT, borderValue): M maps
 * src -> dst and is inverted in float64 first; source coordinates are 22.10 fixed point. */
void bco_warp_affine_nearest(const uint8_t *src, int rows, int cols, const double M[6], uint8_t *dst, int drows,
                             int dcols, uint8_t border);
/* extract_egocentric_costmap (utilities/costmap_utils.py:25-75).  has_window: resulting_origin / resulting_size given
 * (size in metres); otherwise the output has the source's shape.  out must hold out_shape[0]*out_shape[1] bytes
 * (call with out == NULL to get the shape only).  M_used (optional) receives the 2x3 transform given to warpAffine. */
void bco_extract_egocentric(const uint8_t *map, int rows, int cols, const double origin[2], double resolution,
                            const double pose[3], int has_window, const double window_origin[2],
                            const double window_size[2], uint8_t border, uint8_t *out, int32_t out_shape[2],
                            double *M_used);
/* rotate_costmap (costmap_utils.py:78-104), center None */
void bco_rotate_costmap(const uint8_t *map, int rows, int cols, double angle, uint8_t border, uint8_t *out);
/* EgocentricCostmap.observation's goal_n_state vector (envs/egocentric.py:140-160): the next way point in the robot
 * frame (from_global_to_egocentric, coordinate_transformations.py:341-362), its position divided by the window's world
 * size and clipped to [-1, 1], then robot_state.to_numpy_array(); float32.  n_state = 6 (tricycle) / 5 (diff-drive).
 * remaining == 0 (path exhausted) gives zeros. */
void bco_goal_n_state(const double pose[3], const double *next_waypoint, int remaining, const double world_size[2],
                      const double *robot_state, int n_state, float *out);

/* ColoredEgoCostmapRandomAisleTurnEnv's `goal` vector (envs/synth_turn_env.py:412-420): the LAST way point in the robot
 * frame divided by the window's world size, normalised to unit length (np.linalg.norm), then the robot's egocentric
 * state (v, w, wheel_angle).  float64 [5]. */
void bco_goal_direction_state(const double pose[3], const double last_waypoint[3], const double world_size[2],
                              const double ego_state[3], double out[5]);

/* Batched SoA step over n envs with `threads` host threads.
 * state: 7 arrays of n doubles, state[f][i].  maps: shared (map_stride==0) or per-env at i*map_stride bytes.
 * paths: shared (path_stride==0, lens[0]) or per-env at i*path_stride doubles with lens[i].
 * actions: n*2 (float64).  z: n*3 or NULL.  auto_reset: when done, state <- init_* after outputs are written.
 * geom (optional, int32[n]): geometry-pool mode -- env i uses entry geom[i] of the per-env map / origin / path /
 * init arrays instead of entry i; on auto-reset geom[i] <- next_geom[geom[i]] (when given) first, which is
 * RandomMiniEnv.reset() drawing its next world (envs/mini_env.py:469-481) from a pre-sampled chain. */
typedef struct bco_batch {
    int64_t n;
    double *st[7];
    double *min_dist;
    int32_t *target_idx;
    int32_t *cur_iter;
    double *cur_time;
    uint8_t *collided;
    const uint8_t *maps; int64_t map_stride; int32_t rows, cols; /* rows/cols arrays optional */
    const int32_t *rows_per_env; const int32_t *cols_per_env;    /* NULL => rows/cols for all */
    const double *origins; int64_t origin_stride;                /* 0 => shared */
    double resolution;
    const double *paths; int64_t path_stride; const int32_t *lens;
    const double *actions;
    const double *z;
    double *reward; uint8_t *done; uint8_t *collided_now; int32_t *err;
    int32_t auto_reset;
    const double *init_st[7]; const double *init_min_dist; const int32_t *init_target_idx;
    int32_t *geom; const int32_t *next_geom;
    /* delays > 0 (all optional): AoS per env, [n][delay][width] queues and [n][3] / [n][7] observed pose / state */
    double *control_q, *pose_q, *state_q, *obs_pose, *obs_state;
} bco_batch;
int bco_step_batch(const bco_params *p, const bco_batch *b, int threads);
/* `steps` steps back to back, every thread taking its block of envs through all of them (no per-step thread start):
 * step k reads batch k % pool_len of actions_pool [pool_len][n][2] / z_pool [pool_len][n][3] (NULL: no noise input). */
int bco_run_steps(const bco_params *p, const bco_batch *b, int threads, int steps, const double *actions_pool,
                  const double *z_pool, int pool_len);

#ifdef __cplusplus
}
#endif
#endif
