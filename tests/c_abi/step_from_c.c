/* A plain C11 client of libbcplan.so: no Python, no torch, no C++.  It builds a small scene (64 x 64 costmap with a
 * wall, a straight 40-point path), 128 tricycle envs without noise, steps them 60 times with auto-reset and prints
 * the final state as hex floats.  tests/test_gpu_c_abi.py compiles it with gcc, runs it and compares the output with
 * the same scene stepped through the Python host layer: the C ABI alone (include/bcplan.h) is enough to drive the path.
 *
 * gcc -std=c11 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include tests/c_abi/step_from_c.c \
 *     bc_gym_planning_env_amd/libbcplan.so -L/opt/rocm/lib -lamdhip64 -lm -o step_from_c */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bcplan.h"

#define N 128
#define ROWS 64
#define COLS 64
#define M 40
#define STEPS 60

#define CHECK(call)                                                                  \
    do {                                                                             \
        int rc_ = (call);                                                            \
        if (rc_ != 0) {                                                              \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, bcp_last_error());   \
            return 1;                                                                \
        }                                                                            \
    } while (0)
#define HIP(call)                                                                    \
    do {                                                                             \
        hipError_t e_ = (call);                                                      \
        if (e_ != hipSuccess) {                                                      \
            fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));               \
            return 1;                                                                \
        }                                                                            \
    } while (0)

/* robot_models/robot_dimensions_examples.py:171-188 (mm) */
static const double kTricycleMm[16][2] = {
    {1348.35, 0.}, {1338.56, 139.75}, {1306.71, 280.12}, {1224.36, 338.62}, {1093.81, 374.64}, {-214.37, 374.64},
    {-313.62, 308.56}, {-366.36, 117.44}, {-374.01, -135.75}, {-227.96, -459.13}, {-156.72, -458.78},
    {759.8, -442.96}, {849.69, -426.4}, {1171.05, -353.74}, {1303.15, -286.54}, {1341.34, -118.37}};

static void *dev_copy(const void *host, size_t bytes)
{
    void *d = NULL;
    if (hipMalloc(&d, bytes) != hipSuccess) return NULL;
    if (host) {
        if (hipMemcpy(d, host, bytes, hipMemcpyHostToDevice) != hipSuccess) return NULL;
    } else if (hipMemset(d, 0, bytes) != hipSuccess) {
        return NULL;
    }
    return d;
}

int main(void)
{
    const double pi = 3.14159265358979323846;
    bcp_params p;
    memset(&p, 0, sizeof(p));
    p.abi_version = BCP_ABI_VERSION;
    p.model = BCP_MODEL_TRICYCLE;
    p.n_verts = 16;
    for (int k = 0; k < 16; ++k) {
        p.verts[k][0] = kTricycleMm[k][0] / 1000.;
        p.verts[k][1] = kTricycleMm[k][1] / 1000.;
    }
    p.dynamic_model = 1;
    p.model_front_column_pid = 1;
    p.noise_on = 0;
    p.iteration_timeout = 45;
    p.dt = 0.05;
    p.front_wheel_from_axis = 0.964;
    p.max_front_wheel_angle = 0.5 * 170 * pi / 180.;
    p.max_front_wheel_speed = 60. * pi / 180.;
    p.max_linear_acceleration = 1. / 2.5;
    p.max_angular_acceleration = 1. / 2.;
    p.front_column_p_gain = 0.16;
    p.spatial_precision = 0.2;
    p.angular_precision = pi / 8.;
    p.spatial_progress_multiplier = 0.0;
    p.reward_provider = BCP_REWARD_CONTINUOUS;

    bcp_handle *h = NULL;
    CHECK(bcp_create(&p, N, 0, 0, &h));

    /* costmap: free space with a lethal wall at column 40, rows 8 .. 55; origin (-0.5, -1.6), 0.05 m/px */
    static uint8_t map[ROWS * COLS];
    for (int r = 8; r < 56; ++r) map[r * COLS + 40] = 254;
    const double origin[2] = {-0.5, -1.6};
    uint8_t *d_map = dev_copy(map, sizeof(map));
    CHECK(bcp_set_costmaps(h, d_map, ROWS, COLS, 1, NULL, NULL, origin, 0, 0.05, NULL));

    /* path: 40 way points along +x from the origin, 5 cm apart (already "refined") */
    static double path[M * 3];
    for (int j = 0; j < M; ++j) {
        path[3 * j] = 0.05 * j;
        path[3 * j + 1] = 0.0;
        path[3 * j + 2] = 0.0;
    }
    double *d_path = dev_copy(path, sizeof(path));
    CHECK(bcp_set_paths(h, d_path, NULL, M, 1, NULL));

    /* state (SoA) and the initial state PlanEnv.__init__ would make: pose = path[0], target way point 1 */
    static double zeros[N], min_dist[N];
    static int32_t target[N];
    for (int i = 0; i < N; ++i) {
        min_dist[i] = hypot(path[3] - path[0], path[4] - path[1]);
        target[i] = 1;
    }
    bcp_state st[2];
    for (int k = 0; k < 2; ++k) {
        memset(&st[k], 0, sizeof(st[k]));
        st[k].x = dev_copy(zeros, sizeof(zeros));
        st[k].y = dev_copy(zeros, sizeof(zeros));
        st[k].angle = dev_copy(zeros, sizeof(zeros));
        st[k].v = dev_copy(zeros, sizeof(zeros));
        st[k].w = dev_copy(zeros, sizeof(zeros));
        st[k].steering_motor_command = dev_copy(zeros, sizeof(zeros));
        st[k].wheel_angle = dev_copy(zeros, sizeof(zeros));
        st[k].min_spat_dist_so_far = dev_copy(min_dist, sizeof(min_dist));
        st[k].target_idx = dev_copy(target, sizeof(target));
        st[k].current_iter = dev_copy(NULL, N * sizeof(int32_t));
        st[k].robot_collided = dev_copy(NULL, N);
    }
    CHECK(bcp_bind_state(h, &st[0]));
    CHECK(bcp_bind_initial_state(h, &st[1]));

    /* actions: every env drives at 0.5 m/s with its own steering angle (float64 [N, 2]) */
    static double actions[N * 2];
    for (int i = 0; i < N; ++i) {
        actions[2 * i] = 0.5;
        actions[2 * i + 1] = 0.02 * (i % 21 - 10);
    }
    bcp_step_io io;
    memset(&io, 0, sizeof(io));
    io.actions = dev_copy(actions, sizeof(actions));
    io.reward = dev_copy(NULL, N * sizeof(double));
    io.done = dev_copy(NULL, N);
    io.collided_now = dev_copy(NULL, N);
    double total_reward[N];
    int n_done = 0;
    memset(total_reward, 0, sizeof(total_reward));
    for (int t = 0; t < STEPS; ++t) {
        CHECK(bcp_step(h, &io, BCP_STEP_AUTO_RESET, NULL));
        double rew[N];
        uint8_t done[N];
        HIP(hipMemcpy(rew, io.reward, sizeof(rew), hipMemcpyDeviceToHost));
        HIP(hipMemcpy(done, io.done, sizeof(done), hipMemcpyDeviceToHost));
        for (int i = 0; i < N; ++i) {
            total_reward[i] += rew[i];
            n_done += done[i];
        }
    }
    double x[N], y[N], th[N];
    int32_t tgt[N], it[N];
    HIP(hipMemcpy(x, st[0].x, sizeof(x), hipMemcpyDeviceToHost));
    HIP(hipMemcpy(y, st[0].y, sizeof(y), hipMemcpyDeviceToHost));
    HIP(hipMemcpy(th, st[0].angle, sizeof(th), hipMemcpyDeviceToHost));
    HIP(hipMemcpy(tgt, st[0].target_idx, sizeof(tgt), hipMemcpyDeviceToHost));
    HIP(hipMemcpy(it, st[0].current_iter, sizeof(it), hipMemcpyDeviceToHost));
    printf("episodes_ended %d\n", n_done);
    for (int i = 0; i < N; ++i) printf("%d %a %a %a %d %d %a\n", i, x[i], y[i], th[i], tgt[i], it[i], total_reward[i]);
    CHECK(bcp_destroy(h));
    return 0;
}
