// bcplan.hip -- libbcplan.so: batched PlanEnv.step() for MI355X (gfx950).  C ABI in include/bcplan.h.
//
// One fused kernel advances every env by one tick:
//   load SoA state -> robot model (fp64) -> footprint rasterise + lethal-bit test -> rollback on collision
//   -> ContinuousRewardProvider -> done -> (auto-reset) -> store SoA state.
// Compiled with -ffp-contract=off (numpy rounds every product and sum separately).  No CPU path exists here.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>

#include "bcp_device.h"
#include "bcp_raster.h"

using namespace bcp;

// ------------------------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                                \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) return fail(BCP_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

extern "C" const char* bcp_last_error(void) { return g_err; }
extern "C" int bcp_abi_version(void) { return BCP_ABI_VERSION; }

// ------------------------------------------------------------------------------------------------ handle
struct DevState {
    double *x, *y, *angle, *v, *w, *steer, *wheel, *min_dist;
    int32_t *target_idx, *cur_iter;
    uint8_t* collided;
};

struct MapDesc {
    const uint32_t* bits;  // lethal bitmap, [rows][wpr] shared or [N][rows][wpr]
    int32_t rows, cols, wpr;
    int32_t shared;
    int32_t in_lds;        // shared bitmap small enough to be staged in LDS
    int64_t env_stride;    // words per env (0 when shared)
    const double* origins; // device [N,2] when per-env, else NULL
    double ox, oy, inv_res;
};

struct PathDesc {
    const double* pts;  // [len][5] = x, y, theta, cos(theta), sin(theta); shared or [N][max_len][5]
    const int32_t* lens;
    int32_t max_len, shared;
};

struct bcp_handle {
    bcp_params params;
    DevParams dev;
    int64_t n;
    int device;
    int64_t env_id_base;
    uint64_t seed;
    uint64_t step_counter;
    bool have_map, have_path, have_state, have_init;
    double resolution;
    uint32_t* bitmap;      // owned
    size_t bitmap_bytes;
    double* path5;         // owned
    size_t path5_bytes;
    MapDesc map;
    PathDesc path;
    DevState st, init;
};

static DevState to_dev_state(const bcp_state* s)
{
    DevState d;
    d.x = s->x; d.y = s->y; d.angle = s->angle; d.v = s->v; d.w = s->w;
    d.steer = s->steering_motor_command; d.wheel = s->wheel_angle; d.min_dist = s->min_spat_dist_so_far;
    d.target_idx = s->target_idx; d.cur_iter = s->current_iter; d.collided = s->robot_collided;
    return d;
}

static int check_state(const bcp_state* s, int tricycle)
{
    if (!s) return 0;
    if (!s->x || !s->y || !s->angle || !s->v || !s->w || !s->min_spat_dist_so_far || !s->target_idx ||
        !s->current_iter || !s->robot_collided)
        return 0;
    if (tricycle && (!s->steering_motor_command || !s->wheel_angle)) return 0;
    return 1;
}

// ------------------------------------------------------------------------------------------------ kernels
constexpr int kBlock = 64;  // one wavefront per workgroup

// uint8 costmap -> 1-bit lethal mask.  One thread per 32-bit output word.
__global__ void pack_bitmap_kernel(const uint8_t* __restrict__ data, uint32_t* __restrict__ bits, int64_t n_maps,
                                   int rows, int cols, int wpr, const int32_t* __restrict__ valid_rows,
                                   const int32_t* __restrict__ valid_cols)
{
    const int64_t total = n_maps * rows * wpr;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(idx % wpr);
        const int64_t t = idx / wpr;
        const int r = (int)(t % rows);
        const int64_t m = t / rows;
        const int vr = valid_rows ? valid_rows[m] : rows;
        const int vc = valid_cols ? valid_cols[m] : cols;
        uint32_t word = 0;
        if (r < vr) {
            const uint8_t* src = data + (m * rows + r) * (int64_t)cols + (int64_t)w * 32;
            const int lim = min(32, vc - w * 32);
            for (int b = 0; b < lim; ++b) word |= (uint32_t)(src[b] == BCP_LETHAL) << b;
        }
        bits[idx] = word;
    }
}

// path [.,3] -> [.,5] with cos/sin of the heading (utilities/path_tools.py:405)
__global__ void path_trig_kernel(const double* __restrict__ xyt, double* __restrict__ out, int64_t total)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const double th = xyt[3 * i + 2];
        out[5 * i + 0] = xyt[3 * i + 0];
        out[5 * i + 1] = xyt[3 * i + 1];
        out[5 * i + 2] = th;
        out[5 * i + 3] = cos(th);
        out[5 * i + 4] = sin(th);
    }
}

// find_last_reached restricted to j >= target (utilities/path_tools.py:408-448): the reward only asks whether the
// LAST reached index is >= target_idx (envs/base/reward.py:234), so indices below target never matter.
__device__ __forceinline__ int last_reached_from(const DevParams& P, const double* __restrict__ path, int m, int target,
                                                 double x, double y, double th)
{
    for (int j = m - 1; j >= target; --j) {
        const double* s = path + 5 * j;
        const double dx = s[0] - x, dy = s[1] - y;
        if (fabs(dx) > P.sp_prune || fabs(dy) > P.sp_prune) continue;  // then hypot(dx,dy) >= sp
        const double dist = hypot(dx, dy);
        if (!(dist < P.sp)) continue;
        const double ang = fabs(normalize_angle(th - s[2]));
        if (!(ang < P.ap)) continue;
        const double par = s[3] * (x - s[0]) + s[4] * (y - s[1]);
        if (par >= P.par_thr) return j;
    }
    return -1;
}

// ContinuousRewardProvider.reward (envs/base/reward.py:214-259)
__device__ __forceinline__ double reward_step(const DevParams& P, const double* __restrict__ path, int m, double x,
                                              double y, double th, double& min_dist, int& target)
{
    if (target > m - 1) return 0.0;
    const int last = last_reached_from(P, path, m, target, x, y, th);
    if (last >= 0) {
        target = last + 1;
        if (!(target > m - 1)) {
            const double* g = path + 5 * target;
            min_dist = hypot(g[0] - x, g[1] - y);
        } else {
            min_dist = 0.0;
        }
        return 1.0;
    }
    const double* g = path + 5 * target;
    const double d = hypot(g[0] - x, g[1] - y);
    if (d < min_dist) {
        const double r = min_dist - d;
        min_dist = d;
        return r * P.progress_mult;
    }
    return 0.0;
}

struct StepArgs {
    DevParams P;
    MapDesc map;
    PathDesc path;
    DevState st, init;
    int64_t n;
    const void* actions;
    const double* noise_z;
    double* noise_z_out;
    double* reward;
    uint8_t* done;
    uint8_t* collided_now;
    int32_t* err;
    uint32_t flags;
    uint64_t seed, step_counter;
    int64_t env_id_base;
};

// dynamic LDS: [bitmap words (when staged)] [vertex scratch: n_verts * 2 * kBlock words]
extern __shared__ uint32_t lds_dyn[];

__global__ void __launch_bounds__(kBlock) step_kernel(const StepArgs a)
{
    const DevParams& P = a.P;
    const int tid = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * kBlock + tid;
    const bool active = i < a.n;

    // stage the shared lethal bitmap in LDS
    uint32_t* lds_bits = lds_dyn;
    const int map_words = a.map.in_lds ? a.map.rows * a.map.wpr : 0;
    for (int k = tid; k < map_words; k += kBlock) lds_bits[k] = a.map.bits[k];
    VertLds E;
    E.base = lds_dyn + map_words + tid;
    E.stride = kBlock;
    __syncthreads();
    if (!active) return;

    // ---- load state
    Robot r;
    r.p.x = a.st.x[i];
    r.p.y = a.st.y[i];
    r.p.th = a.st.angle[i];
    r.v = a.st.v[i];
    r.w = a.st.w[i];
    const bool tri = P.model == BCP_MODEL_TRICYCLE;
    r.steer = tri ? a.st.steer[i] : 0.0;
    r.wheel = tri ? a.st.wheel[i] : 0.0;
    double min_dist = a.st.min_dist[i];
    int target = a.st.target_idx[i];
    int iter = a.st.cur_iter[i];
    bool collided = a.st.collided[i] != 0;

    double cmd0, cmd1;
    if (a.flags & BCP_STEP_ACTIONS_F32) {
        const float2 c = reinterpret_cast<const float2*>(a.actions)[i];
        cmd0 = (double)c.x;
        cmd1 = (double)c.y;
    } else {
        const double2 c = reinterpret_cast<const double2*>(a.actions)[i];
        cmd0 = c.x;
        cmd1 = c.y;
    }
    double z[3] = {0.0, 0.0, 0.0};
    if (P.noise_on) {
        if (a.noise_z) {
            z[0] = a.noise_z[3 * i + 0];
            z[1] = a.noise_z[3 * i + 1];
            z[2] = a.noise_z[3 * i + 2];
        } else {
            device_normals(a.seed, (uint64_t)(a.env_id_base + i), a.step_counter, z);
        }
    }

    // ---- _env_step (envs/base/env.py:442-461)
    const Pose old = r.p;
    int drawn = 0;
    const int err = robot_step(P, r, cmd0, cmd1, z, drawn);

    MapXform X;
    X.inv_res = a.map.inv_res;
    if (a.map.origins) {
        X.ox = a.map.origins[2 * i + 0];
        X.oy = a.map.origins[2 * i + 1];
    } else {
        X.ox = a.map.ox;
        X.oy = a.map.oy;
    }
    bool hit = false;
    if (a.flags & (1u << 16)) {
    } else if (a.map.in_lds) {
        hit = pose_collides(P, r.p.x, r.p.y, r.p.th, X, (const uint32_t*)lds_bits, a.map.rows, a.map.cols, a.map.wpr, E);
    } else {
        const uint32_t* bits = a.map.bits + (a.map.shared ? 0 : i * a.map.env_stride);
        hit = pose_collides(P, r.p.x, r.p.y, r.p.th, X, bits, a.map.rows, a.map.cols, a.map.wpr, E);
    }
    if (hit) {  // robot.set_pose(*old_position): pose restored, v = w = 0 (tricycle_model.py:471-476)
        r.p = old;
        r.v = 0.0;
        r.w = 0.0;
    }
    // ---- _resolve_state_transition bookkeeping (env.py:382-396)
    iter += 1;
    collided = collided || hit;

    // ---- reward / done (env.py:352, :407-419)
    const double* path = a.path.pts + (a.path.shared ? 0 : i * (int64_t)a.path.max_len * 5);
    const int m = a.path.shared ? a.path.max_len : a.path.lens[i];
    const double rew = (a.flags & (1u << 17)) ? 0.0 : reward_step(P, path, m, r.p.x, r.p.y, r.p.th, min_dist, target);
    const bool done = (target > m - 1) || (iter >= P.iteration_timeout) || collided;

    a.reward[i] = rew;
    a.done[i] = (uint8_t)done;
    if (a.collided_now) a.collided_now[i] = (uint8_t)hit;
    if (a.err) a.err[i] = err;
    if (a.noise_z_out) {
        const double nan = __builtin_nan("");
        a.noise_z_out[3 * i + 0] = (drawn & 1) ? z[0] : nan;
        a.noise_z_out[3 * i + 1] = (drawn & 2) ? z[1] : nan;
        a.noise_z_out[3 * i + 2] = (drawn & 4) ? z[2] : nan;
    }

    if (done && (a.flags & BCP_STEP_AUTO_RESET)) {  // PlanEnv.reset(): set_state(initial_state) (env.py:293-303)
        r.p.x = a.init.x[i];
        r.p.y = a.init.y[i];
        r.p.th = a.init.angle[i];
        r.v = a.init.v[i];
        r.w = a.init.w[i];
        if (tri) {
            r.steer = a.init.steer[i];
            r.wheel = a.init.wheel[i];
        }
        min_dist = a.init.min_dist[i];
        target = a.init.target_idx[i];
        iter = a.init.cur_iter[i];
        collided = a.init.collided[i] != 0;
    }

    // ---- store state
    a.st.x[i] = r.p.x;
    a.st.y[i] = r.p.y;
    a.st.angle[i] = r.p.th;
    a.st.v[i] = r.v;
    a.st.w[i] = r.w;
    if (tri) {
        a.st.steer[i] = r.steer;
        a.st.wheel[i] = r.wheel;
    }
    a.st.min_dist[i] = min_dist;
    a.st.target_idx[i] = target;
    a.st.cur_iter[i] = iter;
    a.st.collided[i] = (uint8_t)collided;
}

__global__ void reset_kernel(DevState st, DevState init, const uint8_t* __restrict__ mask, int64_t n, int tri)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (mask && !mask[i]) return;
    st.x[i] = init.x[i];
    st.y[i] = init.y[i];
    st.angle[i] = init.angle[i];
    st.v[i] = init.v[i];
    st.w[i] = init.w[i];
    if (tri) {
        st.steer[i] = init.steer[i];
        st.wheel[i] = init.wheel[i];
    }
    st.min_dist[i] = init.min_dist[i];
    st.target_idx[i] = init.target_idx[i];
    st.cur_iter[i] = init.cur_iter[i];
    st.collided[i] = init.collided[i];
}

__global__ void __launch_bounds__(kBlock) robot_step_kernel(DevParams P, double* __restrict__ st7, int64_t n,
                                                            const double* __restrict__ actions,
                                                            const double* __restrict__ noise_z, int32_t* __restrict__ err)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    Robot r;
    r.p.x = st7[0 * n + i];
    r.p.y = st7[1 * n + i];
    r.p.th = st7[2 * n + i];
    r.v = st7[3 * n + i];
    r.w = st7[4 * n + i];
    r.steer = st7[5 * n + i];
    r.wheel = st7[6 * n + i];
    double z[3] = {0.0, 0.0, 0.0};
    if (noise_z) {
        z[0] = noise_z[3 * i];
        z[1] = noise_z[3 * i + 1];
        z[2] = noise_z[3 * i + 2];
    }
    int drawn = 0;
    const int e = robot_step(P, r, actions[2 * i], actions[2 * i + 1], z, drawn);
    st7[0 * n + i] = r.p.x;
    st7[1 * n + i] = r.p.y;
    st7[2 * n + i] = r.p.th;
    st7[3 * n + i] = r.v;
    st7[4 * n + i] = r.w;
    st7[5 * n + i] = r.steer;
    st7[6 * n + i] = r.wheel;
    if (err) err[i] = e;
}

__global__ void __launch_bounds__(kBlock) pose_collides_kernel(DevParams P, MapDesc map, const double* __restrict__ poses,
                                                               int64_t n, int64_t n_envs, uint8_t* __restrict__ out)
{
    const int tid = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * kBlock + tid;
    VertLds E;
    E.base = lds_dyn + tid;
    E.stride = kBlock;
    if (i >= n) return;
    const int64_t env = i % n_envs;
    MapXform X;
    X.inv_res = map.inv_res;
    X.ox = map.origins ? map.origins[2 * env] : map.ox;
    X.oy = map.origins ? map.origins[2 * env + 1] : map.oy;
    const uint32_t* bits = map.bits + (map.shared ? 0 : env * map.env_stride);
    out[i] = (uint8_t)pose_collides(P, poses[3 * i], poses[3 * i + 1], poses[3 * i + 2], X, bits, map.rows, map.cols,
                                    map.wpr, E);
}

struct MaskSink {
    uint8_t* img;
    int side, hx, hy;
    __device__ __forceinline__ bool span(int v, int ua, int ub) const
    {
        const int y = v + hy;
        if ((unsigned)y < (unsigned)side)
            for (int x = max(ua + hx, 0); x <= min(ub + hx, side - 1); ++x) img[y * side + x] = 255;
        return false;
    }
    __device__ __forceinline__ bool pixel(int v, int u) const { return span(v, u, u); }
};

__global__ void __launch_bounds__(kBlock) pixel_footprint_kernel(DevParams P, const double* __restrict__ angles, int64_t n,
                                                                 uint8_t* __restrict__ masks, int side,
                                                                 int32_t* __restrict__ shape_hw)
{
    const int tid = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * kBlock + tid;
    VertLds E;
    E.base = lds_dyn + tid;
    E.stride = kBlock;
    if (i >= n) return;
    MaskSink sink;
    sink.img = masks + i * (int64_t)side * side;
    sink.side = side;
    const double c = cos(angles[i]), s = sin(angles[i]);
    footprint_half_sizes(P, c, s, sink.hx, sink.hy);
    shape_hw[2 * i] = 2 * sink.hy + 1;
    shape_hw[2 * i + 1] = 2 * sink.hx + 1;
    raster_runs(P, c, s, E, sink);
}

__global__ void normalize_angle_kernel(const double* __restrict__ in, double* __restrict__ out, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = normalize_angle(in[i]);
}

__global__ void world_to_pixel_kernel(const double* __restrict__ xy, int64_t n, double ox, double oy, double inv_res,
                                      int64_t* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[2 * i] = (int64_t)rint((xy[2 * i] - ox) * inv_res);
    out[2 * i + 1] = (int64_t)rint((xy[2 * i + 1] - oy) * inv_res);
}

// ------------------------------------------------------------------------------------------------ host API
static void fill_dev_params(bcp_handle* h)
{
    const bcp_params& p = h->params;
    DevParams& d = h->dev;
    memset(&d, 0, sizeof(d));
    d.model = p.model;
    d.n_verts = p.n_verts;
    d.dynamic_model = p.dynamic_model;
    d.model_front_column_pid = p.model_front_column_pid;
    d.noise_on = p.noise_on;
    d.iteration_timeout = p.iteration_timeout;
    d.dt = p.dt;
    d.L = p.front_wheel_from_axis;
    d.max_wheel_angle = p.max_front_wheel_angle;
    d.max_wheel_speed = p.max_front_wheel_speed;
    d.max_lin_acc = p.max_linear_acceleration;
    d.max_ang_acc = p.max_angular_acceleration;
    d.p_gain = p.front_column_p_gain;
    for (int k = 0; k < 6; ++k) d.alpha[k] = p.alpha[k];
    d.sp = p.spatial_precision;
    d.ap = p.angular_precision;
    d.progress_mult = p.spatial_progress_multiplier;
    d.par_thr = -p.spatial_precision / 9;
    d.sp_prune = std::nextafter(std::nextafter(p.spatial_precision, INFINITY), INFINITY);
    const double res = h->resolution > 0 ? h->resolution : 1.0;
    for (int k = 0; k < p.n_verts; ++k) {
        d.qverts[k][0] = p.verts[k][0] / res;  // robot_footprint / map_resolution (path_tools.py:145)
        d.qverts[k][1] = p.verts[k][1] / res;
    }
}

static int check_kernel_size(const bcp_params& p, double res)
{
    double r2 = 0;
    for (int k = 0; k < p.n_verts; ++k) {
        double d2 = p.verts[k][0] * p.verts[k][0] + p.verts[k][1] * p.verts[k][1];
        if (d2 > r2) r2 = d2;
    }
    return std::sqrt(r2) / res + 2.0 <= BCP_MAX_KERNEL_HALF;
}

static size_t edge_lds_bytes(const bcp_handle* h) { return (size_t)h->params.n_verts * 2 * kBlock * sizeof(uint32_t); }

extern "C" int bcp_create(const bcp_params* params, int64_t n_envs, int device, int64_t env_id_base, bcp_handle** out)
{
    if (!params || !out) return fail(BCP_E_INVALID, "bcp_create: null argument");
    if (params->abi_version != BCP_ABI_VERSION)
        return fail(BCP_E_INVALID, "bcp_create: abi_version %d != %d", params->abi_version, BCP_ABI_VERSION);
    if (n_envs <= 0) return fail(BCP_E_INVALID, "bcp_create: n_envs must be positive");
    if (params->n_verts < 3 || params->n_verts > BCP_MAX_VERTS)
        return fail(BCP_E_INVALID, "bcp_create: n_verts %d outside [3, %d]", params->n_verts, BCP_MAX_VERTS);
    if (params->model != BCP_MODEL_TRICYCLE && params->model != BCP_MODEL_DIFFDRIVE)
        return fail(BCP_E_INVALID, "bcp_create: unknown robot model %d", params->model);
    if (!(params->dt > 0)) return fail(BCP_E_INVALID, "bcp_create: dt must be > 0 (path_tools.py:307)");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(BCP_E_NO_DEVICE, "bcp_create: no HIP device available (%s); libbcplan has no CPU path",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= count) return fail(BCP_E_INVALID, "bcp_create: device %d of %d", device, count);
    HIP_TRY(hipSetDevice(device));
    bcp_handle* h = new (std::nothrow) bcp_handle();
    if (!h) return fail(BCP_E_INVALID, "bcp_create: out of host memory");
    memset(h, 0, sizeof(*h));
    h->params = *params;
    h->n = n_envs;
    h->device = device;
    h->env_id_base = env_id_base;
    h->seed = 0;
    fill_dev_params(h);
    *out = h;
    return BCP_OK;
}

extern "C" int bcp_destroy(bcp_handle* h)
{
    if (!h) return BCP_OK;
    (void)hipSetDevice(h->device);
    if (h->bitmap) (void)hipFree(h->bitmap);
    if (h->path5) (void)hipFree(h->path5);
    delete h;
    return BCP_OK;
}

extern "C" int bcp_seed(bcp_handle* h, uint64_t seed)
{
    if (!h) return fail(BCP_E_INVALID, "bcp_seed: null handle");
    h->seed = seed;
    h->step_counter = 0;
    return BCP_OK;
}

extern "C" int bcp_set_costmaps(bcp_handle* h, const uint8_t* data, int32_t rows, int32_t cols, int32_t shared,
                                const int32_t* valid_rows, const int32_t* valid_cols, const double* origins,
                                int32_t origins_per_env, double resolution, void* stream)
{
    if (!h || !data || !origins) return fail(BCP_E_INVALID, "bcp_set_costmaps: null argument");
    if (rows <= 0 || cols <= 0 || !(resolution > 0)) return fail(BCP_E_INVALID, "bcp_set_costmaps: bad shape/resolution");
    if (!check_kernel_size(h->params, resolution))
        return fail(BCP_E_INVALID, "bcp_set_costmaps: footprint radius / resolution exceeds %d px", BCP_MAX_KERNEL_HALF);
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const int wpr = (cols + 31) / 32;
    const int64_t n_maps = shared ? 1 : h->n;
    const size_t bytes = (size_t)n_maps * rows * wpr * sizeof(uint32_t);
    if (bytes > h->bitmap_bytes) {
        if (h->bitmap) HIP_TRY(hipFree(h->bitmap));
        h->bitmap = nullptr;
        h->bitmap_bytes = 0;
        HIP_TRY(hipMalloc((void**)&h->bitmap, bytes));
        h->bitmap_bytes = bytes;
    }
    const int64_t total = n_maps * rows * wpr;
    const int threads = 256;
    const int blocks = (int)std::min<int64_t>((total + threads - 1) / threads, 65536);
    hipLaunchKernelGGL(pack_bitmap_kernel, dim3(blocks), dim3(threads), 0, s, data, h->bitmap, n_maps, rows, cols, wpr,
                       valid_rows, valid_cols);
    HIP_TRY(hipGetLastError());
    h->resolution = resolution;
    fill_dev_params(h);
    MapDesc& m = h->map;
    m.bits = h->bitmap;
    m.rows = rows;
    m.cols = cols;
    m.wpr = wpr;
    m.shared = shared ? 1 : 0;
    m.env_stride = shared ? 0 : (int64_t)rows * wpr;
    m.inv_res = 1.0 / resolution;  // anti_resolution = 1./resolution (coordinate_transformations.py:204)
    if (origins_per_env) {
        m.origins = origins;
        m.ox = m.oy = 0;
    } else {
        m.origins = nullptr;
        m.ox = origins[0];
        m.oy = origins[1];
    }
    // stage in LDS when the shared bitmap plus the edge table leaves room for >= 2 workgroups per CU
    const size_t map_bytes = (size_t)rows * wpr * sizeof(uint32_t);
    m.in_lds = (shared && map_bytes + edge_lds_bytes(h) <= 64 * 1024) ? 1 : 0;
    h->have_map = true;
    return BCP_OK;
}

extern "C" int bcp_set_paths(bcp_handle* h, const double* xytheta, const int32_t* lens, int32_t max_len, int32_t shared,
                             void* stream)
{
    if (!h || !xytheta) return fail(BCP_E_INVALID, "bcp_set_paths: null argument");
    if (max_len <= 0) return fail(BCP_E_INVALID, "bcp_set_paths: max_len must be positive");
    if (!shared && !lens) return fail(BCP_E_INVALID, "bcp_set_paths: per-env paths need lens");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = (shared ? 1 : h->n) * (int64_t)max_len;
    const size_t bytes = (size_t)total * 5 * sizeof(double);
    if (bytes > h->path5_bytes) {
        if (h->path5) HIP_TRY(hipFree(h->path5));
        h->path5 = nullptr;
        h->path5_bytes = 0;
        HIP_TRY(hipMalloc((void**)&h->path5, bytes));
        h->path5_bytes = bytes;
    }
    const int threads = 256;
    const int blocks = (int)std::min<int64_t>((total + threads - 1) / threads, 65536);
    hipLaunchKernelGGL(path_trig_kernel, dim3(blocks), dim3(threads), 0, s, xytheta, h->path5, total);
    HIP_TRY(hipGetLastError());
    h->path.pts = h->path5;
    h->path.lens = shared ? nullptr : lens;
    h->path.max_len = max_len;
    h->path.shared = shared ? 1 : 0;
    h->have_path = true;
    return BCP_OK;
}

extern "C" int bcp_bind_state(bcp_handle* h, const bcp_state* state)
{
    if (!h) return fail(BCP_E_INVALID, "bcp_bind_state: null handle");
    if (!check_state(state, h->params.model == BCP_MODEL_TRICYCLE))
        return fail(BCP_E_INVALID, "bcp_bind_state: missing state array");
    h->st = to_dev_state(state);
    h->have_state = true;
    return BCP_OK;
}

extern "C" int bcp_bind_initial_state(bcp_handle* h, const bcp_state* initial)
{
    if (!h) return fail(BCP_E_INVALID, "bcp_bind_initial_state: null handle");
    if (!check_state(initial, h->params.model == BCP_MODEL_TRICYCLE))
        return fail(BCP_E_INVALID, "bcp_bind_initial_state: missing state array");
    h->init = to_dev_state(initial);
    h->have_init = true;
    return BCP_OK;
}

extern "C" int bcp_reset_masked(bcp_handle* h, const uint8_t* mask, void* stream)
{
    if (!h) return fail(BCP_E_INVALID, "bcp_reset_masked: null handle");
    if (!h->have_state || !h->have_init) return fail(BCP_E_STATE, "bcp_reset_masked: state / initial state not bound");
    HIP_TRY(hipSetDevice(h->device));
    const int threads = 256;
    const int blocks = (int)((h->n + threads - 1) / threads);
    hipLaunchKernelGGL(reset_kernel, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, h->st, h->init, mask, h->n,
                       (int)(h->params.model == BCP_MODEL_TRICYCLE));
    HIP_TRY(hipGetLastError());
    return BCP_OK;
}

static int launch_step(bcp_handle* h, const bcp_step_io* io, uint32_t flags, hipStream_t s)
{
    StepArgs a;
    a.P = h->dev;
    a.map = h->map;
    a.path = h->path;
    a.st = h->st;
    a.init = h->init;
    a.n = h->n;
    a.actions = io->actions;
    a.noise_z = io->noise_z;
    a.noise_z_out = io->noise_z_out;
    a.reward = io->reward;
    a.done = io->done;
    a.collided_now = io->collided_now;
    a.err = io->err;
    a.flags = flags;
    a.seed = h->seed;
    a.step_counter = h->step_counter;
    a.env_id_base = h->env_id_base;
    const size_t lds = edge_lds_bytes(h) + (h->map.in_lds ? (size_t)h->map.rows * h->map.wpr * sizeof(uint32_t) : 0);
    const int blocks = (int)((h->n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(step_kernel, dim3(blocks), dim3(kBlock), lds, s, a);
    h->step_counter += 1;
    return BCP_OK;
}

static int check_step(bcp_handle* h, const bcp_step_io* io, uint32_t flags, const char* who)
{
    if (!h || !io) return fail(BCP_E_INVALID, "%s: null argument", who);
    if (!h->have_map || !h->have_path || !h->have_state)
        return fail(BCP_E_STATE, "%s: costmaps, paths and state must be set first", who);
    if ((flags & BCP_STEP_AUTO_RESET) && !h->have_init)
        return fail(BCP_E_STATE, "%s: BCP_STEP_AUTO_RESET needs bcp_bind_initial_state", who);
    if (!io->actions || !io->reward || !io->done) return fail(BCP_E_INVALID, "%s: actions/reward/done are required", who);
    return BCP_OK;
}

extern "C" int bcp_step(bcp_handle* h, const bcp_step_io* io, uint32_t flags, void* stream)
{
    int rc = check_step(h, io, flags, "bcp_step");
    if (rc != BCP_OK) return rc;
    HIP_TRY(hipSetDevice(h->device));
    launch_step(h, io, flags, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return BCP_OK;
}

extern "C" int bcp_time_steps(bcp_handle* h, const bcp_step_io* io, uint32_t flags, int32_t steps, void* stream,
                              float* avg_ms)
{
    int rc = check_step(h, io, flags, "bcp_time_steps");
    if (rc != BCP_OK) return rc;
    if (steps <= 0 || !avg_ms) return fail(BCP_E_INVALID, "bcp_time_steps: steps must be positive");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, s));
    for (int k = 0; k < steps; ++k) launch_step(h, io, flags, s);
    HIP_TRY(hipEventRecord(e1, s));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    HIP_TRY(hipEventDestroy(e0));
    HIP_TRY(hipEventDestroy(e1));
    HIP_TRY(hipGetLastError());
    *avg_ms = ms / (float)steps;
    return BCP_OK;
}

extern "C" int bcp_robot_step(bcp_handle* h, double* state7_io, int64_t n, const double* actions, const double* noise_z,
                              int32_t* err, void* stream)
{
    if (!h || !state7_io || !actions || n <= 0) return fail(BCP_E_INVALID, "bcp_robot_step: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    const int blocks = (int)((n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(robot_step_kernel, dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream, h->dev, state7_io, n,
                       actions, noise_z, err);
    HIP_TRY(hipGetLastError());
    return BCP_OK;
}

extern "C" int bcp_pose_collides(bcp_handle* h, const double* poses, int64_t n, uint8_t* out, void* stream)
{
    if (!h || !poses || !out || n <= 0) return fail(BCP_E_INVALID, "bcp_pose_collides: bad argument");
    if (!h->have_map) return fail(BCP_E_STATE, "bcp_pose_collides: costmaps not set");
    HIP_TRY(hipSetDevice(h->device));
    const int blocks = (int)((n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(pose_collides_kernel, dim3(blocks), dim3(kBlock), edge_lds_bytes(h), (hipStream_t)stream, h->dev,
                       h->map, poses, n, h->n, out);
    HIP_TRY(hipGetLastError());
    return BCP_OK;
}

extern "C" int bcp_pixel_footprint(bcp_handle* h, const double* angles, int64_t n, double resolution, uint8_t* masks,
                                   int32_t side, int32_t* shape_hw, void* stream)
{
    if (!h || !angles || !masks || !shape_hw || n <= 0 || side <= 0)
        return fail(BCP_E_INVALID, "bcp_pixel_footprint: bad argument");
    if (!(resolution > 0) || !check_kernel_size(h->params, resolution))
        return fail(BCP_E_INVALID, "bcp_pixel_footprint: footprint radius / resolution exceeds %d px", BCP_MAX_KERNEL_HALF);
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    DevParams P = h->dev;
    for (int k = 0; k < h->params.n_verts; ++k) {
        P.qverts[k][0] = h->params.verts[k][0] / resolution;
        P.qverts[k][1] = h->params.verts[k][1] / resolution;
    }
    HIP_TRY(hipMemsetAsync(masks, 0, (size_t)n * side * side, s));
    const int blocks = (int)((n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(pixel_footprint_kernel, dim3(blocks), dim3(kBlock), edge_lds_bytes(h), s, P, angles, n, masks, side,
                       shape_hw);
    HIP_TRY(hipGetLastError());
    return BCP_OK;
}

extern "C" int bcp_normalize_angle(bcp_handle* h, const double* in, double* out, int64_t n, void* stream)
{
    if (!h || !in || !out || n <= 0) return fail(BCP_E_INVALID, "bcp_normalize_angle: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    hipLaunchKernelGGL(normalize_angle_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in,
                       out, n);
    HIP_TRY(hipGetLastError());
    return BCP_OK;
}

extern "C" int bcp_world_to_pixel(bcp_handle* h, const double* xy, int64_t n, const double* origin, double resolution,
                                  int64_t* out, void* stream)
{
    if (!h || !xy || !origin || !out || n <= 0 || !(resolution > 0))
        return fail(BCP_E_INVALID, "bcp_world_to_pixel: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    hipLaunchKernelGGL(world_to_pixel_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, xy, n,
                       origin[0], origin[1], 1.0 / resolution, out);
    HIP_TRY(hipGetLastError());
    return BCP_OK;
}
