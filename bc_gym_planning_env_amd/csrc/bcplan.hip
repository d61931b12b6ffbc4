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
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "bcp_device.h"
#include "bcp_raster.h"
#include "bcp_coop.h"

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
    double *pose_seen, *state_seen;         // [3][n] / [7][n], delays > 0 only
    double *control_q, *pose_q, *state_q;   // [delay][width][n]
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
    const double* pts;   // [len][5] = x, y, theta, cos(theta), sin(theta); shared or [N][max_len][5]
    const double* bbox;  // [8] = xmin, xmax, ymin, ymax of the way points, then the bucket grid x0, 1/wx, y0, 1/wy
    const int16_t* index; // [2 axes][kPathBuckets][2] = first / last way point index that can be reached from a bucket
    const int32_t* lens;
    int32_t max_len, shared;
};

struct Pending;

// Everything a step needs that only changes when the caller re-binds something.  It lives in DEVICE memory (uploaded
// when dirty) and the kernels get a pointer: kernel arguments sit in host memory on this platform, and a kernel that
// takes kilobytes of arguments by value pays a PCIe-latency scalar load every time it touches a new field.
struct StepStatic {
    DevParams P;
    MapDesc map;
    CullDesc cull;
    PathDesc path;
    DevState st, init;
    int64_t n;
    int64_t env_id_base;
    int32_t exact_mode, dense_threshold, wide;
    int32_t pending_cap;       // slots per shard
    int32_t lds_path_doubles;  // > 0: the shared path (max_len * 5 doubles) is staged in LDS by the fast step kernel
    struct Pending* pending;   // [kShards][pending_cap] parking slots for undecided envs (nullptr: no second kernel)
    // geometry pool (bcp_set_geometry_pool): env i uses entry geom_of_env[i] of the non-shared map / path / initial
    // state arrays; a reset moves it to next_geom[entry].  nullptr: env i uses entry i.
    int32_t* geom_of_env;
    const int32_t* next_geom;
};

// Per-launch kernel arguments (small).
struct StepArgs {
    const StepStatic* S;
    const void* actions;
    const double* noise_z;
    double* noise_z_out;
    double* reward;
    uint8_t* done;
    uint8_t* collided_now;
    int32_t* err;
    int32_t* pending_count;    // [kShards] this step's counters of parked envs (one per shard: no hot atomic)
    int32_t* pending_next;     // [kShards] the next step's counters (the two sets alternate); kernel 1 zeroes them
    // adaptive split between "settle in place" and "park for kernel 2" (nullptr: S->dense_threshold is used as is):
    // kernel 2 of step t counts the undecided poses of step t and picks the threshold of step t + 1
    const int32_t* threshold_now;
    int32_t* threshold_next;
    int32_t* inplace_count;    // undecided poses settled inside kernel 1 this step (the parked ones are in pending_count)
    int32_t* inplace_next;     // next step's counter; kernel 1 zeroes it
    uint64_t seed, step_counter;
    uint32_t flags;
};

// Ablation switches in the upper half of the step flags (tools/ablate*.py time the step with stages removed; results
// are then WRONG by construction).  Not part of the ABI: bcplan.h only defines bits 0-1.
enum : uint32_t {
    kAblateNoCollision = 1u << 16,   // skip pose_collides altogether
    kAblateNoReward = 1u << 17,      // skip the reward scan
    kAblateNoCoop = 1u << 19,        // kernel 2: skip the cooperative rasteriser
    kAblateNoPark = 1u << 21,        // kernel 1: do not park undecided envs
    kAblateNoClassify = 1u << 22     // kernel 1: skip the distance-field lookups
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
    double* path_bbox;     // owned
    size_t path_bbox_bytes;
    int16_t* path_index;   // owned
    size_t path_index_bytes;
    uint8_t* edt;          // owned: distance transform of the shared costmap (padded)
    size_t edt_bytes;
    uint8_t* edt_col;      // owned scratch of the transform
    size_t edt_col_bytes;
    MapDesc map;
    CullDesc cull;
    PathDesc path;
    DevState st, init;
    StepStatic host_static;   // host image of the device-resident step parameters
    StepStatic* dev_static;   // owned
    bool static_dirty;        // host_static must be rebuilt and uploaded before the next step
    void* pending;            // owned: Pending[n]
    int32_t* pending_count;   // owned: two alternating sets of kShards counters
    int32_t pending_cap;      // parking slots per shard
    int32_t defer;            // settle undecided envs in a second kernel (shared map with distance field)
    int32_t exact_mode;       // 0 auto, 1 cooperative only, 2 per-thread only
    int32_t dense_threshold;  // auto: more ambiguous lanes than this in a wave -> per-thread rasteriser
    int32_t adaptive;         // the threshold above is only the fallback: kernel 2 re-decides every step
    int32_t* adapt;           // owned: [2] thresholds + [2] in-place counters, alternating by step parity
    int32_t cull_enabled;
    int32_t wide;             // kernel image may exceed 96 px: 8-word row masks in the cooperative path
    int32_t* ego_bins;        // owned: [2][bins] image counts / first slots per map entry (egocentric views)
    int64_t ego_bins_cap;
    int32_t* ego_order;       // owned: [2][images] rank within the bin / images grouped by map entry
    int64_t ego_order_cap;
    const uint8_t* map_data;  // caller-owned raw costmap(s) as given to bcp_set_costmaps (egocentric views read them)
    const int32_t* map_valid_rows;
    const int32_t* map_valid_cols;
    int32_t n_geoms;          // > 0: geometry pool of that many entries
    int32_t* geom_of_env;     // caller-owned device int32 [n]
    const int32_t* next_geom; // caller-owned device int32 [n_geoms] or nullptr
};

// number of entries of a non-shared map / path / initial-state array
static int64_t n_slots(const bcp_handle* h) { return h->n_geoms > 0 ? h->n_geoms : h->n; }

static DevState to_dev_state(const bcp_state* s)
{
    DevState d;
    d.x = s->x; d.y = s->y; d.angle = s->angle; d.v = s->v; d.w = s->w;
    d.steer = s->steering_motor_command; d.wheel = s->wheel_angle; d.min_dist = s->min_spat_dist_so_far;
    d.target_idx = s->target_idx; d.cur_iter = s->current_iter; d.collided = s->robot_collided;
    d.pose_seen = s->pose_seen; d.state_seen = s->robot_state_seen;
    d.control_q = s->control_queue; d.pose_q = s->poses_queue; d.state_q = s->robot_state_queue;
    return d;
}

static int check_state(const bcp_state* s, int tricycle, const bcp_params* p = nullptr, bool queues = true)
{
    if (!s) return 0;
    if (p) {   // delays > 0 need the arrays State exposes, and (for the live state) the queues
        if (p->pose_delay > 0 && (!s->pose_seen || (queues && !s->poses_queue))) return 0;
        if (p->state_delay > 0 && (!s->robot_state_seen || (queues && !s->robot_state_queue))) return 0;
        if (p->control_delay > 0 && queues && !s->control_queue) return 0;
    }
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

constexpr int kPathBuckets = 64;

// Per path: bounding box of the way points and a 1-D bucket grid per axis over [min - sp, max + sp].
__global__ void path_bbox_kernel(const double* __restrict__ xyt, const int32_t* __restrict__ lens, int max_len,
                                 int64_t n_paths, double sp_prune, double* __restrict__ bbox)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_paths) return;
    const int m = lens ? lens[p] : max_len;
    const double* q = xyt + p * (int64_t)max_len * 3;
    double x0 = INFINITY, x1 = -INFINITY, y0 = INFINITY, y1 = -INFINITY;
    for (int j = 0; j < m; ++j) {
        x0 = fmin(x0, q[3 * j]);
        x1 = fmax(x1, q[3 * j]);
        y0 = fmin(y0, q[3 * j + 1]);
        y1 = fmax(y1, q[3 * j + 1]);
    }
    double* o = bbox + 8 * p;
    o[0] = x0;
    o[1] = x1;
    o[2] = y0;
    o[3] = y1;
    const double wx = fmax((x1 - x0 + 2.0 * sp_prune) / kPathBuckets, 1e-9);
    const double wy = fmax((y1 - y0 + 2.0 * sp_prune) / kPathBuckets, 1e-9);
    o[4] = x0 - sp_prune;
    o[5] = 1.0 / wx;
    o[6] = y0 - sp_prune;
    o[7] = 1.0 / wy;
}

// index[p][axis][b] = {first, last} way point whose coordinate lies within sp of bucket b (widened by a guard band
// that swallows the rounding of the bucket computation); {32767, -1} when there is none.  Any way point with
// |x_j - x| <= sp_prune for a query x that falls into bucket b is inside [first, last].
__global__ void path_index_kernel(const double* __restrict__ xyt, const int32_t* __restrict__ lens, int max_len,
                                  int64_t n_paths, double sp_prune, const double* __restrict__ bbox,
                                  int16_t* __restrict__ index)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_paths * 2 * kPathBuckets) return;
    const int b = (int)(t % kPathBuckets);
    const int axis = (int)((t / kPathBuckets) % 2);
    const int64_t p = t / (2 * kPathBuckets);
    const int m = lens ? lens[p] : max_len;
    const double* q = xyt + p * (int64_t)max_len * 3;
    const double o = bbox[8 * p + 4 + 2 * axis], w = 1.0 / bbox[8 * p + 5 + 2 * axis];
    const double guard = 1e-6 * w + 1e-12;
    const double lo = o + b * w - sp_prune - guard, hi = o + (b + 1) * w + sp_prune + guard;
    int first = 32767, last = -1;
    for (int j = 0; j < m; ++j) {
        const double v = q[3 * j + axis];
        if (v >= lo && v <= hi) {
            first = min(first, j);
            last = j;
        }
    }
    index[2 * t] = (int16_t)first;
    index[2 * t + 1] = (int16_t)last;
}

// find_last_reached restricted to j >= target (utilities/path_tools.py:408-448): the reward only asks whether the
// LAST reached index is >= target_idx (envs/base/reward.py:234), so indices below target never matter.
// A way point can only be reached when |x_j - x| and |y_j - y| are both below spatial_precision, so the scan is
// confined to the index window the two bucket tables allow for this pose (usually a handful of way points).
// index window [lo, hi] of the way points that can be within spatial_precision of (x, y); empty when lo > hi
struct PathWindow {
    int lo, hi;
};

__device__ __forceinline__ PathWindow path_window(const DevParams& P, const double* __restrict__ bbox,
                                                  const int16_t* __restrict__ index, double x, double y)
{
    PathWindow w;
    w.lo = 0;
    w.hi = -1;
    if (x < bbox[0] - P.sp_prune || x > bbox[1] + P.sp_prune || y < bbox[2] - P.sp_prune || y > bbox[3] + P.sp_prune)
        return w;  // farther than spatial_precision from the bounding box of the whole path
    const int bx = min(max((int)floor((x - bbox[4]) * bbox[5]), 0), kPathBuckets - 1);
    const int by = min(max((int)floor((y - bbox[6]) * bbox[7]), 0), kPathBuckets - 1);
    const int16_t* ix = index + 2 * bx;
    const int16_t* iy = index + 2 * (kPathBuckets + by);
    w.lo = max((int)ix[0], (int)iy[0]);
    w.hi = min((int)ix[1], (int)iy[1]);
    return w;
}

template <typename PathPtr>
__device__ __forceinline__ int last_reached_from(const DevParams& P, PathPtr path, PathWindow w, int m, int target,
                                                 double x, double y, double th)
{
    if (target > m - 1) return -1;
    const int lo = max(w.lo, target);
    const int hi = min(w.hi, m - 1);
    for (int j = hi; j >= lo; --j) {
        const PathPtr s = path + 5 * j;
        // all five values of the way point are fetched up front (one latency instead of three dependent ones)
        const double sx = s[0], sy = s[1], sth = s[2], sc = s[3], ss = s[4];
        const double dx = sx - x, dy = sy - y;
        // the three reach conditions are independent predicates; evaluate the cheap ones first
        if (fabs(dx) > P.sp_prune || fabs(dy) > P.sp_prune) continue;   // then hypot(dx,dy) >= sp
        const double par = sc * (x - sx) + ss * (y - sy);               // path_tools.py:405
        if (!(par >= P.par_thr)) continue;
        const double q = dx * dx + dy * dy;
        bool near = q < P.sp2_lo;
        if (!near && q <= P.sp2_hi) near = hypot(dx, dy) < P.sp;        // too close to call from q
        if (!near) continue;
        if (fabs(normalize_angle(th - sth)) < P.ap) return j;
    }
    return -1;
}

// ContinuousRewardProvider.reward (envs/base/reward.py:214-259)
template <typename PathPtr>
__device__ __forceinline__ double reward_step(const DevParams& P, PathPtr path, PathWindow w, int m, double x, double y,
                                              double th, double& min_dist, int& target)
{
    if (target > m - 1) return 0.0;
    const int last = last_reached_from(P, path, w, m, target, x, y, th);
    if (last >= 0) {
        target = last + 1;
        if (!(target > m - 1)) {
            const PathPtr g = path + 5 * target;
            min_dist = hypot(g[0] - x, g[1] - y);
        } else {
            min_dist = 0.0;
        }
        return 1.0;
    }
    const PathPtr g = path + 5 * target;
    const double d = hypot(g[0] - x, g[1] - y);
    if (d < min_dist) {
        const double r = min_dist - d;
        min_dist = d;
        return r * P.progress_mult;
    }
    return 0.0;
}

// ContinuousRewardPurePursuitProvider.reward (envs/base/reward.py:330-353) with update_goal (:125-139): the target is
// the first way point from target_idx on that is more than 2 m away (np.linalg.norm = sqrt of an fma-contracted
// 2-term dot product, like every 2-element np.dot in this code base), the goal is always the LAST way point.
template <typename PathPtr>
__device__ __forceinline__ double reward_pure_pursuit(PathPtr path, int m, double x, double y, bool collided,
                                                      double& min_dist, int& target)
{
    int found = m - 1;
    for (int i = target; i < m; ++i) {
        const double dx = path[5 * i] - x, dy = path[5 * i + 1] - y;
        if (sqrt(fma(dy, dy, dx * dx)) > 2.) {
            found = i;
            break;
        }
    }
    target = found;
    const PathPtr g = path + 5 * (m - 1);
    const double dist = hypot(g[0] - x, g[1] - y);
    double reward = -0.05;
    reward += min_dist - dist;
    min_dist = dist;
    if (collided) reward -= 100;
    return reward;
}

// _get_element_from_list_with_delay (envs/base/env.py:27-49) for the k-th push since the last reset: queue q is
// [delay][W][n]; element k lives in slot (k - 1) % delay.  `v` holds the new element on entry, the delayed one on exit.
template <int W>
__device__ __forceinline__ void fifo_delay(double* __restrict__ q, int delay, int64_t n, int64_t i, int k, double (&v)[W])
{
    if (delay <= 0) return;
    const int slot = (k - 1) % delay;
    double* cell = q + ((int64_t)slot * W) * n + i;
    if (k <= delay) {   // the list is not longer than `delay` yet: append, hand back the first element
#pragma unroll
        for (int c = 0; c < W; ++c) cell[c * n] = v[c];
        if (k > 1) {
#pragma unroll
            for (int c = 0; c < W; ++c) v[c] = q[c * n + i];
        }
    } else {            // pop(0): element k - delay, whose slot the new element takes
#pragma unroll
        for (int c = 0; c < W; ++c) {
            const double first = cell[c * n];
            cell[c * n] = v[c];
            v[c] = first;
        }
    }
}

#ifdef BCP_DIAG
__device__ unsigned long long g_diag[1024 * 8];
__device__ unsigned long long g_diag1[1024 * 8];
#define DIAG1_STAMP(k) do { if (threadIdx.x == 0) g_diag1[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define DIAG_STAMP(k) do { if (threadIdx.x == 0) g_diag[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int bcp_diag_read(unsigned long long* out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_diag), sizeof(g_diag));
}
extern "C" int bcp_diag1_read(unsigned long long* out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_diag1), sizeof(g_diag1));
}
#else
#define DIAG_STAMP(k) do { } while (0)
#define DIAG1_STAMP(k) do { } while (0)
#endif

constexpr int kShards = 64;  // a wave parks into shard (block index % kShards)

// dynamic LDS of the collision kernels:
//   [lethal bitmap words (when the shared map is staged)] [qverts: n_verts * 2 doubles] [vertex scratch of the
//   per-thread rasteriser: n_verts * 2 * kBlock words]
extern __shared__ uint32_t lds_dyn[];

struct CollisionLds {
    bool staged;      // the shared lethal bitmap sits at LDS offset 0
    LdsWords bits;
    LdsF64 qverts;
    VertLds scratch;
};

__device__ __forceinline__ CollisionLds collision_lds_setup(const DevParams& P, const MapDesc& map, int tid)
{
    CollisionLds L;
    const int map_words = map.in_lds ? map.rows * map.wpr : 0;
    const LdsU32 lds = (LdsU32)lds_dyn;
    for (int k = tid; k < map_words; k += kBlock) lds[k] = map.bits[k];
    const int q_off = (map_words + 1) & ~1;  // 8-byte alignment for the doubles
    __attribute__((address_space(3))) double* q = (__attribute__((address_space(3))) double*)(lds + q_off);
    for (int k = tid; k < 2 * P.n_verts; k += kBlock) q[k] = P.qverts[k >> 1][k & 1];
    L.staged = map.in_lds != 0;
    L.bits = lds;
    L.qverts = q;
    L.scratch.base = lds + q_off + 4 * P.n_verts + tid;
    L.scratch.stride = kBlock;
    __syncthreads();
    return L;
}

static size_t collision_lds_bytes(int n_verts, int in_lds, int rows, int wpr)
{
    size_t words = in_lds ? (size_t)rows * wpr : 0;
    words = (words + 1) & ~(size_t)1;
    words += 4 * (size_t)n_verts;            // qverts (doubles)
    words += 2 * (size_t)n_verts * kBlock;   // per-thread vertex scratch
    return words * sizeof(uint32_t);
}

// pose_collides (envs/base/env.py:464-489) for the pose held by each lane.  EVERY lane of the wave must call this
// (inactive lanes pass active = false): ambiguous poses are settled one at a time by the whole wave.
__device__ __forceinline__ bool collides_wave(const DevParams& P, const MapDesc& map, const CullDesc& cull,
                                              const CollisionLds& L, int exact_mode, int dense_threshold, bool wide,
                                              bool active, int64_t env, double x, double y, double th)
{
    double ox = map.ox, oy = map.oy;
    if (map.origins) {
        ox = map.origins[2 * env + 0];
        oy = map.origins[2 * env + 1];
    }
    const int px = (int)rint((x - ox) * map.inv_res);   // world_to_pixel, coordinate_transformations.py:185-205
    const int py = (int)rint((y - oy) * map.inv_res);
    const double c = cos(th), s = sin(th);
    int cls = active ? classify(cull, map.shared ? 0 : env, map.rows, map.cols, px, py, c, s) : kFree;
    bool hit = cls == kHit;
    uint64_t amb = __ballot(cls == kAmbiguous);
    if (amb == 0) return hit;
    const bool dense = exact_mode == 2 || (exact_mode == 0 && (int)__popcll(amb) > dense_threshold);
    if (dense) {
        // many undecided lanes: one per-thread rasteriser pass settles them all at once
        if (cls == kAmbiguous) {
            if (L.staged) {
                CollisionSink<LdsWords> sink{L.bits, map.rows, map.cols, map.wpr, px, py};
                hit = raster_runs(P, c, s, L.scratch, sink);
            } else {
                const uint32_t* words = map.bits + (map.shared ? 0 : env * map.env_stride);
                CollisionSink<const uint32_t*> sink{words, map.rows, map.cols, map.wpr, px, py};
                hit = raster_runs(P, c, s, L.scratch, sink);
            }
        }
        return hit;
    }
    // few undecided lanes: the wave rasterises them cooperatively, one pose at a time
    const int ln = lane_id();
    const double vqx = ln < P.n_verts ? L.qverts[2 * ln] : 0.0, vqy = ln < P.n_verts ? L.qverts[2 * ln + 1] : 0.0;
    while (amb) {
        const int src = __ffsll((unsigned long long)amb) - 1;
        amb &= amb - 1;
        const double c_ = bcast_d(c, src), s_ = bcast_d(s, src);
        const int px_ = bcast_i(px, src), py_ = bcast_i(py, src);
        bool h;
        if (L.staged) {
            h = coop_collides(P, vqx, vqy, c_, s_, px_, py_, L.bits, map.rows, map.cols, map.wpr, wide);
        } else {
            const int64_t env_ = ((int64_t)bcast_i((int)(env >> 32), src) << 32) | (uint32_t)bcast_i((int)env, src);
            const uint32_t* words = map.bits + (map.shared ? 0 : env_ * map.env_stride);
            h = coop_collides(P, vqx, vqy, c_, s_, px_, py_, words, map.rows, map.cols, map.wpr, wide);
        }
        if (lane_id() == src) hit = h;
    }
    return hit;
}

// One env's state after the robot model ran, before the collision verdict is known.
struct Pending {
    double c, s;        // cos / sin of the new heading (as used by the classification)
    int32_t px, py;     // world_to_pixel of the new position
    Robot r;            // after robot.step()
    Pose old;           // pose before the step (rollback target)
    double min_dist;
    double z[3];
    int32_t target, iter, err, drawn;
    int32_t collided;   // sticky flag before this step
    int32_t env_lo, env_hi;
    int32_t geom;       // geometry-pool entry of the env during this step (pool mode only)
};

// entry of the non-shared map / path arrays that env i uses
__device__ __forceinline__ int64_t slot_of(const StepStatic* S, int64_t i, const Pending& q)
{
    return S->geom_of_env ? (int64_t)q.geom : i;
}

// Everything of PlanEnv.step() that follows pose_collides(): rollback (env.py:458-459), bookkeeping and delay queues
// (:377-396), reward (:352), done (:407-419), outputs, optional reset, state write-back.  Runs on one lane for env i.
// PLAIN = true (the two-kernel step: no delays, continuous reward provider -- see step_uses_deferral) compiles the
// delay queues and the pure-pursuit branch out.
template <bool PLAIN>
__device__ __forceinline__ void finalize_env(const StepArgs& a, int64_t i, Pending& q, bool hit, LdsF64 lds_path = nullptr,
                                             const PathWindow* free_window = nullptr)
{
    const DevParams& P = a.S->P;
    const int pose_delay = PLAIN ? 0 : P.pose_delay, state_delay = PLAIN ? 0 : P.state_delay;
    const bool pure_pursuit = !PLAIN && P.reward_provider == BCP_REWARD_PURE_PURSUIT;
    const bool tri = P.model == BCP_MODEL_TRICYCLE;
    const int64_t n = a.S->n;
    Robot& r = q.r;
    if (hit) {  // robot.set_pose(*old_position): pose restored, v = w = 0 (tricycle_model.py:471-476)
        r.p = q.old;
        r.v = 0.0;
        r.w = 0.0;
    }
    int iter = q.iter + 1;
    bool collided = q.collided != 0 || hit;
    double min_dist = q.min_dist;
    int target = q.target;
    // State.pose / State.robot_state: what the reward provider and the observation see (env.py:377-394)
    double seen[3] = {r.p.x, r.p.y, r.p.th};
    double seen_rs[7] = {r.p.x, r.p.y, r.p.th, r.v, r.w, r.steer, r.wheel};
    if (pose_delay) fifo_delay<3>(a.S->st.pose_q, pose_delay, n, i, iter, seen);
    if (state_delay) fifo_delay<7>(a.S->st.state_q, state_delay, n, i, iter, seen_rs);

    // shared path: uniform pointers (scalar cache); private paths: per-lane pointers
    double rew = 0.0;
    int m;
    bool goal;
    const int64_t g = slot_of(a.S, i, q);
    const double* pts = a.S->path.pts + (a.S->path.shared ? 0 : g * (int64_t)a.S->path.max_len * 5);
    m = a.S->path.shared ? a.S->path.max_len : a.S->path.lens[g];
    if (pure_pursuit) {
        if (!(a.flags & kAblateNoReward)) rew = reward_pure_pursuit(pts, m, seen[0], seen[1], collided, min_dist, target);
        goal = hypot(pts[5 * (m - 1)] - seen[0], pts[5 * (m - 1) + 1] - seen[1]) < 1.0;   // done(), reward.py:141-150
    } else {
        if (!(a.flags & kAblateNoReward)) {
            // way-point window of the pose: the caller may have looked it up already for the un-rolled-back pose
            const double* bbox = a.S->path.bbox + (a.S->path.shared ? 0 : g * 8);
            const int16_t* index = a.S->path.index + (a.S->path.shared ? 0 : g * (int64_t)(4 * kPathBuckets));
            const PathWindow w =
                (free_window && !hit && !pose_delay) ? *free_window : path_window(P, bbox, index, seen[0], seen[1]);
            if (lds_path && a.S->path.shared)  // way points staged in LDS by the step kernel
                rew = reward_step(P, lds_path, w, m, seen[0], seen[1], seen[2], min_dist, target);
            else
                rew = reward_step(P, pts, w, m, seen[0], seen[1], seen[2], min_dist, target);
        }
        goal = target > m - 1;
    }
    const bool done = goal || (iter >= P.iteration_timeout) || collided;
    if (free_window) DIAG1_STAMP(5);

    a.reward[i] = rew;
    a.done[i] = (uint8_t)done;
    if (a.collided_now) a.collided_now[i] = (uint8_t)hit;
    if (a.err) a.err[i] = q.err;
    if (a.noise_z_out) {
        const double nan = __builtin_nan("");
        a.noise_z_out[3 * i + 0] = (q.drawn & 1) ? q.z[0] : nan;
        a.noise_z_out[3 * i + 1] = (q.drawn & 2) ? q.z[1] : nan;
        a.noise_z_out[3 * i + 2] = (q.drawn & 4) ? q.z[2] : nan;
    }

    if (done && (a.flags & BCP_STEP_AUTO_RESET)) {  // PlanEnv.reset(): set_state(initial_state) (env.py:293-303)
        int64_t k = i;
        if (a.S->geom_of_env) {  // RandomMiniEnv.reset(): the env moves on to its next geometry (mini_env.py:469-481).
            // Computed from the entry the step started with, so kernel 2 redoing an env that kernel 1 already reset
            // lands on the same geometry (a hit always ends the episode, so both reset or neither does).
            k = a.S->next_geom ? a.S->next_geom[g] : g;
            a.S->geom_of_env[i] = (int32_t)k;
        }
        r.p.x = a.S->init.x[k];
        r.p.y = a.S->init.y[k];
        r.p.th = a.S->init.angle[k];
        r.v = a.S->init.v[k];
        r.w = a.S->init.w[k];
        if (tri) {
            r.steer = a.S->init.steer[k];
            r.wheel = a.S->init.wheel[k];
        }
        min_dist = a.S->init.min_dist[k];
        target = a.S->init.target_idx[k];
        iter = a.S->init.cur_iter[k];
        collided = a.S->init.collided[k] != 0;
        // the restored State exposes the initial pose / robot state; its queues are empty (pushes restart at k = 1)
        seen[0] = r.p.x;
        seen[1] = r.p.y;
        seen[2] = r.p.th;
        seen_rs[0] = r.p.x;
        seen_rs[1] = r.p.y;
        seen_rs[2] = r.p.th;
        seen_rs[3] = r.v;
        seen_rs[4] = r.w;
        seen_rs[5] = r.steer;
        seen_rs[6] = r.wheel;
    }

    if (free_window) DIAG1_STAMP(6);
    a.S->st.x[i] = r.p.x;
    a.S->st.y[i] = r.p.y;
    a.S->st.angle[i] = r.p.th;
    a.S->st.v[i] = r.v;
    a.S->st.w[i] = r.w;
    if (tri) {
        a.S->st.steer[i] = r.steer;
        a.S->st.wheel[i] = r.wheel;
    }
    a.S->st.min_dist[i] = min_dist;
    a.S->st.target_idx[i] = target;
    a.S->st.cur_iter[i] = iter;
    a.S->st.collided[i] = (uint8_t)collided;
    if (pose_delay) {
#pragma unroll
        for (int c = 0; c < 3; ++c) a.S->st.pose_seen[c * n + i] = seen[c];
    }
    if (state_delay) {
#pragma unroll
        for (int c = 0; c < 7; ++c) a.S->st.state_seen[c * n + i] = seen_rs[c];
    }
}

// ---- loads shared by the step kernels ------------------------------------------------------------------------
template <bool PLAIN>
__device__ __forceinline__ void load_env(const StepArgs& a, int64_t i, bool active, Pending& q, double& cmd0, double& cmd1)
{
    const DevParams& P = a.S->P;
    Robot& r = q.r;
    r.p.x = a.S->st.x[i];
    r.p.y = a.S->st.y[i];
    r.p.th = a.S->st.angle[i];
    r.v = a.S->st.v[i];
    r.w = a.S->st.w[i];
    const bool tri = P.model == BCP_MODEL_TRICYCLE;
    r.steer = tri ? a.S->st.steer[i] : 0.0;
    r.wheel = tri ? a.S->st.wheel[i] : 0.0;
    q.min_dist = a.S->st.min_dist[i];
    q.target = a.S->st.target_idx[i];
    q.iter = a.S->st.cur_iter[i];
    q.collided = a.S->st.collided[i] != 0;
    q.geom = a.S->geom_of_env ? a.S->geom_of_env[i] : 0;
    if (a.flags & BCP_STEP_ACTIONS_F32) {
        const float2 c = reinterpret_cast<const float2*>(a.actions)[i];
        cmd0 = (double)c.x;
        cmd1 = (double)c.y;
    } else {
        const double2 c = reinterpret_cast<const double2*>(a.actions)[i];
        cmd0 = c.x;
        cmd1 = c.y;
    }
    if (!PLAIN && P.control_delay && active) {   // the robot executes the command given control_delay steps ago (env.py:371-373)
        double cmd[2] = {cmd0, cmd1};
        fifo_delay<2>(a.S->st.control_q, P.control_delay, a.S->n, i, q.iter + 1, cmd);
        cmd0 = cmd[0];
        cmd1 = cmd[1];
    }
    q.z[0] = q.z[1] = q.z[2] = 0.0;
    if (P.noise_on) {
        if (a.noise_z) {
            q.z[0] = a.noise_z[3 * i + 0];
            q.z[1] = a.noise_z[3 * i + 1];
            q.z[2] = a.noise_z[3 * i + 2];
        } else {
            device_normals(a.seed, (uint64_t)(a.S->env_id_base + i), a.step_counter, q.z);
        }
    }
}

// General step kernel: robot model, collision settled in place by collides_wave (distance-field classification when
// there is one, then the cooperative / per-thread exact rasterisers), reward, done, write-back.  Used when there
// is no distance field, when the batch is too small to need load balancing, or when a mode is forced.
__global__ void __launch_bounds__(kBlock) step_kernel(const StepArgs a)
{
    const DevParams& P = a.S->P;
    const int tid = threadIdx.x;
    const int64_t gi = (int64_t)blockIdx.x * kBlock + tid;
    const bool active = gi < a.S->n;
    const int64_t i = active ? gi : a.S->n - 1;  // inactive lanes of the last wave shadow env n-1 and never store

    const CollisionLds L = collision_lds_setup(P, a.S->map, tid);
    Pending q;
    double cmd0, cmd1;
    load_env<false>(a, i, active, q, cmd0, cmd1);

    // ---- _env_step (envs/base/env.py:442-461)
    q.old = q.r.p;
    q.drawn = 0;
    q.err = robot_step(P, q.r, cmd0, cmd1, q.z, q.drawn);
    bool hit = false;
    if (!(a.flags & kAblateNoCollision))
        hit = collides_wave(P, a.S->map, a.S->cull, L, a.S->exact_mode, a.S->dense_threshold, a.S->wide != 0, active,
                            slot_of(a.S, i, q), q.r.p.x, q.r.p.y, q.r.p.th);
    if (!active) return;
    finalize_env<false>(a, i, q, hit);
}


// Fast step kernel (kernel 1 of the two-kernel step; needs a distance field).  A pose is cleared in O(1) by the
// outer test; the few envs it cannot clear are finished optimistically ("free") AND parked in `pending`, and kernel 2
// redoes those that really collide.  Waves with many undecided lanes (robots hugging walls) settle them in place.
// Memory operations are grouped so that independent round trips overlap: every wave runs alone on its SIMD, so an
// exposed L2 / HBM latency is pure stall.
template <bool WIDE>
__global__ void __launch_bounds__(kBlock) step_fast_kernel(const StepArgs a)
{
    const DevParams& P = a.S->P;
    const int tid = threadIdx.x;
    const int64_t gi = (int64_t)blockIdx.x * kBlock + tid;
    const bool active = gi < a.S->n;
    const int64_t i = active ? gi : a.S->n - 1;
    DIAG1_STAMP(0);

    // (1) loads for the LDS staging of the scaled footprint and of the shared path (up to 8 doubles per lane per
    //     round), issued first ...
    __attribute__((address_space(3))) double* qv = (__attribute__((address_space(3))) double*)lds_dyn;
    const int nq = 2 * P.n_verts;
    const double my_q = tid < nq ? P.qverts[tid >> 1][tid & 1] : 0.0;
    double t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int k = u * kBlock + tid;
        t[u] = k < a.S->lds_path_doubles ? a.S->path.pts[k] : 0.0;
    }
    // (2) ... then state, action, noise (the first-touch lines of this step): all of it is in flight together
    Pending q;
    double cmd0, cmd1;
    load_env<true>(a, i, active, q, cmd0, cmd1);
    if (gi < kShards) a.pending_next[gi] = 0;  // arm the counters of the NEXT step (the two sets alternate)
    if (gi == 0 && a.inplace_next) *a.inplace_next = 0;
    // (3) LDS writes (the staging loads return first, in issue order)
    if (tid < nq) qv[tid] = my_q;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int k = u * kBlock + tid;
        if (k < a.S->lds_path_doubles) qv[nq + k] = t[u];
    }
    for (int k = 8 * kBlock + tid; k < a.S->lds_path_doubles; k += kBlock) qv[nq + k] = a.S->path.pts[k];  // long paths
    __syncthreads();
    DIAG1_STAMP(1);
    const LdsF64 lds_path = a.S->lds_path_doubles ? (LdsF64)(qv + nq) : (LdsF64) nullptr;

    // ---- _env_step (envs/base/env.py:442-461)
    Robot& r = q.r;
    q.old = r.p;
    q.drawn = 0;
    q.err = robot_step(P, r, cmd0, cmd1, q.z, q.drawn);
    DIAG1_STAMP(2);

    // (3) everything that depends only on the new pose is looked up together: distance-field samples and the
    //     way-point window of the reward
    const int64_t g = slot_of(a.S, i, q);
    double ox = a.S->map.ox, oy = a.S->map.oy;
    if (a.S->map.origins) {
        ox = a.S->map.origins[2 * g + 0];
        oy = a.S->map.origins[2 * g + 1];
    }
    const int px = (int)rint((r.p.x - ox) * a.S->map.inv_res);  // world_to_pixel, coordinate_transformations.py:185-205
    const int py = (int)rint((r.p.y - oy) * a.S->map.inv_res);
    const double c = cos(r.p.th), s = sin(r.p.th);
    const int64_t map_env = a.S->map.shared ? 0 : g;
    OuterLookups look;
    look.off_map = true;
    if (!(a.flags & (kAblateNoCollision | kAblateNoClassify))) look = outer_lookups_issue(a.S->cull, map_env, a.S->map.rows, a.S->map.cols, px, py, c, s);
    const PathWindow win = path_window(P, a.S->path.bbox + (a.S->path.shared ? 0 : g * 8),
                                       a.S->path.index + (a.S->path.shared ? 0 : g * (int64_t)(4 * kPathBuckets)), r.p.x, r.p.y);
    const int cls = active ? outer_lookups_verdict(a.S->cull, look) : kFree;
    DIAG1_STAMP(3);

    bool hit = false;
    const uint64_t amb = __ballot(cls == kAmbiguous);
    const int n_amb = (int)__popcll(amb);
    const int threshold = a.threshold_now ? *a.threshold_now : a.S->dense_threshold;
    if (n_amb > threshold) {
        // many undecided lanes in this wave: settle them in place, one pose at a time by the whole wave
        if (tid == 0 && a.inplace_count) atomicAdd(a.inplace_count, n_amb);
        const bool inner = cls == kAmbiguous && classify_inner_hit(a.S->cull, map_env, px, py, c, s);
        hit = inner;
        uint64_t todo = __ballot(cls == kAmbiguous && !inner);
        const double vqx = tid < P.n_verts ? qv[2 * tid] : 0.0, vqy = tid < P.n_verts ? qv[2 * tid + 1] : 0.0;
        while (todo) {
            const int src = __ffsll((unsigned long long)todo) - 1;
            todo &= todo - 1;
            const int64_t env_ = ((int64_t)bcast_i((int)(g >> 32), src) << 32) | (uint32_t)bcast_i((int)g, src);
            const uint32_t* words = a.S->map.bits + (a.S->map.shared ? 0 : env_ * a.S->map.env_stride);
            const bool h = coop_collides<WIDE>(P, vqx, vqy, bcast_d(c, src), bcast_d(s, src), bcast_i(px, src),
                                               bcast_i(py, src), words, a.S->map.rows, a.S->map.cols, a.S->map.wpr);
            if (tid == src) hit = h;
        }
    } else if (cls == kAmbiguous && !(a.flags & kAblateNoPark)) {
        // a few undecided lanes: park the pre-verdict state for kernel 2 (load-balanced over the whole GPU) and carry
        // on as if the pose were free, which it is for nearly every parked env
        const int shard = (int)(blockIdx.x % kShards);
        const int slot = atomicAdd(a.pending_count + shard, 1);
        q.c = c;
        q.s = s;
        q.px = px;
        q.py = py;
        q.env_lo = (int32_t)(uint32_t)i;
        q.env_hi = (int32_t)(i >> 32);
        a.S->pending[(int64_t)slot * kShards + shard] = q;  // interleaved: the used slots stay in a few pages
    }
    DIAG1_STAMP(4);
    if (!active) return;
    finalize_env<true>(a, i, q, hit, lds_path, &win);
    DIAG1_STAMP(7);
}

// Kernel 2 of a step: kPendingWaves wavefronts per parked env: the lanes rasterise
// the footprint together (coop_collides, wave w takes the row chunks w, w+2, ...); on a collision thread 0 redoes
// the env's finalisation from the parked state.  The first entry is fetched speculatively, together with the
// counter that says whether it exists, so the two round trips overlap.
constexpr int kPendingWaves = 4;  // wave = 2 * (row-chunk slot) + (edge slot)
constexpr int kParkCapacity = 8192;  // undecided poses per step that kernel 2 takes without the waves' help

template <bool WIDE>
__global__ void __launch_bounds__(kBlock * kPendingWaves) step_pending_kernel(const StepArgs a)
{
    DIAG_STAMP(0);
    const DevParams& P = a.S->P;
    const int lane = threadIdx.x % kBlock, wave = threadIdx.x / kBlock;
    const int shard = (int)(blockIdx.x % kShards);
    const int stride = gridDim.x / kShards;
    const Pending* slots = a.S->pending + shard;
    const double vqx = lane < P.n_verts ? P.qverts[lane][0] : 0.0, vqy = lane < P.n_verts ? P.qverts[lane][1] : 0.0;
    const int count = a.pending_count[shard];
    if (blockIdx.x == 0 && a.threshold_next && threadIdx.x < kShards) {
        // Undecided poses of this step, parked + settled in place.  Few of them: kernel 2 absorbs them all in one or
        // two rounds, so the next step parks everything (no wave is held up by its own unlucky lanes).  Many (robots
        // hugging walls everywhere): kernel 2 would need dozens of rounds, the waves settle their own instead.
        int total = a.pending_count[threadIdx.x];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o);
        if (threadIdx.x == 0) *a.threshold_next = total + *a.inplace_count <= kParkCapacity ? 64 : a.S->dense_threshold;
    }
    for (int idx = blockIdx.x / kShards; idx < a.S->pending_cap; idx += stride) {
        const Pending* e = slots + (int64_t)idx * kShards;   // in bounds whatever `count` says
        const double c = e->c, s = e->s;
        const int px = e->px, py = e->py;
        const int64_t i = ((int64_t)e->env_hi << 32) | (uint32_t)e->env_lo;
        const int64_t g = a.S->geom_of_env ? (int64_t)e->geom : i;
        if (idx >= count) break;
        DIAG_STAMP(1);
        const uint32_t* words = a.S->map.bits + (a.S->map.shared ? 0 : g * a.S->map.env_stride);
        // (no inner distance-field test here: nearly every parked pose is free, so the test would cost a dependent
        //  round trip per pose and almost never spare the rasteriser)
        bool hit = false;
        DIAG_STAMP(2);
        if (!(a.flags & kAblateNoCoop))
            hit = coop_collides_quad<WIDE>(P, vqx, vqy, c, s, px, py, words, a.S->map.rows, a.S->map.cols, a.S->map.wpr, wave,
                                           (LdsU32)lds_dyn);
        DIAG_STAMP(3);
        hit = __syncthreads_or(hit);  // wave-uniform verdicts of the block's waves
        DIAG_STAMP(4);
        // kernel 1 already finished this env as "free"; only a collision changes anything
        if (hit && threadIdx.x == 0) {
            Pending q = *e;
            finalize_env<true>(a, i, q, true);
        }
    }
}

__global__ void reset_kernel(DevState st, DevState init, const uint8_t* __restrict__ mask, int64_t n, int tri,
                             int32_t* __restrict__ geom_of_env, const int32_t* __restrict__ next_geom)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (mask && !mask[i]) return;
    int64_t k = i;
    if (geom_of_env) {  // geometry pool: a reset draws the env's next geometry (mini_env.py:469-481)
        k = geom_of_env[i];
        if (next_geom) k = next_geom[k];
        geom_of_env[i] = (int32_t)k;
    }
    st.x[i] = init.x[k];
    st.y[i] = init.y[k];
    st.angle[i] = init.angle[k];
    st.v[i] = init.v[k];
    st.w[i] = init.w[k];
    if (tri) {
        st.steer[i] = init.steer[k];
        st.wheel[i] = init.wheel[k];
    }
    st.min_dist[i] = init.min_dist[k];
    st.target_idx[i] = init.target_idx[k];
    st.cur_iter[i] = init.cur_iter[k];
    st.collided[i] = init.collided[k];
    // delays > 0: the restored State exposes the initial pose / robot state; the queues are empty (pushes restart)
    if (st.pose_seen) {
        st.pose_seen[0 * n + i] = init.x[k];
        st.pose_seen[1 * n + i] = init.y[k];
        st.pose_seen[2 * n + i] = init.angle[k];
    }
    if (st.state_seen) {
        st.state_seen[0 * n + i] = init.x[k];
        st.state_seen[1 * n + i] = init.y[k];
        st.state_seen[2 * n + i] = init.angle[k];
        st.state_seen[3 * n + i] = init.v[k];
        st.state_seen[4 * n + i] = init.w[k];
        st.state_seen[5 * n + i] = tri ? init.steer[k] : 0.0;
        st.state_seen[6 * n + i] = tri ? init.wheel[k] : 0.0;
    }
}

// Monte-Carlo fan-out: env `src`'s complete state copied into every selected env
__global__ void broadcast_state_kernel(DevState st, int32_t* __restrict__ geom_of_env, const uint8_t* __restrict__ mask,
                                       int64_t n, int64_t src, int tri, int control_delay, int pose_delay, int state_delay)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || i == src) return;
    if (mask && !mask[i]) return;
    st.x[i] = st.x[src];
    st.y[i] = st.y[src];
    st.angle[i] = st.angle[src];
    st.v[i] = st.v[src];
    st.w[i] = st.w[src];
    if (tri) {
        st.steer[i] = st.steer[src];
        st.wheel[i] = st.wheel[src];
    }
    st.min_dist[i] = st.min_dist[src];
    st.target_idx[i] = st.target_idx[src];
    st.cur_iter[i] = st.cur_iter[src];
    st.collided[i] = st.collided[src];
    if (geom_of_env) geom_of_env[i] = geom_of_env[src];
    if (st.pose_seen)
        for (int c = 0; c < 3; ++c) st.pose_seen[c * n + i] = st.pose_seen[c * n + src];
    if (st.state_seen)
        for (int c = 0; c < 7; ++c) st.state_seen[c * n + i] = st.state_seen[c * n + src];
    if (st.control_q)
        for (int c = 0; c < 2 * control_delay; ++c) st.control_q[c * n + i] = st.control_q[c * n + src];
    if (st.pose_q)
        for (int c = 0; c < 3 * pose_delay; ++c) st.pose_q[c * n + i] = st.pose_q[c * n + src];
    if (st.state_q)
        for (int c = 0; c < 7 * state_delay; ++c) st.state_q[c * n + i] = st.state_q[c * n + src];
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

__global__ void __launch_bounds__(kBlock) pose_collides_kernel(DevParams P, MapDesc map, CullDesc cull, int exact_mode,
                                                               int dense_threshold, int wide,
                                                               const double* __restrict__ poses, int64_t n, int64_t n_envs,
                                                               const int32_t* __restrict__ geom_of_env,
                                                               uint8_t* __restrict__ out)
{
    const int tid = threadIdx.x;
    const int64_t gi = (int64_t)blockIdx.x * kBlock + tid;
    const bool active = gi < n;
    const int64_t i = active ? gi : n - 1;
    const CollisionLds L = collision_lds_setup(P, map, tid);
    const int64_t env = geom_of_env ? (int64_t)geom_of_env[i % n_envs] : i % n_envs;
    const bool hit = collides_wave(P, map, cull, L, exact_mode, dense_threshold, wide != 0, active, env, poses[3 * i],
                                   poses[3 * i + 1], poses[3 * i + 2]);
    if (active) out[i] = (uint8_t)hit;
}

// get_pixel_footprint: one wave per angle, rasterised by the cooperative path; lane = image row
struct MaskRowSink {
    uint8_t* img;
    int side, hx, hy;
    __device__ __forceinline__ void extent(int, int) {}
    __device__ __forceinline__ bool chunk_matters(int, bool) const { return true; }
    __device__ __forceinline__ bool rows(int y, bool valid, const uint32_t cover[8], int ubase) const
    {
        const int ky = y + hy;
        if (valid && (unsigned)ky < (unsigned)side) {
            for (int b = 0; b < 256; ++b) {
                const int kx = ubase + b + hx;
                if ((cover[b >> 5] >> (b & 31)) & 1u)
                    if ((unsigned)kx < (unsigned)side) img[ky * side + kx] = 255;
            }
        }
        return false;
    }
};

__global__ void __launch_bounds__(kBlock) pixel_footprint_kernel(DevParams P, const double* __restrict__ angles, int64_t n,
                                                                 uint8_t* __restrict__ masks, int side,
                                                                 int32_t* __restrict__ shape_hw)
{
    const int tid = threadIdx.x;
    __attribute__((address_space(3))) double* q = (__attribute__((address_space(3))) double*)lds_dyn;
    for (int k = tid; k < 2 * P.n_verts; k += kBlock) q[k] = P.qverts[k >> 1][k & 1];
    __syncthreads();
    const int64_t i = blockIdx.x;
    const double c = cos(angles[i]), s = sin(angles[i]);
    MaskRowSink sink;
    sink.img = masks + i * (int64_t)side * side;
    sink.side = side;
    footprint_half_sizes(P, c, s, sink.hx, sink.hy);
    if (tid == 0) {
        shape_hw[2 * i] = 2 * sink.hy + 1;
        shape_hw[2 * i + 1] = 2 * sink.hx + 1;
    }
    coop_raster<8, 1>(P, tid < P.n_verts ? q[2 * tid] : 0.0, tid < P.n_verts ? q[2 * tid + 1] : 0.0, c, s, sink);
}

// same image through the per-thread rasteriser (one thread per angle): cross-checks the two exact paths
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

__global__ void __launch_bounds__(kBlock) pixel_footprint_thread_kernel(DevParams P, const double* __restrict__ angles,
                                                                        int64_t n, uint8_t* __restrict__ masks, int side,
                                                                        int32_t* __restrict__ shape_hw)
{
    const int tid = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * kBlock + tid;
    VertLds E;
    E.base = (LdsU32)lds_dyn + tid;
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

// ---- Euclidean distance transform of the lethal cells over the padded map(s) (classify(), bcp_coop.h) ----------
// Distances are only ever compared with thresholds <= `clamp`, so the transform is exact up to `clamp` and
// saturates there.  pass 1: per padded column, vertical distance to the nearest lethal cell of that column.
__global__ void edt_columns_kernel(const uint32_t* __restrict__ bits, int64_t n_maps, int rows, int cols, int wpr, int pad,
                                   int clamp, uint8_t* __restrict__ g)
{
    const int W = cols + 2 * pad, H = rows + 2 * pad;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_maps * W) return;
    const int cp = (int)(t % W);
    const int64_t m = t / W;
    const int c = cp - pad;
    const uint32_t* mb = bits + m * (int64_t)rows * wpr;
    uint8_t* mg = g + m * (int64_t)W * H;
    const bool in_cols = c >= 0 && c < cols;
    int d = clamp;
    for (int rp = 0; rp < H; ++rp) {  // downward sweep
        const int r = rp - pad;
        const bool leth = in_cols && r >= 0 && r < rows && ((mb[r * wpr + (c >> 5)] >> (c & 31)) & 1u);
        d = leth ? 0 : min(d + 1, clamp);
        mg[rp * W + cp] = (uint8_t)d;
    }
    d = clamp;
    for (int rp = H - 1; rp >= 0; --rp) {  // upward sweep
        const int r = rp - pad;
        const bool leth = in_cols && r >= 0 && r < rows && ((mb[r * wpr + (c >> 5)] >> (c & 31)) & 1u);
        d = leth ? 0 : min(d + 1, clamp);
        mg[rp * W + cp] = (uint8_t)min((int)mg[rp * W + cp], d);
    }
}

// pass 2: d^2(r,c) = min over |c - c'| < clamp of (c - c')^2 + g(r,c')^2, stored as floor(min(clamp, d)).
__global__ void edt_rows_kernel(const uint8_t* __restrict__ g, int64_t n_maps, int W, int H, int clamp,
                                uint8_t* __restrict__ out)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_maps * (int64_t)W * H) return;
    const int cp = (int)(idx % W);
    const uint8_t* row = g + (idx - cp);
    int best = clamp * clamp;
    const int lo = max(0, cp - clamp + 1), hi = min(W - 1, cp + clamp - 1);
    for (int k = lo; k <= hi; ++k) {
        const int gv = row[k];
        const int dd = (cp - k) * (cp - k) + gv * gv;
        best = dd < best ? dd : best;
    }
    int sq = (int)sqrt((double)best);
    while (sq * sq > best) --sq;
    while ((sq + 1) * (sq + 1) <= best) ++sq;
    out[idx] = (uint8_t)min(sq, clamp);
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

// ---- egocentric observation ------------------------------------------------------------
// extract_egocentric_costmap (utilities/costmap_utils.py:25-75) = cv2.getRotationMatrix2D + cv2.warpAffine with
// INTER_NEAREST for every env at once.  OpenCV's nearest-neighbour warp works in 22.10 fixed point:
//   X(x, y) = (sat_int((M1*y + M2)*1024) + 512 + sat_int(M0*x*1024)) >> 10      (and likewise Y with M4, M5, M3)
// with M the float64 inverse of the 2x3 transform; a destination pixel copies src[Y][X] or takes the border value.
struct EgoArgs {
    const uint8_t* data;         // raw costmaps: [rows][cols] shared or one per map entry
    int64_t map_stride;          // bytes per map entry (0 when shared)
    const int32_t* valid_rows;   // per-entry true shape (optional)
    const int32_t* valid_cols;
    int32_t rows, cols;          // allocation shape of one map
    const double* origins;       // per-entry origins or nullptr
    double ox, oy, res, inv_res;
    const double* poses;         // [n,3] or nullptr: the bound state
    const double *sx, *sy, *sth;
    const int32_t* geom_of_env;
    int32_t shared;
    int32_t has_window;
    double win_ox, win_oy;
    int32_t drows, dcols;        // output shape
    uint32_t cols_magic;         // floor(2^32 / cols) + 1: idx / cols == umulhi(idx, magic) for idx * cols < 2^32
    int32_t stage_map;           // shared map is copied to LDS (rows * cols bytes)
    int32_t border;
    int64_t n_envs;              // image i shows the costmap of env i % n_envs
    int64_t n_images;
    uint8_t* out;                // [n][drows][dcols]
};

__device__ __forceinline__ int sat_int(double v)   // cv::saturate_cast<int>(double): nearest-even, saturating
{
    const double r = rint(v);
    return r >= 2147483647.0 ? 2147483647 : (r <= -2147483648.0 ? (-2147483647 - 1) : (int)r);
}

// clamp(v, lo, hi) as ONE instruction (the compiler emits v_max + v_min for min(max()))
__device__ __forceinline__ int clamp_med3(int v, int lo, int hi)
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(lo), "v"(hi));
    return r;
}

// byte at a raw LDS address (no symbol base is added: the caller folds the base into the address)
__device__ __forceinline__ uint32_t lds_byte_at(uint32_t addr)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t) * (__attribute__((address_space(3))) const uint8_t*)addr;
#else
    (void)addr;
    return 0;
#endif
}

typedef uint64_t __attribute__((aligned(1))) u64_unaligned;
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;

// ---- building blocks ------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) int* LdsI32;
typedef __attribute__((address_space(3))) uint8_t* LdsU8;
constexpr int kRowOff = -2147483647 - 1;   // row-table X0 of a row that lies off the map (real X0 are >= INT_MIN + 512)

struct EgoXform {
    double M[6];   // dst -> src, as cv::warpAffine uses it
    int vrows, vcols;
    int g_lo, g_hi;   // map entry
};

// cv2.getRotationMatrix2D(world_to_pixel(pose), 180*theta/pi, 1), the window shift, and warpAffine's inversion
__device__ __forceinline__ EgoXform ego_transform(const EgoArgs& a, int64_t img)
{
    EgoXform T;
    const int64_t me = img % a.n_envs;
    const int64_t g = a.shared ? 0 : (a.geom_of_env ? (int64_t)a.geom_of_env[me] : me);
    double ox = a.ox, oy = a.oy;
    if (a.origins) {
        ox = a.origins[2 * g];
        oy = a.origins[2 * g + 1];
    }
    double px, py, th;
    if (a.poses) {
        px = a.poses[3 * img];
        py = a.poses[3 * img + 1];
        th = a.poses[3 * img + 2];
    } else {
        px = a.sx[img];
        py = a.sy[img];
        th = a.sth[img];
    }
    double* M = T.M;
    const float cx = (float)rint((px - ox) * a.inv_res), cy = (float)rint((py - oy) * a.inv_res);   // Point2f centre
    const double angle = (180 * th / M_PI) * (M_PI / 180);
    const double alpha = cos(angle), beta = sin(angle);
    M[0] = alpha;
    M[1] = beta;
    M[2] = (1 - alpha) * cx - beta * cy;
    M[3] = -beta;
    M[4] = alpha;
    M[5] = beta * cx + (1 - alpha) * cy;
    if (a.has_window) {
        // shift so that the window origin lands on output pixel (0, 0); composed in float32 (costmap_utils.py:50-64)
        const double dsx = rint((a.win_ox - (ox - px)) * a.inv_res), dsy = rint((a.win_oy - (oy - py)) * a.inv_res);
        float t[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) t[k] = (float)M[k];
        t[2] = t[2] + (-(float)dsx);
        t[5] = t[5] + (-(float)dsy);
#pragma unroll
        for (int k = 0; k < 6; ++k) M[k] = (double)t[k];
    }
    {   // cv::warpAffine inverts the transform in float64
        double D = M[0] * M[4] - M[1] * M[3];
        D = D != 0 ? 1. / D : 0;
        const double A11 = M[4] * D, A22 = M[0] * D;
        M[0] = A11;
        M[1] *= -D;
        M[3] *= -D;
        M[4] = A22;
        const double b1 = -M[0] * M[2] - M[1] * M[5];
        const double b2 = -M[3] * M[2] - M[4] * M[5];
        M[2] = b1;
        M[5] = b2;
    }
    T.vrows = a.valid_rows ? a.valid_rows[g] : a.rows;
    T.vcols = a.valid_cols ? a.valid_cols[g] : a.cols;
    T.g_lo = (int)(uint32_t)g;
    T.g_hi = (int)(g >> 32);
    return T;
}

// lane k's transform, broadcast to the whole wave (scalar registers)
struct EgoImage {
    double m0, m1, m2, m3, m4, m5;
    int vr, vc;
    int64_t g;
};

__device__ __forceinline__ EgoImage ego_broadcast(const EgoXform& T, int k)
{
    EgoImage I;
    I.m0 = bcast_d(T.M[0], k);
    I.m1 = bcast_d(T.M[1], k);
    I.m2 = bcast_d(T.M[2], k);
    I.m3 = bcast_d(T.M[3], k);
    I.m4 = bcast_d(T.M[4], k);
    I.m5 = bcast_d(T.M[5], k);
    I.vr = bcast_i(T.vrows, k);
    I.vc = bcast_i(T.vcols, k);
    I.g = ((int64_t)bcast_i(T.g_hi, k) << 32) | (uint32_t)bcast_i(T.g_lo, k);
    return I;
}

// LDS copy of one costmap with a one-cell ring of the border value.  Whole workgroup; ends with a barrier.
// The map is fetched as aligned dwords, eight independent loads in flight per thread (a cold map costs a few memory
// round trips instead of one per row), and scattered into the ringed layout byte by byte.
__device__ __forceinline__ void ego_stage_map(const EgoArgs& a, const uint8_t* __restrict__ src, int vr, int vc, LdsU8 lmap,
                                              int pitch, int map_bytes)
{
    __attribute__((address_space(3))) uint32_t* l32 = (__attribute__((address_space(3))) uint32_t*)lmap;
    for (int k = threadIdx.x; k < map_bytes / 4; k += 256) l32[k] = (uint32_t)a.border * 0x01010101u;
    __syncthreads();
    const int total = a.rows * a.cols;
    const int off = (int)((uintptr_t)src & 3);   // the map entry need not start on a dword boundary
    const uint32_t* __restrict__ w32 = reinterpret_cast<const uint32_t*>(src - off);
    const int n_words = (off + total + 3) >> 2;
    for (int w0 = threadIdx.x; w0 < n_words; w0 += 8 * 256) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int w = w0 + u * 256;
            v[u] = w < n_words ? w32[w] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int w = w0 + u * 256;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int idx = 4 * w + j - off;                       // linear index into the map entry
                if (idx >= 0 && idx < total) {
                    const int r = (int)__umulhi((uint32_t)idx, a.cols_magic), c = idx - r * a.cols;
                    if (r < vr && c < vc) lmap[(r + 1) * pitch + 1 + c] = (uint8_t)(v[u] >> (8 * j));
                }
            }
        }
    }
    __syncthreads();
}

// per-row terms of one image (rounding term included; staged sampling: ring offset and LDS base folded in) and the
// off-map flag of each row, for rows t0, t0 + tstep, ...
template <bool STAGED>
__device__ __forceinline__ void ego_row_terms(const EgoArgs& a, const EgoImage& I, int x_shift, LdsI32 row_tab, int t0,
                                              int tstep)
{
    const int last_cx = sat_int(I.m0 * (a.dcols - 1) * 1024), last_cy = sat_int(I.m3 * (a.dcols - 1) * 1024);
    for (int y = t0; y < a.drows; y += tstep) {
        const int rx = sat_int((I.m1 * y + I.m2) * 1024) + 512, ry = sat_int((I.m4 * y + I.m5) * 1024) + 512;
        const int xa = rx >> 10, xb = (rx + last_cx) >> 10, ya = ry >> 10, yb = (ry + last_cy) >> 10;
        const bool off = (xa < 0 && xb < 0) || (xa >= I.vc && xb >= I.vc) || (ya < 0 && yb < 0) || (ya >= I.vr && yb >= I.vr);
        row_tab[2 * y] = off ? kRowOff : (STAGED ? rx + x_shift * 1024 : rx);
        row_tab[2 * y + 1] = STAGED ? ry + 1024 : ry;
    }
}

// the pixels of rows r0, r0 + rstep, ... for this lane's column groups.  Pixel groups are PX columns wide; the last
// group of a row is shifted left so that it ends at the last column (it recomputes a few pixels of its neighbour
// instead of needing a narrower store).
template <bool STAGED, int PX>
__device__ __forceinline__ void ego_pixels(const EgoArgs& a, const EgoImage& I, int x_shift, int pitch,
                                           const uint8_t* __restrict__ src, LdsI32 row_tab, uint8_t* __restrict__ image,
                                           int cg, int r0, int rstep)
{
    const uint32_t border = (uint32_t)a.border;
    const uint64_t border8 = (uint64_t)border * 0x0101010101010101ull;
    const int x_lo = x_shift - 1, x_hi = x_shift + I.vc, y_lo = 0, y_hi = I.vr + 1;   // ring coordinates (staged sampling)
    for (int xg = PX * cg; xg < a.dcols; xg += 128) {
        const int x0 = min(xg, a.dcols - PX);
        int ccx[PX], ccy[PX];
#pragma unroll
        for (int j = 0; j < PX; ++j) {   // cv::hal::warpAffine's adelta / bdelta for this lane's columns
            ccx[j] = sat_int(I.m0 * (x0 + j) * 1024);
            ccy[j] = sat_int(I.m3 * (x0 + j) * 1024);
        }
        for (int y = r0; y < a.drows; y += rstep) {
            const int rx = row_tab[2 * y], ry = row_tab[2 * y + 1];
            uint64_t packed = border8;
            if (rx != kRowOff) {
                uint32_t half[2] = {0, 0};
#pragma unroll
                for (int j = 0; j < PX; ++j) {
                    // (saturate_cast<short> never bites: |X|, |Y| < 2^21 and maps are < 2^15)
                    const int X = (rx + ccx[j]) >> 10, Y = (ry + ccy[j]) >> 10;
                    uint32_t val;
                    if (STAGED) {   // clamp onto the border ring of the LDS copy: every address is valid
                        const int xc = clamp_med3(X, x_lo, x_hi), yc = clamp_med3(Y, y_lo, y_hi);
                        val = lds_byte_at((uint32_t)(__mul24(yc, pitch) + xc));
                    } else {        // global gather: only the in-map lanes issue a load
                        val = border;
                        if ((unsigned)X < (unsigned)I.vc && (unsigned)Y < (unsigned)I.vr)
                            val = (uint32_t)src[(uint32_t)__mul24(Y, a.cols) + (uint32_t)X];
                    }
                    half[j >> 2] |= val << (8 * (j & 3));
                }
                packed = ((uint64_t)half[1] << 32) | half[0];
            }
            uint8_t* const p = image + (int64_t)y * a.dcols + x0;
            if (PX == 8)
                *reinterpret_cast<u64_unaligned*>(p) = packed;
            else
                *reinterpret_cast<u32_unaligned*>(p) = (uint32_t)packed;
        }
    }
}

__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One WAVEFRONT per image, persistent workgroups of 4 waves that stage a shared costmap in LDS once and then walk
// over images.
//   * transforms: lane l of a wave prepares the (inverted) warp matrix of the wave's l-th image, so the float64
//     sin / cos / inversion work is done once per image by one lane; the wave then takes the images one by one and
//     broadcasts that lane's matrix (v_readlane -> scalar registers).
//   * pixels: lane = (row mod 4, group of 8 consecutive columns).  The column terms of a lane's 8 pixels stay in
//     registers, the per-row terms come from a small per-wave LDS table, and the 8 pixels leave as one 64-bit store
//     (image rows are dcols bytes apart, so these stores are generally unaligned).
//   * rows whose source segment lies entirely off the map are filled with the border value without sampling: the
//     source coordinates are monotone in x, so it is enough to look at the row's two ends.
//   * shared map: the LDS copy carries a one-cell ring of the border value and the source coordinates are clamped
//     onto it (v_med3), so a pixel is add, add, shift, shift, clamp, clamp, multiply-add, LDS byte read, pack --
//     no bounds compare and no select.  The ring offset and the LDS base address ride in the per-row terms.
// LDS: [shared map + ring, dword padded] [4 waves x drows x {X0 (kRowOff = row is off the map), Y0}]
// STAGED = false: maps that do not fit LDS are sampled straight from global memory.
template <bool STAGED, int PX>
__global__ void __launch_bounds__(256) ego_costmap_kernel(const EgoArgs a)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pitch = a.cols + 2;
    const int map_bytes = STAGED ? (((a.rows + 2) * pitch + 3) & ~3) : 0;
    const LdsU8 lmap = (LdsU8)lds_dyn;
    const LdsI32 row_tab = (LdsI32)(lmap + map_bytes) + wave * (2 * a.drows);
    if (STAGED)
        ego_stage_map(a, a.data, a.valid_rows ? a.valid_rows[0] : a.rows, a.valid_cols ? a.valid_cols[0] : a.cols, lmap,
                      pitch, map_bytes);
    // staged sampling: X' = X + x_shift and Y' = Y + 1 index the ringed copy directly (raw LDS byte address)
    const int x_shift = 1 + (int)(uint32_t)(uintptr_t)lmap;
    constexpr int kGroups = 128 / PX, kRows = 64 / kGroups;   // lanes of a wave: kRows image rows x kGroups pixel groups
    const int cg = lane % kGroups, rl = lane / kGroups;
    const int64_t P = (int64_t)a.drows * a.dcols;
    const int64_t first = (int64_t)blockIdx.x * 4 + wave, stride = (int64_t)gridDim.x * 4;
    for (int64_t base = first; base < a.n_images; base += 64 * stride) {
        EgoXform T;
        memset(&T, 0, sizeof(T));
        const int64_t mine = base + lane * stride;   // lane l: transform of the wave's l-th image of this batch
        if (mine < a.n_images) T = ego_transform(a, mine);
        const int64_t left = (a.n_images - base + stride - 1) / stride;
        const int count = (int)(left < 64 ? left : 64);
        for (int k = 0; k < count; ++k) {            // the wave's images, one at a time
            const int64_t img = base + k * stride;
            const EgoImage I = ego_broadcast(T, k);
            ego_row_terms<STAGED>(a, I, x_shift, row_tab, lane, 64);
            wave_lds_sync();
            ego_pixels<STAGED, PX>(a, I, x_shift, pitch, a.data + I.g * a.map_stride, row_tab, a.out + img * P, cg, rl, kRows);
            wave_lds_sync();   // the table is rewritten for the next image
        }
    }
}

// Private / pooled costmaps that fit LDS: images are first grouped by map entry (ego_bin_* kernels below); every
// workgroup then takes an equal slice of that grouped list and walks through it run by run (a run = consecutive
// images of one map entry): stage the entry (with the border ring), produce the run's images four at a time, one per
// wavefront exactly like the shared-map kernel, and let the 4 waves share each of the up to three left-over images --
// wave w takes every 4th slice of rows -- so that nobody idles (private maps: every run is a single image).
// LDS: [map + ring] [4 row tables].
template <int PX>
__global__ void __launch_bounds__(256) ego_costmap_binned_kernel(const EgoArgs a, const int32_t* __restrict__ bin_start,
                                                                 const int32_t* __restrict__ bin_count,
                                                                 const int32_t* __restrict__ order)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pitch = a.cols + 2;
    const int map_bytes = ((a.rows + 2) * pitch + 3) & ~3;
    const LdsU8 lmap = (LdsU8)lds_dyn;
    const LdsI32 tables = (LdsI32)(lmap + map_bytes);
    const LdsI32 wave_tab = tables + wave * (2 * a.drows);
    const int x_shift = 1 + (int)(uint32_t)(uintptr_t)lmap;
    constexpr int kGroups = 128 / PX, kRows = 64 / kGroups;
    const int cg = lane % kGroups, rl = lane / kGroups;
    const int64_t P = (int64_t)a.drows * a.dcols;
    const int64_t chunk = (a.n_images + gridDim.x - 1) / gridDim.x;
    const int64_t lo = blockIdx.x * chunk, hi = min(lo + chunk, a.n_images);
    for (int64_t pos = lo; pos < hi;) {
        const int64_t me = (int64_t)order[pos] % a.n_envs;
        const int64_t g = a.geom_of_env ? (int64_t)a.geom_of_env[me] : me;
        const int64_t run_end = min(hi, (int64_t)bin_start[g] + bin_count[g]);
        const int run = (int)(run_end - pos);   // (uniform over the workgroup)
        __syncthreads();                        // everyone is done with the previous map
        ego_stage_map(a, a.data + g * a.map_stride, a.valid_rows ? a.valid_rows[g] : a.rows,
                      a.valid_cols ? a.valid_cols[g] : a.cols, lmap, pitch, map_bytes);
        const int whole = run & ~3;
        // ---- one image per wavefront: wave w takes the run's images w, w + 4, ...; lane l prepares the l-th of them
        for (int base = wave; base < whole; base += 256) {
            EgoXform T;
            memset(&T, 0, sizeof(T));
            int my_img = 0;
            if (base + 4 * lane < whole) {
                my_img = order[pos + base + 4 * lane];
                T = ego_transform(a, my_img);
            }
            const int batch = min(64, (whole - base + 3) / 4);
            for (int k = 0; k < batch; ++k) {
                const int64_t img = (uint32_t)bcast_i(my_img, k);
                const EgoImage I = ego_broadcast(T, k);
                ego_row_terms<true>(a, I, x_shift, wave_tab, lane, 64);
                wave_lds_sync();
                ego_pixels<true, PX>(a, I, x_shift, pitch, nullptr, wave_tab, a.out + img * P, cg, rl, kRows);
                wave_lds_sync();
            }
        }
        // ---- the left-over images: the four waves share each of them (one row table, two barriers per image)
        if (run > whole) {
            int my_img = 0;
            EgoXform T;
            memset(&T, 0, sizeof(T));
            if (whole + lane < run) {
                my_img = order[pos + whole + lane];
                T = ego_transform(a, my_img);
            }
            for (int k = 0; k < run - whole; ++k) {
                const int64_t img = (uint32_t)bcast_i(my_img, k);
                const EgoImage I = ego_broadcast(T, k);
                __syncthreads();   // the table is free (earlier images are finished)
                ego_row_terms<true>(a, I, x_shift, tables, threadIdx.x, 256);
                __syncthreads();
                ego_pixels<true, PX>(a, I, x_shift, pitch, nullptr, tables, a.out + img * P, cg, wave * kRows + rl, 4 * kRows);
            }
        }
        pos = run_end;
    }
}

// ---- grouping images by map entry: count -> exclusive scan -> scatter ------------------------------------------
__global__ void ego_bin_count_kernel(const int32_t* __restrict__ geom_of_env, int64_t n_envs, int64_t n_images,
                                     int32_t* __restrict__ bin_count, int32_t* __restrict__ rank)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_images) return;
    const int64_t me = i % n_envs;
    const int64_t g = geom_of_env ? (int64_t)geom_of_env[me] : me;
    rank[i] = atomicAdd(bin_count + g, 1);
}

__global__ void __launch_bounds__(1024) ego_bin_scan_kernel(const int32_t* __restrict__ bin_count, int64_t n_bins,
                                                            int32_t* __restrict__ bin_start)
{
    __shared__ int32_t part[1024];
    __shared__ int32_t carry;
    const int tid = threadIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < n_bins; base += 1024) {
        const int64_t i = base + tid;
        const int32_t v = i < n_bins ? bin_count[i] : 0;
        part[tid] = v;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {   // Hillis-Steele inclusive scan
            const int32_t t = tid >= d ? part[tid - d] : 0;
            __syncthreads();
            part[tid] += t;
            __syncthreads();
        }
        if (i < n_bins) bin_start[i] = carry + part[tid] - v;
        __syncthreads();
        if (tid == 1023) carry += part[1023];
        __syncthreads();
    }
}

__global__ void ego_bin_scatter_kernel(const int32_t* __restrict__ geom_of_env, int64_t n_envs, int64_t n_images,
                                       const int32_t* __restrict__ bin_start, const int32_t* __restrict__ rank,
                                       int32_t* __restrict__ order)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_images) return;
    const int64_t me = i % n_envs;
    const int64_t g = geom_of_env ? (int64_t)geom_of_env[me] : me;
    order[bin_start[g] + rank[i]] = (int32_t)i;
}

// EgocentricCostmap.observation's goal_n_state (envs/egocentric.py:140-160), one thread per env
__global__ void goal_n_state_kernel(const StepStatic* __restrict__ S, double wsx, double wsy, int n_state,
                                    float* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S->n) return;
    const int64_t g = S->geom_of_env ? (int64_t)S->geom_of_env[i] : i;
    const int m = S->path.shared ? S->path.max_len : S->path.lens[g];
    // Observation.path: the way points still ahead, path[target_idx:] (reward.py:59-64) -- or, for the pure-pursuit
    // provider, path[:target_idx + 1] (reward.py:118-123), whose first row is always way point 0
    const int target = S->P.reward_provider == BCP_REWARD_PURE_PURSUIT ? 0 : S->st.target_idx[i];
    float* o = out + i * (3 + n_state);
    if (target > m - 1) {   // nothing left of the path: zeros (egocentric.py:142-150)
        for (int k = 0; k < 3 + n_state; ++k) o[k] = 0.0f;
        return;
    }
    const double* wp = S->path.pts + ((S->path.shared ? 0 : g * (int64_t)S->path.max_len) + target) * 5;
    const int64_t n = S->n;
    // Observation.pose / .robot_state are the delayed ones when delays are configured
    const bool dp = S->P.pose_delay > 0, ds = S->P.state_delay > 0;
    const double x = dp ? S->st.pose_seen[i] : S->st.x[i], y = dp ? S->st.pose_seen[n + i] : S->st.y[i];
    const double th = dp ? S->st.pose_seen[2 * n + i] : S->st.angle[i];
    // inverse_transform (coordinate_transformations.py:57-84), then project_poses (:310-328)
    const double c = cos(th), s = sin(th);
    const double tx = -x * c - y * s, ty = x * s - y * c, tt = normalize_angle(-th);
    const double ct = cos(tt), st = sin(tt);
    const double ex = ct * wp[0] + (-st) * wp[1] + tx;
    const double ey = st * wp[0] + ct * wp[1] + ty;
    const double eth = normalize_angle(wp[2] + tt);
    o[0] = (float)fmin(fmax(ex / wsx, -1.0), 1.0);
    o[1] = (float)fmin(fmax(ey / wsy, -1.0), 1.0);
    o[2] = (float)eth;
    // robot_state.to_numpy_array(): x, y, angle, v, w (, wheel_angle)
    o[3] = (float)(ds ? S->st.state_seen[i] : S->st.x[i]);
    o[4] = (float)(ds ? S->st.state_seen[n + i] : S->st.y[i]);
    o[5] = (float)(ds ? S->st.state_seen[2 * n + i] : S->st.angle[i]);
    o[6] = (float)(ds ? S->st.state_seen[3 * n + i] : S->st.v[i]);
    o[7] = (float)(ds ? S->st.state_seen[4 * n + i] : S->st.w[i]);
    if (n_state > 5) o[8] = (float)(ds ? S->st.state_seen[6 * n + i] : S->st.wheel[i]);
}

// ------------------------------------------------------------------------------------------------ host API
// ---- sample points of the distance-field classification (see bcp_coop.h) -----------------------------------
static double seg_dist(double px, double py, double ax, double ay, double bx, double by)
{
    const double vx = bx - ax, vy = by - ay, wx = px - ax, wy = py - ay;
    const double vv = vx * vx + vy * vy;
    double t = vv > 0 ? (wx * vx + wy * vy) / vv : 0.0;
    t = t < 0 ? 0 : (t > 1 ? 1 : t);
    const double cx = ax + t * vx, cy = ay + t * vy;
    return std::sqrt((px - cx) * (px - cx) + (py - cy) * (py - cy));
}

static bool point_in_polygon(double px, double py, const double (*v)[2], int k)
{
    bool in = false;
    for (int i = 0, j = k - 1; i < k; j = i++) {
        if (((v[i][1] > py) != (v[j][1] > py)) &&
            (px < (v[j][0] - v[i][0]) * (py - v[i][1]) / (v[j][1] - v[i][1]) + v[i][0]))
            in = !in;
    }
    return in;
}

// Worst-case slack, in pixels, between the real rotated footprint and the pixel set cv2.fillPoly produces from it:
// vertex rounding moves the contour by <= sqrt(.5), Bresenham strays <= .5 from the rounded contour, 16.16 slopes
// add < .01; a sample centre is itself rounded to a pixel (<= sqrt(.5)).
static const double kSlackOuter = 0.7072 + 0.5 + 0.01 + 0.7072;
static const double kSlackInner = 0.7072 + 0.7072 + 0.05;

static void build_cull_geometry(const bcp_params& p, double res, CullDesc* C)
{
    const int K = p.n_verts;
    double xmin = 1e300, xmax = -1e300, ymin = 1e300, ymax = -1e300, rmax = 0;
    for (int k = 0; k < K; ++k) {
        xmin = std::min(xmin, p.verts[k][0]);
        xmax = std::max(xmax, p.verts[k][0]);
        ymin = std::min(ymin, p.verts[k][1]);
        ymax = std::max(ymax, p.verts[k][1]);
        rmax = std::max(rmax, std::sqrt(p.verts[k][0] * p.verts[k][0] + p.verts[k][1] * p.verts[k][1]));
    }
    C->reach = (int)std::ceil(rmax / res) + 2;
    C->pad = 2 * C->reach + 4;
    const double ay = 0.5 * (ymin + ymax), half_w = 0.5 * (ymax - ymin);
    // axis segment: pulled in from the ends by a quarter of the half width, so that the round caps of the capsule
    // still cover the corners of a box-like footprint without inflating the radius (corner distance hypot(w/4, w))
    double a0 = xmin + 0.25 * half_w, a1 = xmax - 0.25 * half_w;
    if (a0 > a1) a0 = a1 = 0.5 * (xmin + xmax);
    // OUTER: capsule around the axis segment [a0,a1] x {ay} that contains every vertex (hence the polygon), covered
    // by n_out discs: a disc row of spacing h covers the capsule of radius rho when its radius is sqrt(rho^2+(h/2)^2)
    double rho = 0;
    for (int k = 0; k < K; ++k) rho = std::max(rho, seg_dist(p.verts[k][0], p.verts[k][1], a0, ay, a1, ay));
    const int n_out = a1 > a0 ? std::min(kMaxSamples, std::max(2, (int)std::ceil((a1 - a0) / (0.5 * rho)) + 1)) : 1;
    const double h = n_out > 1 ? (a1 - a0) / (n_out - 1) : 0.0;
    const double r_out = std::sqrt(rho * rho + 0.25 * h * h) / res + kSlackOuter;
    C->n_out = n_out;
    for (int i = 0; i < n_out; ++i) C->out_x[i] = (a0 + i * h) / res;
    C->t_out = (int)std::floor(r_out) + 1;  // floor(d) >= t_out  =>  d > r_out
    // INNER: discs centred on the same axis that lie inside the polygon
    C->n_in = 0;
    for (int j = 0; j < kMaxSamples; ++j) {
        const double bx = kMaxSamples > 1 ? a0 + (a1 - a0) * j / (kMaxSamples - 1) : a0;
        if (!point_in_polygon(bx, ay, p.verts, K)) continue;
        double rin = 1e300;
        for (int k = 0; k < K; ++k) {
            const int kn = (k + 1) % K;
            rin = std::min(rin, seg_dist(bx, ay, p.verts[k][0], p.verts[k][1], p.verts[kn][0], p.verts[kn][1]));
        }
        const double r = rin / res - kSlackInner;   // lethal cell within r of the sample pixel => inside the mask
        const int t = (int)std::floor(r) - 1;       // floor(d) <= t  =>  d < t + 1 <= r
        if (t < 0) continue;
        C->in_x[C->n_in] = bx / res;
        C->t_in[C->n_in] = t;
        ++C->n_in;
        if (a1 <= a0) break;
    }
    C->axis_y = ay / res;
}

static int footprint_is_wide(const bcp_params& p, double res)
{
    double d2 = 0;
    for (int i = 0; i < p.n_verts; ++i)
        for (int j = 0; j < i; ++j) {
            const double dx = p.verts[i][0] - p.verts[j][0], dy = p.verts[i][1] - p.verts[j][1];
            d2 = std::max(d2, dx * dx + dy * dy);
        }
    return std::sqrt(d2) / res + 3.0 > 96.0;  // row masks of the cooperative path: 3 words unless wider
}

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
    d.sp2_lo = p.spatial_precision * p.spatial_precision * (1.0 - 1e-13);
    d.sp2_hi = p.spatial_precision * p.spatial_precision * (1.0 + 1e-13);
    d.reward_provider = p.reward_provider;
    d.control_delay = p.control_delay;
    d.pose_delay = p.pose_delay;
    d.state_delay = p.state_delay;
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
    if (params->reward_provider != BCP_REWARD_CONTINUOUS && params->reward_provider != BCP_REWARD_PURE_PURSUIT)
        return fail(BCP_E_INVALID, "bcp_create: unknown reward provider %d", params->reward_provider);
    if (params->control_delay < 0 || params->pose_delay < 0 || params->state_delay < 0)
        return fail(BCP_E_INVALID, "bcp_create: delays must be >= 0");
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
    h->exact_mode = 0;
    h->dense_threshold = 6;
    h->adaptive = 1;
    h->cull_enabled = 1;
    h->defer = 1;
    h->static_dirty = true;
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
    if (h->path_bbox) (void)hipFree(h->path_bbox);
    if (h->path_index) (void)hipFree(h->path_index);
    if (h->edt) (void)hipFree(h->edt);
    if (h->edt_col) (void)hipFree(h->edt_col);
    if (h->pending) (void)hipFree(h->pending);
    if (h->pending_count) (void)hipFree(h->pending_count);
    if (h->adapt) (void)hipFree(h->adapt);
    if (h->dev_static) (void)hipFree(h->dev_static);
    if (h->ego_bins) (void)hipFree(h->ego_bins);
    if (h->ego_order) (void)hipFree(h->ego_order);
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

extern "C" int bcp_set_geometry_pool(bcp_handle* h, int32_t n_geoms, int32_t* geom_of_env, const int32_t* next_geom)
{
    if (!h) return fail(BCP_E_INVALID, "bcp_set_geometry_pool: null handle");
    if (n_geoms < 0 || (n_geoms > 0 && !geom_of_env))
        return fail(BCP_E_INVALID, "bcp_set_geometry_pool: n_geoms > 0 needs geom_of_env");
    if ((n_geoms > 0) != (h->n_geoms > 0) || (n_geoms > 0 && n_geoms != h->n_geoms)) {
        // the non-shared arrays change their entry count: they have to be given again
        h->have_map = h->have_path = h->have_init = false;
    }
    h->n_geoms = n_geoms;
    h->geom_of_env = n_geoms > 0 ? geom_of_env : nullptr;
    h->next_geom = n_geoms > 0 ? next_geom : nullptr;
    h->static_dirty = true;
    return BCP_OK;
}

extern "C" int bcp_set_tuning(bcp_handle* h, int32_t key, int32_t value)
{
    if (!h) return fail(BCP_E_INVALID, "bcp_set_tuning: null handle");
    h->static_dirty = true;
    switch (key) {
        case BCP_TUNE_EXACT_MODE:
            if (value < 0 || value > 2) return fail(BCP_E_INVALID, "bcp_set_tuning: exact mode must be 0, 1 or 2");
            h->exact_mode = value;
            return BCP_OK;
        case BCP_TUNE_DENSE_THRESHOLD:
            h->dense_threshold = value;
            h->adaptive = 0;   // an explicit threshold is taken as is
            return BCP_OK;
        case BCP_TUNE_DEFER:
            h->defer = value ? 1 : 0;
            return BCP_OK;
        case BCP_TUNE_CULL:
            h->cull_enabled = value ? 1 : 0;
            h->cull.on = (value && h->cull.edt) ? 1 : 0;
            return BCP_OK;
        default:
            return fail(BCP_E_INVALID, "bcp_set_tuning: unknown key %d", key);
    }
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
    const int64_t n_maps = shared ? 1 : n_slots(h);
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
    h->map_data = data;
    h->map_valid_rows = valid_rows;
    h->map_valid_cols = valid_cols;
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
    // stage the shared bitmap in LDS when the whole collision scratch then stays within 64 KiB per workgroup
    m.in_lds = (shared && collision_lds_bytes(h->params.n_verts, 1, rows, wpr) <= 64 * 1024) ? 1 : 0;
    h->wide = footprint_is_wide(h->params, resolution);
    // distance field for the O(1) pre-classification (shared maps)
    CullDesc& C = h->cull;
    memset(&C, 0, sizeof(C));
    build_cull_geometry(h->params, resolution, &C);
    if (h->cull_enabled) {
        // shared map: padding wide enough that every sample of a pose whose image touches the map is stored;
        // private maps: just enough that a sample outside the stored rectangle (more than `pad` px away from every
        // cell of the map) is known to clear the outer test
        if (!shared) C.pad = std::max(8, C.t_out);
        const int clamp = std::min(255, std::max(C.t_out + 1, 2));
        const int W = cols + 2 * C.pad, H = rows + 2 * C.pad;
        const size_t cells = (size_t)n_maps * W * H;
        if (cells > h->edt_bytes) {
            if (h->edt) HIP_TRY(hipFree(h->edt));
            h->edt = nullptr;
            h->edt_bytes = 0;
            HIP_TRY(hipMalloc((void**)&h->edt, cells));
            h->edt_bytes = cells;
        }
        if (cells > h->edt_col_bytes) {
            if (h->edt_col) HIP_TRY(hipFree(h->edt_col));
            h->edt_col = nullptr;
            h->edt_col_bytes = 0;
            HIP_TRY(hipMalloc((void**)&h->edt_col, cells));
            h->edt_col_bytes = cells;
        }
        const int64_t n_cols_total = n_maps * W;
        hipLaunchKernelGGL(edt_columns_kernel, dim3((unsigned)((n_cols_total + 63) / 64)), dim3(64), 0, s, h->bitmap, n_maps,
                           rows, cols, wpr, C.pad, clamp, h->edt_col);
        hipLaunchKernelGGL(edt_rows_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, s, h->edt_col, n_maps, W, H,
                           clamp, h->edt);
        HIP_TRY(hipGetLastError());
        C.edt = h->edt;
        C.width = W;
        C.height = H;
        C.env_stride = shared ? 0 : (int64_t)W * H;
        C.on = C.t_out <= clamp ? 1 : 0;
        if (!h->pending) {
            const int64_t blocks = (h->n + kBlock - 1) / kBlock;
            h->pending_cap = (int32_t)(((blocks + kShards - 1) / kShards) * kBlock);  // every env of a shard's blocks
            HIP_TRY(hipMalloc(&h->pending, (size_t)kShards * h->pending_cap * sizeof(Pending)));
            HIP_TRY(hipMalloc((void**)&h->pending_count, 2 * kShards * sizeof(int32_t)));
            HIP_TRY(hipMemsetAsync(h->pending_count, 0, 2 * kShards * sizeof(int32_t), s));
            HIP_TRY(hipMalloc((void**)&h->adapt, 4 * sizeof(int32_t)));
            const int32_t init[4] = {h->dense_threshold, h->dense_threshold, 0, 0};
            HIP_TRY(hipMemcpyAsync(h->adapt, init, sizeof(init), hipMemcpyHostToDevice, s));
            HIP_TRY(hipStreamSynchronize(s));   // (`init` is on the stack)
        }
    }
    h->have_map = true;
    h->static_dirty = true;
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
    const int64_t total = (shared ? 1 : n_slots(h)) * (int64_t)max_len;
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
    const int64_t n_paths = shared ? 1 : n_slots(h);
    const size_t bb_bytes = (size_t)n_paths * 8 * sizeof(double);
    if (bb_bytes > h->path_bbox_bytes) {
        if (h->path_bbox) HIP_TRY(hipFree(h->path_bbox));
        h->path_bbox = nullptr;
        h->path_bbox_bytes = 0;
        HIP_TRY(hipMalloc((void**)&h->path_bbox, bb_bytes));
        h->path_bbox_bytes = bb_bytes;
    }
    const size_t ix_bytes = (size_t)n_paths * 4 * kPathBuckets * sizeof(int16_t);
    if (ix_bytes > h->path_index_bytes) {
        if (h->path_index) HIP_TRY(hipFree(h->path_index));
        h->path_index = nullptr;
        h->path_index_bytes = 0;
        HIP_TRY(hipMalloc((void**)&h->path_index, ix_bytes));
        h->path_index_bytes = ix_bytes;
    }
    if (max_len > 32766) return fail(BCP_E_INVALID, "bcp_set_paths: paths longer than 32766 way points are not supported");
    hipLaunchKernelGGL(path_bbox_kernel, dim3((unsigned)((n_paths + 255) / 256)), dim3(256), 0, s, xytheta,
                       shared ? nullptr : lens, max_len, n_paths, h->dev.sp_prune, h->path_bbox);
    const int64_t n_idx = n_paths * 2 * kPathBuckets;
    hipLaunchKernelGGL(path_index_kernel, dim3((unsigned)((n_idx + 255) / 256)), dim3(256), 0, s, xytheta,
                       shared ? nullptr : lens, max_len, n_paths, h->dev.sp_prune, h->path_bbox, h->path_index);
    HIP_TRY(hipGetLastError());
    h->path.pts = h->path5;
    h->path.bbox = h->path_bbox;
    h->path.index = h->path_index;
    h->path.lens = shared ? nullptr : lens;
    h->path.max_len = max_len;
    h->path.shared = shared ? 1 : 0;
    h->have_path = true;
    h->static_dirty = true;
    return BCP_OK;
}

extern "C" int bcp_bind_state(bcp_handle* h, const bcp_state* state)
{
    if (!h) return fail(BCP_E_INVALID, "bcp_bind_state: null handle");
    if (!check_state(state, h->params.model == BCP_MODEL_TRICYCLE, &h->params))
        return fail(BCP_E_INVALID, "bcp_bind_state: missing state array (delays > 0 need pose_seen / robot_state_seen "
                                   "and the queues)");
    h->st = to_dev_state(state);
    h->have_state = true;
    h->static_dirty = true;
    return BCP_OK;
}

extern "C" int bcp_bind_initial_state(bcp_handle* h, const bcp_state* initial)
{
    if (!h) return fail(BCP_E_INVALID, "bcp_bind_initial_state: null handle");
    if (!check_state(initial, h->params.model == BCP_MODEL_TRICYCLE))
        return fail(BCP_E_INVALID, "bcp_bind_initial_state: missing state array");
    // (the initial State exposes the initial pose / robot state themselves and has empty queues: nothing more to bind)
    h->init = to_dev_state(initial);
    h->have_init = true;
    h->static_dirty = true;
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
                       (int)(h->params.model == BCP_MODEL_TRICYCLE), h->geom_of_env, h->next_geom);
    HIP_TRY(hipGetLastError());
    return BCP_OK;
}

extern "C" int bcp_broadcast_state(bcp_handle* h, int64_t src, const uint8_t* mask, void* stream)
{
    if (!h) return fail(BCP_E_INVALID, "bcp_broadcast_state: null handle");
    if (!h->have_state) return fail(BCP_E_STATE, "bcp_broadcast_state: state not bound");
    if (src < 0 || src >= h->n) return fail(BCP_E_INVALID, "bcp_broadcast_state: source env %lld of %lld", (long long)src,
                                            (long long)h->n);
    HIP_TRY(hipSetDevice(h->device));
    const bcp_params& p = h->params;
    hipLaunchKernelGGL(broadcast_state_kernel, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, h->st,
                       h->n_geoms > 0 ? h->geom_of_env : nullptr, mask, h->n, src, (int)(p.model == BCP_MODEL_TRICYCLE),
                       p.control_delay, p.pose_delay, p.state_delay);
    HIP_TRY(hipGetLastError());
    return BCP_OK;
}

// (re)builds the device-resident StepStatic block; returns whether the two-kernel (deferring) step is in effect
static bool step_uses_deferral(const bcp_handle* h)
{
    // delays and the pure-pursuit provider run through the general kernel: the optimistic finalisation of the fast
    // kernel cannot be redone once a FIFO slot has been overwritten
    const bcp_params& p = h->params;
    const bool plain = p.control_delay == 0 && p.pose_delay == 0 && p.state_delay == 0 &&
                       p.reward_provider == BCP_REWARD_CONTINUOUS;
    return plain && h->defer && h->cull.on && h->exact_mode == 0 && h->pending != nullptr;
}

static int upload_step_static(bcp_handle* h, hipStream_t s)
{
    StepStatic& S = h->host_static;
    S.P = h->dev;
    S.map = h->map;
    S.cull = h->cull;
    S.path = h->path;
    S.st = h->st;
    S.init = h->init;
    S.n = h->n;
    S.env_id_base = h->env_id_base;
    S.exact_mode = h->exact_mode;
    // fewer waves than SIMDs: nothing to balance, settle every undecided pose inside the step kernel
    S.dense_threshold = (h->n + kBlock - 1) / kBlock < 1024 && h->exact_mode == 0 ? -1 : h->dense_threshold;
    S.wide = h->wide;
    S.pending_cap = h->pending_cap;
    const bool defer = step_uses_deferral(h);
    S.pending = defer ? (Pending*)h->pending : nullptr;
    S.geom_of_env = h->n_geoms > 0 ? h->geom_of_env : nullptr;
    S.next_geom = h->n_geoms > 0 ? h->next_geom : nullptr;
    S.lds_path_doubles =
        (defer && h->path.shared && h->path.max_len * 5 * sizeof(double) <= 24 * 1024) ? h->path.max_len * 5 : 0;
    if (!h->dev_static) HIP_TRY(hipMalloc((void**)&h->dev_static, sizeof(StepStatic)));
    // pageable source: the copy is staged before the call returns, so host_static may change afterwards
    HIP_TRY(hipMemcpyAsync(h->dev_static, &S, sizeof(StepStatic), hipMemcpyHostToDevice, s));
    h->static_dirty = false;
    return BCP_OK;
}

static int launch_step(bcp_handle* h, const bcp_step_io* io, uint32_t flags, hipStream_t s, bool first_only = false)
{
    if (h->static_dirty) {
        const int rc = upload_step_static(h, s);
        if (rc != BCP_OK) return rc;
    }
    const StepStatic& S = h->host_static;
    StepArgs a;
    a.S = h->dev_static;
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
    a.pending_count = h->pending_count + (h->step_counter & 1) * kShards;
    a.pending_next = h->pending_count + ((h->step_counter + 1) & 1) * kShards;
    const bool adapt = h->adaptive && h->adapt && S.pending && S.dense_threshold >= 0;
    a.threshold_now = adapt ? h->adapt + (h->step_counter & 1) : nullptr;
    a.threshold_next = adapt ? h->adapt + ((h->step_counter + 1) & 1) : nullptr;
    a.inplace_count = adapt ? h->adapt + 2 + (h->step_counter & 1) : nullptr;
    a.inplace_next = adapt ? h->adapt + 2 + ((h->step_counter + 1) & 1) : nullptr;
    const int blocks = (int)((h->n + kBlock - 1) / kBlock);
    if (S.pending) {
        // kernel 1 settles every env the distance field decides; kernel 2 rasterises the parked rest
        const size_t lds1 = ((size_t)h->params.n_verts * 2 + S.lds_path_doubles) * sizeof(double);
        const size_t lds2 = (size_t)2 * 4 * (S.wide ? 8 : 3) * 64 * sizeof(uint32_t);
        const int waves = 2048;  // a multiple of kShards: 32 teams per shard, so that a shard rarely needs a second round
        const bool second = !first_only && S.dense_threshold >= 0;  // (threshold < 0: everything settled in place)
        if (S.wide) {
            hipLaunchKernelGGL(step_fast_kernel<true>, dim3(blocks), dim3(kBlock), lds1, s, a);
            if (second) hipLaunchKernelGGL(step_pending_kernel<true>, dim3(waves), dim3(kBlock * kPendingWaves), lds2, s, a);
        } else {
            hipLaunchKernelGGL(step_fast_kernel<false>, dim3(blocks), dim3(kBlock), lds1, s, a);
            if (second) hipLaunchKernelGGL(step_pending_kernel<false>, dim3(waves), dim3(kBlock * kPendingWaves), lds2, s, a);
        }
    } else {
        const size_t lds = collision_lds_bytes(h->params.n_verts, h->map.in_lds, h->map.rows, h->map.wpr);
        hipLaunchKernelGGL(step_kernel, dim3(blocks), dim3(kBlock), lds, s, a);
    }
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

static int time_loop(bcp_handle* h, const bcp_step_io* io, uint32_t flags, int steps, hipStream_t s, bool first_only,
                     float* avg_ms)
{
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, s));
    for (int k = 0; k < steps; ++k) launch_step(h, io, flags, s, first_only);
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

extern "C" int bcp_time_step_kernels(bcp_handle* h, const bcp_step_io* io, uint32_t flags, int32_t steps, void* stream,
                                     float* kernel_ms)
{
    int rc = check_step(h, io, flags, "bcp_time_step_kernels");
    if (rc != BCP_OK) return rc;
    if (steps <= 0 || !kernel_ms) return fail(BCP_E_INVALID, "bcp_time_step_kernels: bad steps / output");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    // full steps first (the state advances), then the same number of step_kernel-only launches on the reached state:
    // envs parked by a lone step_kernel are never finished, so every launch of that loop sees the same batch.
    float full = 0, first = 0;
    rc = time_loop(h, io, flags, steps, s, false, &full);
    if (rc != BCP_OK) return rc;
    rc = time_loop(h, io, flags, steps, s, true, &first);
    if (rc != BCP_OK) return rc;
    kernel_ms[0] = first;
    kernel_ms[1] = full > first ? full - first : 0.0f;
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
    hipLaunchKernelGGL(pose_collides_kernel, dim3(blocks), dim3(kBlock),
                       collision_lds_bytes(h->params.n_verts, h->map.in_lds, h->map.rows, h->map.wpr),
                       (hipStream_t)stream, h->dev, h->map, h->cull, h->exact_mode, h->dense_threshold, h->wide, poses, n,
                       h->n, h->geom_of_env, out);
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
    if (h->exact_mode == 2) {  // per-thread rasteriser
        const int blocks = (int)((n + kBlock - 1) / kBlock);
        hipLaunchKernelGGL(pixel_footprint_thread_kernel, dim3(blocks), dim3(kBlock),
                           (size_t)h->params.n_verts * 2 * kBlock * sizeof(uint32_t), s, P, angles, n, masks, side, shape_hw);
    } else {                   // cooperative rasteriser: one wave per angle
        hipLaunchKernelGGL(pixel_footprint_kernel, dim3((unsigned)n), dim3(kBlock),
                           (size_t)h->params.n_verts * 2 * sizeof(double), s, P, angles, n, masks, side, shape_hw);
    }
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

// ---- egocentric observation ----------------------------------------------------------------------------------
static int ego_shape(const bcp_handle* h, const double* window_size, int32_t* drows, int32_t* dcols)
{
    if (window_size) {
        const double inv = 1.0 / h->resolution;
        *dcols = (int32_t)std::nearbyint(window_size[0] * inv);  // world_to_pixel(resulting_size, (0, 0), resolution)
        *drows = (int32_t)std::nearbyint(window_size[1] * inv);
    } else {
        *drows = h->map.rows;
        *dcols = h->map.cols;
    }
    return *drows > 0 && *dcols > 0 && (int64_t)*drows * *dcols * *dcols < (int64_t)1 << 32 && *dcols <= 8192 && *drows <= 8192;
}

extern "C" int bcp_egocentric_shape(bcp_handle* h, const double* window_size, int32_t* shape_hw)
{
    if (!h || !shape_hw) return fail(BCP_E_INVALID, "bcp_egocentric_shape: null argument");
    if (!h->have_map) return fail(BCP_E_STATE, "bcp_egocentric_shape: costmaps not set");
    if (!ego_shape(h, window_size, &shape_hw[0], &shape_hw[1]))
        return fail(BCP_E_INVALID, "bcp_egocentric_shape: unsupported window size");
    return BCP_OK;
}

extern "C" int bcp_egocentric_costmaps(bcp_handle* h, const double* poses, int64_t n, const double* window_origin,
                                       const double* window_size, uint8_t border_value, uint8_t* out, void* stream)
{
    if (!h || !out) return fail(BCP_E_INVALID, "bcp_egocentric_costmaps: null argument");
    if (!h->have_map) return fail(BCP_E_STATE, "bcp_egocentric_costmaps: costmaps not set");
    if (!poses && !h->have_state) return fail(BCP_E_STATE, "bcp_egocentric_costmaps: no poses given and no state bound");
    if (n <= 0 || (!poses && n != h->n)) return fail(BCP_E_INVALID, "bcp_egocentric_costmaps: n must be n_envs without poses");
    if ((window_origin == nullptr) != (window_size == nullptr))
        return fail(BCP_E_INVALID, "bcp_egocentric_costmaps: window origin and size go together");
    EgoArgs a;
    memset(&a, 0, sizeof(a));
    if (!ego_shape(h, window_size, &a.drows, &a.dcols))
        return fail(BCP_E_INVALID, "bcp_egocentric_costmaps: unsupported window size");
    HIP_TRY(hipSetDevice(h->device));
    a.data = h->map_data;
    a.shared = h->map.shared;
    a.rows = h->map.rows;
    a.cols = h->map.cols;
    a.map_stride = a.shared ? 0 : (int64_t)a.rows * a.cols;
    a.valid_rows = h->map_valid_rows;
    a.valid_cols = h->map_valid_cols;
    a.origins = h->map.origins;
    a.ox = h->map.ox;
    a.oy = h->map.oy;
    a.res = h->resolution;
    a.inv_res = h->map.inv_res;
    a.poses = poses;
    a.sx = h->st.x;
    a.sy = h->st.y;
    a.sth = h->st.angle;
    if (h->params.pose_delay > 0 && h->st.pose_seen) {   // the observation shows State.pose, i.e. the delayed pose
        a.sx = h->st.pose_seen;
        a.sy = h->st.pose_seen + h->n;
        a.sth = h->st.pose_seen + 2 * h->n;
    }
    a.geom_of_env = h->n_geoms > 0 ? h->geom_of_env : nullptr;
    a.has_window = window_origin != nullptr;
    if (window_origin) {
        a.win_ox = window_origin[0];
        a.win_oy = window_origin[1];
    }
    const size_t map_bytes = ((size_t)(a.rows + 2) * (a.cols + 2) + 3) & ~(size_t)3;   // LDS copy with a border ring
    const size_t row_bytes = (size_t)a.drows * 2 * sizeof(int32_t);                   // one row table
    a.border = border_value;
    a.out = out;
    a.n_envs = h->n;
    a.n_images = n;
    a.cols_magic = (uint32_t)(((uint64_t)1 << 32) / (uint64_t)a.cols) + 1;   // (staged maps are < 64 KB: exact)
    if (a.dcols < 4) return fail(BCP_E_INVALID, "bcp_egocentric_costmaps: windows narrower than 4 px are not supported");
    const bool px8 = a.dcols >= 8;   // 8 pixels (one 64-bit store) per lane; narrow windows fall back to 4
    hipStream_t st = (hipStream_t)stream;
    int cus = 0;
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device));
    cus = std::max(cus, 1);
    const dim3 block(256);
    if (!a.shared && map_bytes + 4 * row_bytes <= 60 * 1024 && n < ((int64_t)1 << 31)) {
        // private / pooled maps that fit LDS: group the images by map entry, then one workgroup per entry at a time
        const int64_t n_bins = n_slots(h);
        if (n_bins > h->ego_bins_cap) {
            if (h->ego_bins) HIP_TRY(hipFree(h->ego_bins));
            h->ego_bins = nullptr;
            h->ego_bins_cap = 0;
            HIP_TRY(hipMalloc((void**)&h->ego_bins, (size_t)2 * n_bins * sizeof(int32_t)));
            h->ego_bins_cap = n_bins;
        }
        if (n > h->ego_order_cap) {
            if (h->ego_order) HIP_TRY(hipFree(h->ego_order));
            h->ego_order = nullptr;
            h->ego_order_cap = 0;
            HIP_TRY(hipMalloc((void**)&h->ego_order, (size_t)2 * n * sizeof(int32_t)));
            h->ego_order_cap = n;
        }
        int32_t* bin_count = h->ego_bins;
        int32_t* bin_start = h->ego_bins + h->ego_bins_cap;
        int32_t* rank = h->ego_order;
        int32_t* order = h->ego_order + h->ego_order_cap;
        HIP_TRY(hipMemsetAsync(bin_count, 0, (size_t)n_bins * sizeof(int32_t), st));
        const dim3 per_image((unsigned)((n + 255) / 256));
        hipLaunchKernelGGL(ego_bin_count_kernel, per_image, block, 0, st, a.geom_of_env, a.n_envs, n, bin_count, rank);
        hipLaunchKernelGGL(ego_bin_scan_kernel, dim3(1), dim3(1024), 0, st, bin_count, n_bins, bin_start);
        hipLaunchKernelGGL(ego_bin_scatter_kernel, per_image, block, 0, st, a.geom_of_env, a.n_envs, n, bin_start, rank, order);
        const size_t lds = map_bytes + 4 * row_bytes;
        const void* fn = px8 ? (const void*)ego_costmap_binned_kernel<8> : (const void*)ego_costmap_binned_kernel<4>;
        int per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, lds));
        const dim3 grid((unsigned)std::min<int64_t>(n, (int64_t)std::max(per_cu, 1) * cus));
        a.stage_map = 1;
        if (px8) hipLaunchKernelGGL((ego_costmap_binned_kernel<8>), grid, block, lds, st, a, bin_start, bin_count, order);
        else hipLaunchKernelGGL((ego_costmap_binned_kernel<4>), grid, block, lds, st, a, bin_start, bin_count, order);
    } else {
        // shared map (staged in LDS when it fits) or maps too large for LDS (sampled from global memory):
        // persistent workgroups, as many as are resident at once
        a.stage_map = (a.shared && map_bytes + 4 * row_bytes <= 60 * 1024) ? 1 : 0;
        const size_t lds = 4 * row_bytes + (a.stage_map ? map_bytes : 0);
        const void* fn = a.stage_map ? (px8 ? (const void*)ego_costmap_kernel<true, 8> : (const void*)ego_costmap_kernel<true, 4>)
                                     : (px8 ? (const void*)ego_costmap_kernel<false, 8> : (const void*)ego_costmap_kernel<false, 4>);
        int per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, lds));
        const dim3 grid((unsigned)std::min<int64_t>((n + 3) / 4, (int64_t)std::max(per_cu, 1) * cus));
        if (a.stage_map) {
            if (px8) hipLaunchKernelGGL((ego_costmap_kernel<true, 8>), grid, block, lds, st, a);
            else hipLaunchKernelGGL((ego_costmap_kernel<true, 4>), grid, block, lds, st, a);
        } else {
            if (px8) hipLaunchKernelGGL((ego_costmap_kernel<false, 8>), grid, block, lds, st, a);
            else hipLaunchKernelGGL((ego_costmap_kernel<false, 4>), grid, block, lds, st, a);
        }
    }
    HIP_TRY(hipGetLastError());
    return BCP_OK;
}

extern "C" int bcp_goal_n_state(bcp_handle* h, const double* world_size, float* out, void* stream)
{
    if (!h || !world_size || !out) return fail(BCP_E_INVALID, "bcp_goal_n_state: null argument");
    if (!h->have_path || !h->have_state) return fail(BCP_E_STATE, "bcp_goal_n_state: paths and state must be set first");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    if (h->static_dirty) {
        const int rc = upload_step_static(h, s);
        if (rc != BCP_OK) return rc;
    }
    const int n_state = h->params.model == BCP_MODEL_TRICYCLE ? 6 : 5;
    hipLaunchKernelGGL(goal_n_state_kernel, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, s, h->dev_static,
                       world_size[0], world_size[1], n_state, out);
    HIP_TRY(hipGetLastError());
    return BCP_OK;
}
