// bcplan.hip -- libbcplan.so: batched PlanEnv.step() for MI355X (gfx950).  C ABI in include/bcplan.h.
//
// This file holds the handle, the set-up kernels (lethal bitmap, distance transform, path tables), the operator
// seams and every C entry point.  The step itself lives in bcp_step.h (robot model -> collision classification /
// exact rasteriser -> rollback -> reward provider -> done -> optional reset -> state write-back; device code in
// bcp_device.h, bcp_raster.h, bcp_coop.h), the egocentric observation in bcp_ego.h.
// Compiled with -ffp-contract=off (numpy rounds every product and sum separately).  No CPU path exists here.
#include <hip/hip_runtime.h>
#include <mutex>

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
#include "bcp_step.h"
#include "bcp_ego.h"
#include "bcp_sample.h"

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
struct bcp_handle {
    bcp_params params;
    DevParams dev;
    int64_t n;
    int device;
    int64_t env_id_base;
    uint64_t seed;
    uint64_t* tick;           // owned, device: step counter (two views), noise seed, ticket -- see StepArgs::tick; [4]: waits that gave up
    bool have_map, have_path, have_state, have_init;
    double resolution;
    uint32_t* bitmap;      // owned
    size_t bitmap_bytes;
    uint32_t* near_coarse; // owned: CullDesc::step_near when it is not the tiles themselves
    size_t near_coarse_bytes;
    int32_t near_shift;    // BCP_NEAR_SHIFT / BCP_TUNE_NEAR_SHIFT: resolution of step_near for private maps (-1: the library's rule)
    uint32_t* map_tiles;   // owned: the bitmap once more in tiles of 32 x 32 cells (MapDesc::tiles)
    size_t map_tiles_bytes;
    double* path5;         // owned
    size_t path5_bytes;
    uint32_t* path_pre;    // owned: [paths][max_len][2] {x, y as uint16 steps | cos, sin as int16}: the prefilter record of private paths
    size_t path_pre_bytes;
    double* path_bbox;     // owned
    size_t path_bbox_bytes;
    int16_t* path_index;   // owned
    size_t path_index_bytes;
    uint8_t* edt;          // owned: distance transform of the shared costmap (padded)
    size_t edt_bytes;
    uint8_t* edt_col;      // owned scratch of the transform
    size_t edt_col_bytes;
    uint32_t* near;        // owned: the field as 1-bit tiles (CullDesc::near)
    size_t near_bytes;
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
    int32_t exact_mode;       // 0 auto, 1 cooperative only, 2 per-thread only, 3 cooperative cell by cell
    int32_t dense_threshold;  // auto: more ambiguous lanes than this in a wave -> per-thread rasteriser
    int32_t adaptive;         // the threshold above is only the fallback: kernel 2 re-decides every step
    int32_t* adapt;           // owned: [2] thresholds + [2] in-place counters, alternating by step parity
    int32_t cull_enabled;
    int32_t wide;             // kernel image may exceed 96 px: 8-word row masks in the cooperative path
    int32_t* ego_bins;        // owned: [2][bins] image counts / first slots per map entry (egocentric views)
    int64_t ego_bins_cap;
    int32_t* ego_order;       // owned: [2][images] rank within the bin / images grouped by map entry
    int64_t ego_order_cap;
    // sparse egocentric views (ego_sparse_kernel): per map entry the list of its non-zero cells
    uint32_t* ego_cells;      // owned: [entries][ego_cell_cap] (nullptr while the maps count as dense)
    int32_t* ego_cell_counts; // owned: [entries] + [1] running maximum
    int64_t ego_cells_entries;
    int32_t ego_cell_cap;     // stride of a list, sized from the counting pass
    bool ego_cells_built;     // counts (and lists, if any) describe the current maps (rebuilt entry by entry by a pool refresh)
    bool ego_cells_refused;   // allocation failed once: the sampling kernels serve this handle
    int32_t ego_cells_max;    // host copy of the maximum count, -1 = not fetched since the last (re)build
    int32_t ego_sparse;       // BCP_TUNE_EGO_SPARSE: 0 never, 1 cost model, >= 2 explicit limit of cells per map
    int32_t ego_stride;       // BCP_TUNE_EGO_LIST_STRIDE: 0 = lists sized from the counts, else this many cells per entry (tests)
    int32_t ego_route[4];     // what the last bcp_egocentric_costmaps call ran: kernel, largest count, list stride, limit
    // watchdog of the step kernel's bounded waits: every kWatchdogSteps calls bcp_step copies tick[4] to pinned host memory
    // behind the step (no synchronisation) and a later call looks at what arrived
    uint64_t* waits_host;     // owned, pinned
    hipEvent_t waits_event;   // owned
    bool waits_in_flight;
    uint64_t waits_seen;
    uint32_t steps_since_probe;
    hipEvent_t refresh_done;  // owned: end of the last bcp_refresh_mini_worlds (whoever derives data from the maps on
    bool refresh_recorded;    // another stream waits for it first)
    hipStream_t side_stream;  // owned: the CU-masked stream of bcp_side_stream (nullptr: not created)
    int32_t side_share;       // ... and the share of the CUs it was created with
    const uint8_t* map_data;  // caller-owned raw costmap(s) as given to bcp_set_costmaps (egocentric views read them)
    const int32_t* map_valid_rows;
    const int32_t* map_valid_cols;
    int32_t n_geoms;          // > 0: geometry pool of that many entries
    int32_t* geom_of_env;     // caller-owned device int32 [n]
    const int32_t* next_geom; // caller-owned device int32 [n_geoms] or nullptr
    const double* path_src;   // caller-owned way points [.,max_len,3] as given to bcp_set_paths
    int32_t* ring;            // owned scratch of bcp_refresh_mini_worlds
    size_t ring_bytes;
    int32_t ring_episodes;    // of the last bcp_plan_mini_worlds
    bool ring_planned, ring_refreshed;   // plan -> refresh -> release, in that order
    int32_t edt_in_lds;       // distance transform of maps that fit: the LDS-resident kernel (BCP_TUNE_EDT_LDS)
    int32_t last_step_form;   // 0 none yet, 1 single-kernel step, 2 two-kernel step (parking counters in use)
    int32_t fused;            // settle parked poses inside the step launch (step_local_kernel) instead of a second launch
    uint64_t* parked_slots;   // owned: a word per workgroup of step_local_kernel, its parked poses so far (bcp_parked_poses)
    int64_t parked_cap;
    int32_t local_pairs;      // BCP_TUNE_LOCAL_PAIRS: workgroup size of step_local_kernel (0 = default, 1, 2, 4 x 64 envs)
    // near_dilate_kernel: 1-bit tiles without the uint8 field (pool refresh under the single-launch step)
    int32_t near_dilate;      // BCP_TUNE_NEAR_DILATE: 0 never, 1 pool refreshes (default), 2 every build (after the field: tests)
    uint8_t* edt_stale;       // owned: [entries] 1 = the entry's uint8 field does not describe its map (tiles do)
    int32_t* edt_stale_list;  // owned: [entries] + [1] count, scratch of ensure_fields
    int64_t edt_stale_cap;
    bool edt_lazy;            // a refresh has left stale fields behind since the last full build
};

// number of entries of a non-shared map / path / initial-state array
static int64_t n_slots(const bcp_handle* h) { return h->n_geoms > 0 ? h->n_geoms : h->n; }

// grid of a grid-stride kernel; a selection's size is only known on the device, so those launches get a chip-filling
// grid that does not grow with the upper bound
constexpr size_t kMaxDynamicLds = 150 * 1024;   // of the 160 KB a gfx950 workgroup can have

static unsigned stride_grid(int64_t work_items, int threads, bool selection = false)
{
    return (unsigned)std::max<int64_t>(1, std::min<int64_t>((work_items + threads - 1) / threads, selection ? 4096 : 65536));
}

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

// ------------------------------------------------------------------------------------------------ kernels (one-time, operator seams)
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
                                                               uint8_t* __restrict__ out, int origin_in_map,
                                                               const int32_t* __restrict__ valid_rows,
                                                               const int32_t* __restrict__ valid_cols)
{
    const int tid = threadIdx.x;
    const int64_t gi = (int64_t)blockIdx.x * kBlock + tid;
    const bool active = gi < n;
    const int64_t i = active ? gi : n - 1;
    const CollisionLds L = collision_lds_setup(P, map, tid);
    const int64_t env = geom_of_env ? (int64_t)geom_of_env[i % n_envs] : i % n_envs;
    bool hit = collides_wave(P, map, cull, L, exact_mode, dense_threshold, wide != 0, active, env, poses[3 * i],
                             poses[3 * i + 1], poses[3 * i + 2]);
    if (origin_in_map) {   // is_robot_colliding: a robot whose own pixel is off the map never collides (costmap_utils.py:127-130)
        const double ox = map.origins ? map.origins[2 * env] : map.ox, oy = map.origins ? map.origins[2 * env + 1] : map.oy;
        const int64_t px = (int64_t)rint((poses[3 * i] - ox) * map.inv_res), py = (int64_t)rint((poses[3 * i + 1] - oy) * map.inv_res);
        const int rows = (!map.shared && valid_rows) ? valid_rows[env] : map.rows;
        const int cols = (!map.shared && valid_cols) ? valid_cols[env] : map.cols;
        if (px < 0 || py < 0 || px >= cols || py >= rows) hit = false;
    }
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
__device__ __forceinline__ void edt_column(const uint32_t* __restrict__ bits, int64_t m, int cp, int rows, int cols, int wpr,
                                           int pad, int clamp, uint8_t* __restrict__ g)
{
    const int W = cols + 2 * pad, H = rows + 2 * pad;
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

__global__ void edt_columns_kernel(const uint32_t* __restrict__ bits, EntrySelect sel, int rows, int cols, int wpr, int pad,
                                   int clamp, uint8_t* __restrict__ g)
{
    const int W = cols + 2 * pad;
    const int64_t total = sel.size() * W;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x)
        edt_column(bits, sel.entry(t / W), (int)(t % W), rows, cols, wpr, pad, clamp, g);
}

// pass 2: d^2(r,c) = min over |c - c'| < clamp of (c - c')^2 + g(r,c')^2, stored as floor(min(clamp, d)).
__device__ __forceinline__ void edt_cell(const uint8_t* __restrict__ g, int64_t idx, int W, int clamp, uint8_t* __restrict__ out)
{
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

__global__ void edt_rows_kernel(const uint8_t* __restrict__ g, EntrySelect sel, int W, int H, int clamp,
                                uint8_t* __restrict__ out)
{
    const int64_t per = (int64_t)W * H, total = sel.size() * per;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (int64_t)gridDim.x * blockDim.x)
        edt_cell(g, sel.entry(it / per) * per + it % per, W, clamp, out);
}

// The distance field as one bit per cell, "a lethal cell is closer than t_out", in 32 x 32-cell tiles (CullDesc::near):
// all the outer test of the step asks.  One thread per output word = 32 consecutive cells of one row.
typedef uint32_t __attribute__((aligned(1))) EdtUnalignedWord;
__global__ void near_tiles_kernel(const uint8_t* __restrict__ edt, EntrySelect sel, int W, int H, int tiles_x, int tiles_y,
                                  int t_out, uint32_t* __restrict__ tiles)
{
    const int64_t per = (int64_t)tiles_x * tiles_y * 32, total = sel.size() * per;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (int64_t)gridDim.x * blockDim.x) {
        const int tx = (int)(it % tiles_x);
        const int64_t t = it / tiles_x;
        const int y = (int)(t % (tiles_y * 32));
        const int64_t e = sel.entry(t / (tiles_y * 32));
        uint32_t word = 0;
        if (y < H) {
            const uint8_t* row = edt + (e * H + y) * (int64_t)W + tx * 32;
            if (tx * 32 + 32 <= W) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t four = *reinterpret_cast<const EdtUnalignedWord*>(row + 4 * k);
#pragma unroll
                    for (int j = 0; j < 4; ++j) word |= (uint32_t)((int)((four >> (8 * j)) & 255u) < t_out) << (4 * k + j);
                }
            } else {
                for (int j = 0; tx * 32 + j < W; ++j) word |= (uint32_t)((int)row[j] < t_out) << j;
            }
        }
        tiles[e * per + ((int64_t)(y >> 5) * tiles_x + tx) * 32 + (y & 31)] = word;
    }
}

// CullDesc::step_near: the tiles at 1 / 2^shift of the resolution, a bit = the OR of the 2^shift x 2^shift bits it stands
// for.  One thread per output word: 2^shift rows of 2^shift neighbouring tiles, OR-ed and squeezed.
__global__ void near_coarsen_kernel(const uint32_t* __restrict__ tiles, EntrySelect sel, int tiles_x, int tiles_y, int shift,
                                    int ctx, int cty, uint32_t* __restrict__ coarse)
{
    const int64_t per = (int64_t)tiles_x * tiles_y * 32, cper = (int64_t)ctx * cty * 32, total = sel.size() * cper;
    const int f = 1 << shift, bits_out = 32 >> shift;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = sel.entry(it / cper);
        const int k = (int)(it % cper);
        const int Y = (k / (32 * ctx)) * 32 + (k & 31), TX = (k >> 5) % ctx;   // coarse row, coarse tile column
        uint32_t word = 0;
        for (int part = 0; part < f; ++part) {          // fine tile column part of this coarse word
            const int tx = TX * f + part;
            uint32_t rows = 0;
            for (int dy = 0; dy < f; ++dy) {
                const int y = Y * f + dy;
                if (tx < tiles_x && y < tiles_y * 32) rows |= tiles[e * per + ((int64_t)(y >> 5) * tiles_x + tx) * 32 + (y & 31)];
            }
            uint32_t squeezed = 0;
            for (int b = 0; b < bits_out; ++b) squeezed |= (uint32_t)(((rows >> (b << shift)) & ((1u << f) - 1u)) != 0) << b;
            word |= squeezed << (part * bits_out);
        }
        coarse[e * cper + k] = word;
    }
}

// The same transform for maps that fit into LDS (every private / pool map), one workgroup per map, `clamp` <= 60:
//   pass 1: h(r, c) = distance to the nearest lethal cell of ROW r, from the row's bit mask with clz / ctz on the 64 bits
//           either side of c -- no sweep, every cell on its own; four cells per thread, packed into an LDS dword;
//   pass 2: d^2(r, c) = min over |r - r'| < clamp of (r - r')^2 + h(r', c)^2, rows taken from the centre outwards and
//           abandoned once (r - r')^2 alone reaches the best value so far.
// It computes the very min the two kernels above compute (the order of the two 1-D passes does not matter), from LDS
// instead of through the caches: ~20 x faster, which is what lets a pool be topped up between steps.
__device__ __forceinline__ uint32_t edt_row_word(LdsWords row, int wpr, int w) { return (w >= 0 && w < wpr) ? row[w] : 0u; }

// bit i = column start + i of the row (zero outside the map), i = 0 .. 63
__device__ __forceinline__ uint64_t edt_row_window(LdsWords row, int wpr, int start)
{
    const int w0 = start >> 5, sh = start & 31;   // (arithmetic shift: floor for negative starts)
    const uint64_t lo = ((uint64_t)edt_row_word(row, wpr, w0 + 1) << 32) | edt_row_word(row, wpr, w0);
    const uint64_t hi = edt_row_word(row, wpr, w0 + 2);
    return sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
}

typedef unsigned short EdtU16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t __attribute__((aligned(1))) EdtU32Unaligned;

__device__ __forceinline__ EdtU16x2 edt_pair(uint32_t word, uint32_t selector)
{
    const uint32_t v = __builtin_amdgcn_perm(0u, word, selector);
    return __builtin_bit_cast(EdtU16x2, v);
}

__global__ void __launch_bounds__(256) edt_lds_kernel(const uint32_t* __restrict__ bits, EntrySelect sel, int rows, int cols,
                                                      int wpr, int pad, int clamp, uint8_t* __restrict__ out)
{
    const int W = cols + 2 * pad, H = rows + 2 * pad, Wq = (W + 3) / 4;
    const LdsU32 bm = (LdsU32)lds_dyn;   // [rows][wpr] lethal mask
    const LdsU32 hq = bm + rows * wpr;   // [H][Wq] h, four cells per dword
    __attribute__((address_space(3))) uint8_t* const isq =
        (__attribute__((address_space(3))) uint8_t*)(hq + H * Wq);   // [clamp^2 + 1] min(clamp, floor(sqrt(.)))
    const int tid = threadIdx.x;
    const uint32_t far4 = (uint32_t)clamp * 0x01010101u;
    const int64_t n_sel = sel.size();
    for (int v = tid; v <= clamp * clamp; v += 256) {
        int sq = (int)__builtin_amdgcn_sqrtf((float)v);   // v <= 3600: the fix-ups make it exact
        while (sq * sq > v) --sq;
        while ((sq + 1) * (sq + 1) <= v) ++sq;
        isq[v] = (uint8_t)min(sq, clamp);
    }
    for (int64_t k = blockIdx.x; k < n_sel; k += gridDim.x) {
        const int64_t m = sel.entry(k);
        __syncthreads();   // the previous map's pass 2 is done with the LDS
        for (int i = tid; i < rows * wpr; i += 256) bm[i] = bits[m * (int64_t)rows * wpr + i];
        __syncthreads();
        // (a wave per row, a lane per group of four cells: no divisions, and the four cells share their two windows)
        for (int rp = tid >> 6; rp < H; rp += 4) {
            const int r = rp - pad;
            for (int q = tid & 63; q < Wq; q += 64) {
                uint32_t packed = far4;
                if (r >= 0 && r < rows) {
                    const LdsWords row = (LdsWords)(bm + r * wpr);
                    const int c0 = q * 4 - pad;
                    // left: bit 63 = column c0, bit 63 - j = column c0 - j;  right: bit j = column c0 + j
                    const uint64_t left = edt_row_window(row, wpr, c0 - 63), right = edt_row_window(row, wpr, c0);
                    packed = 0;
#pragma unroll
                    for (int b = 0; b < 4; ++b) {   // the same two windows seen from column c0 + b (clamp <= 60)
                        const uint64_t lb = (left >> b) | (right << (63 - b)), rb = right >> b;
                        const int dr = rb ? (int)__builtin_ctzll(rb) : 64, dl = lb ? (int)__builtin_clzll(lb) : 64;
                        packed |= (uint32_t)min(min(dr, dl), clamp) << (8 * b);
                    }
                }
                hq[rp * Wq + q] = packed;
            }
        }
        __syncthreads();
        uint8_t* field = out + m * (int64_t)W * H;
        // (two cells per packed 16-bit operation: all values are <= 2 * clamp^2 <= 7200; the square roots come from a
        //  table; no early exit -- it would cost as much per round as the round itself)
        for (int rp = tid >> 6; rp < H; rp += 4) {
            for (int q = tid & 63; q < Wq; q += 64) {
                EdtU16x2 best_lo = {(unsigned short)(clamp * clamp), (unsigned short)(clamp * clamp)}, best_hi = best_lo;
                for (int d = 0; d < clamp; ++d) {
                    const unsigned short dd = (unsigned short)(d * d);
                    const EdtU16x2 dd2 = {dd, dd};
                    const uint32_t up = rp - d >= 0 ? hq[(rp - d) * Wq + q] : far4;
                    const uint32_t dn = rp + d < H ? hq[(rp + d) * Wq + q] : far4;
                    // bytes 0, 1 / 2, 3 of a dword, zero-extended to a pair of 16-bit values (v_perm_b32)
                    const EdtU16x2 h_lo = __builtin_elementwise_min(edt_pair(up, 0x0c010c00u), edt_pair(dn, 0x0c010c00u));
                    const EdtU16x2 h_hi = __builtin_elementwise_min(edt_pair(up, 0x0c030c02u), edt_pair(dn, 0x0c030c02u));
                    best_lo = __builtin_elementwise_min(best_lo, (EdtU16x2)(h_lo * h_lo + dd2));
                    best_hi = __builtin_elementwise_min(best_hi, (EdtU16x2)(h_hi * h_hi + dd2));
                }
                const uint32_t four = (uint32_t)isq[best_lo.x] | ((uint32_t)isq[best_lo.y] << 8) |
                                      ((uint32_t)isq[best_hi.x] << 16) | ((uint32_t)isq[best_hi.y] << 24);
                uint8_t* const dst = field + rp * W + q * 4;
                if (q * 4 + 3 < W) {
                    *reinterpret_cast<EdtU32Unaligned*>(dst) = four;
                } else {
                    for (int b = 0; q * 4 + b < W; ++b) dst[b] = (uint8_t)(four >> (8 * b));
                }
            }
        }
    }
}

// The 1-bit tiles WITHOUT the distance field: bit (x, y) = "a lethal cell lies within dx^2 + dy^2 < t_out^2" is the lethal
// mask dilated by a disc, and a disc is a stack of horizontal runs: with reach(w) = isqrt(t_out^2 - 1 - w^2),
//     near(x, y) = OR over |w| < t_out of  V_|w|(x + w, y),     V_w(x, y) = OR over |dy| <= reach(w) of lethal(x, y + dy).
// reach() grows as w shrinks, so one pass from w = t_out - 1 down to 0 ORs every row within reach into a 96-bit window
// exactly once and shifts the window by +-w: ~250 integer instructions per 32-cell output word against ~1200 of the
// distance transform + threshold (edt_lds_kernel + near_tiles_kernel), and no 16 KB uint8 field to write and read back.
// The bits are those of near_tiles_kernel by construction (floor(sqrt(D2)) < t_out  <=>  D2 <= t_out^2 - 1; the transform's
// windows are wider than t_out); tests/test_gpu_pool.py::test_near_tiles_by_dilation_vs_thresholded_field compares the two word for word.  What a pool refresh runs
// while the steps only read the tiles (step_local_kernel); the uint8 field of such entries is marked stale (ensure_fields).
// One workgroup per map; LDS: the padded lethal rows with a zero word either side and t_out - 1 zero rows above and below.
__global__ void __launch_bounds__(256) near_dilate_kernel(const uint32_t* __restrict__ bits, EntrySelect sel, int rows, int cols,
                                                          int wpr, int pad, int t_out, int W, int H, int tiles_x, int tiles_y,
                                                          uint32_t* __restrict__ tiles, uint8_t* __restrict__ stale)
{
    const LdsU32 P = (LdsU32)lds_dyn;
    const int tid = threadIdx.x;
    const int margin = t_out - 1, pitch = tiles_x + 2, Ht = tiles_y * 32, Hp = Ht + 2 * margin;
    const LdsU32 reach = P + Hp * pitch;   // [t_out]
    for (int w = tid; w < t_out; w += 256) {
        const int v = t_out * t_out - 1 - w * w;
        int sq = (int)__builtin_amdgcn_sqrtf((float)v);   // v < 1024: the fix-ups make it exact
        while (sq * sq > v) --sq;
        while ((sq + 1) * (sq + 1) <= v) ++sq;
        reach[w] = (uint32_t)sq;
    }
    const int64_t n_sel = sel.size(), per = (int64_t)tiles_x * tiles_y * 32;
    for (int64_t k = blockIdx.x; k < n_sel; k += gridDim.x) {
        const int64_t m = sel.entry(k);
        const uint32_t* mb = bits + m * (int64_t)rows * wpr;
        __syncthreads();   // the previous map's words are no longer read
        // P[yp][1 + kx] bit b = lethal(column 32 kx + b - pad, row yp - margin - pad); zero outside the map
        for (int i = tid; i < Hp * pitch; i += 256) {
            const int yp = i / pitch, kp = i - yp * pitch;
            const int r = yp - margin - pad, start = 32 * (kp - 1) - pad;
            uint32_t word = 0;
            if (r >= 0 && r < rows && kp >= 1 && kp <= tiles_x) {
                const int w0 = start >> 5, sh = start & 31;   // (arithmetic shift: floor for negative starts)
                const uint32_t lo = (w0 >= 0 && w0 < wpr) ? mb[r * wpr + w0] : 0u;
                const uint32_t hi = (w0 + 1 >= 0 && w0 + 1 < wpr) ? mb[r * wpr + w0 + 1] : 0u;
                word = sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
                const int left = cols + pad - 32 * (kp - 1);   // columns >= cols are not part of the map
                word = left >= 32 ? word : (left > 0 ? word & ((1u << left) - 1u) : 0u);
            }
            P[i] = word;
        }
        __syncthreads();
        for (int i = tid; i < Ht * tiles_x; i += 256) {
            const int y = i / tiles_x, kx = i - y * tiles_x;
            const LdsU32 centre = P + (y + margin) * pitch + kx;   // words kx - 1, kx, kx + 1 of row y
            uint32_t a = 0, b = 0, c = 0, word = 0;
            int in = -1;
            for (int w = t_out - 1; w >= 0; --w) {
                const int need = (int)reach[w];
                while (in < need) {
                    ++in;
                    const LdsU32 up = centre - in * pitch, dn = centre + in * pitch;
                    a |= up[0] | dn[0];
                    b |= up[1] | dn[1];
                    c |= up[2] | dn[2];
                }
                word |= w ? (b << w) | (a >> (32 - w)) | (b >> w) | (c << (32 - w)) : b;
            }
            const int left = W - 32 * kx;   // near_tiles_kernel leaves cells outside the padded field clear
            word = (y < H) ? (left >= 32 ? word : (left > 0 ? word & ((1u << left) - 1u) : 0u)) : 0u;
            tiles[m * per + ((int64_t)(y >> 5) * tiles_x + kx) * 32 + (y & 31)] = word;
        }
        if (stale && tid == 0) stale[m] = 1;
    }
}

// entries whose uint8 distance field is stale (near_dilate_kernel ran for them) -> a list for edt_lds_kernel & co.
__global__ void stale_fields_list_kernel(uint8_t* __restrict__ stale, int64_t n, int32_t* __restrict__ list, int32_t* __restrict__ count)
{
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
        if (stale[e]) {
            stale[e] = 0;
            list[atomicAdd(count, 1)] = (int32_t)e;
        }
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

// ColoredEgoCostmapRandomAisleTurnEnv's `goal` vector (envs/synth_turn_env.py:412-420), one thread per env: the LAST way
// point in the robot frame over the window's world size, normalised to unit length, then (v, w, wheel_angle)
__global__ void goal_direction_state_kernel(const StepStatic* __restrict__ S, double wsx, double wsy, double* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S->n) return;
    const int64_t g = S->geom_of_env ? (int64_t)S->geom_of_env[i] : i;
    const int m = S->path.shared ? S->path.max_len : S->path.lens[g];
    const double* wp = S->path.pts + ((S->path.shared ? 0 : g * (int64_t)S->path.max_len) + (m - 1)) * 5;
    const double x = S->st.x[i], y = S->st.y[i], th = S->st.angle[i];   // the robot's own pose (not the delayed one)
    const double c = cos(th), s = sin(th);
    const double tx = -x * c - y * s, ty = x * s - y * c, tt = normalize_angle(-th);
    const double ct = cos(tt), st = sin(tt);
    const double gx = (ct * wp[0] + (-st) * wp[1] + tx) / wsx, gy = (st * wp[0] + ct * wp[1] + ty) / wsy;
    const double norm = sqrt(fma(gy, gy, gx * gx));   // np.linalg.norm: fma-contracted 2-term dot
    double* o = out + 5 * i;
    o[0] = gx / norm;
    o[1] = gy / norm;
    o[2] = S->st.v[i];
    o[3] = S->st.w[i];
    o[4] = S->P.model == BCP_MODEL_TRICYCLE ? S->st.wheel[i] : 0.0;
}


// ---- reward-provider / path-tools operator seams (envs/base/reward.py:184-259, utilities/path_tools.py:298-448) ----
// reward_provider.reward(state) + .done(state) for n (pose, provider state) pairs; pose i is scored against the path of
// env i % n_envs (its current pool entry in geometry-pool mode) with the very device functions the step kernels use.
__global__ void reward_kernel(const StepStatic* __restrict__ S, const double* __restrict__ poses, int64_t n,
                              double* __restrict__ min_dist_io, int32_t* __restrict__ target_io,
                              const uint8_t* __restrict__ collided, double* __restrict__ reward, uint8_t* __restrict__ goal)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const DevParams& P = S->P;
    const int64_t e = i % S->n;
    const int64_t g = S->path.shared ? 0 : (S->geom_of_env ? (int64_t)S->geom_of_env[e] : e);
    const double* pts = S->path.pts + g * (int64_t)S->path.max_len * 5;
    const int m = S->path.shared ? S->path.max_len : S->path.lens[g];
    const double x = poses[3 * i], y = poses[3 * i + 1], th = poses[3 * i + 2];
    double min_dist = min_dist_io[i];
    int target = target_io[i];
    double rew;
    bool reached;
    if (P.reward_provider == BCP_REWARD_PURE_PURSUIT) {
        rew = reward_pure_pursuit(pts, m, x, y, collided && collided[i], min_dist, target);
        reached = hypot(pts[5 * (m - 1)] - x, pts[5 * (m - 1) + 1] - y) < 1.0;   // reward.py:141-150
    } else {
        const PathWindow w = path_window_of(P, S->path.shared != 0, S->path.bbox, S->path.index, g, x, y);
        rew = reward_step(P, pts, w, m, x, y, th, min_dist, target);
        reached = target > m - 1;                                                 // reward.py:66-69
    }
    min_dist_io[i] = min_dist;
    target_io[i] = target;
    reward[i] = rew;
    if (goal) goal[i] = (uint8_t)reached;
}

// find_last_reached(pose, path, spatial_precision, angular_precision) (path_tools.py:432-448): index of the LAST way
// point of the whole path the pose has reached, -1 for None
__global__ void find_last_reached_kernel(const StepStatic* __restrict__ S, const double* __restrict__ poses, int64_t n,
                                         int32_t* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t e = i % S->n;
    const int64_t g = S->path.shared ? 0 : (S->geom_of_env ? (int64_t)S->geom_of_env[e] : e);
    const double* pts = S->path.pts + g * (int64_t)S->path.max_len * 5;
    const int m = S->path.shared ? S->path.max_len : S->path.lens[g];
    const double x = poses[3 * i], y = poses[3 * i + 1], th = poses[3 * i + 2];
    const PathWindow w = path_window_of(S->P, S->path.shared != 0, S->path.bbox, S->path.index, g, x, y);
    out[i] = last_reached_from(S->P, pts, w, m, 0, x, y, th);
}

// path_velocity(path) (path_tools.py:298-323) for an n-row (t, x, y, angle) path: row j of the output belongs to the
// segment j -> j + 1.  err: BCP_ERR_ANGLE_JUMP where the reference raises, BCP_ERR_TIME_ORDER where its assert fires.
__global__ void path_velocity_kernel(const double* __restrict__ path, int64_t n, double* __restrict__ v,
                                     double* __restrict__ w, int32_t* __restrict__ err)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n - 1) return;
    const double* a = path + 4 * j;
    const double* b = a + 4;
    const double dt = b[0] - a[0];
    Pose p0 = {a[1], a[2], a[3]}, p1 = {b[1], b[2], b[3]};
    double vv, ww;
    int e = path_velocity(p0, p1, dt, vv, ww);
    if (!(dt > 0)) e |= BCP_ERR_TIME_ORDER;
    v[j] = vv;
    w[j] = ww;
    if (err) err[j] = e;
}

// is_footprint_colliding_impl(image_slice, blit_mask, lethal) (costmap_utils.py:106-136): any(image_slice[blit_mask] ==
// lethal) for n (slice, mask) pairs of one shape; one wavefront per pair, 4 cells per lane and load, wave-wide OR.
__global__ void __launch_bounds__(256) footprint_colliding_kernel(const uint8_t* __restrict__ slices,
                                                                  const uint8_t* __restrict__ masks, int64_t n,
                                                                  int64_t cells, uint32_t lethal, uint8_t* __restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const uint8_t* s = slices + i * cells;
    const uint8_t* k = masks + i * cells;
    // the pair's first byte is only byte aligned: peel up to the first 4-byte boundary of BOTH arrays when they agree,
    // otherwise go byte by byte (n * cells is rarely worth more)
    bool hit = false;
    const bool words = (((uintptr_t)s | (uintptr_t)k) & 3) == 0;
    const int64_t n4 = words ? cells / 4 : 0;
    const uint32_t l4 = lethal * 0x01010101u;
    for (int64_t q = lane; q < n4 && !hit; q += 64) {
        const uint32_t sv = reinterpret_cast<const uint32_t*>(s)[q], kv = reinterpret_cast<const uint32_t*>(k)[q];
        const uint32_t x = sv ^ l4;   // a zero byte <=> the cell is lethal
#pragma unroll
        for (int b = 0; b < 4; ++b) hit |= ((x >> (8 * b)) & 0xFFu) == 0 && ((kv >> (8 * b)) & 0xFFu) != 0;
    }
    for (int64_t q = 4 * n4 + lane; q < cells && !hit; q += 64) hit |= s[q] == lethal && k[q] != 0;
    hit = __any(hit);
    if (lane == 0) out[i] = (uint8_t)hit;
}


// the standard normals the step kernels draw for (seed, global env index, step counter): introspection of the noise stream
__global__ void device_normals_kernel(uint64_t seed, int64_t env_id_base, int64_t n, uint64_t step0, int32_t n_steps,
                                      double* __restrict__ out)
{
    const int64_t total = n * n_steps;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = it / n, i = it % n;
        double z[3];
        device_normals(seed, (uint64_t)(env_id_base + i), step0 + (uint64_t)k, z);
        out[3 * it + 0] = z[0];
        out[3 * it + 1] = z[1];
        out[3 * it + 2] = z[2];
    }
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

// robot_footprint / map_resolution (path_tools.py:145), divided on the host in fp64, and its bounding box
static void scale_footprint(DevParams& d, const bcp_params& p, double res)
{
    d.qbox[0] = d.qbox[2] = 1e30f;
    d.qbox[1] = d.qbox[3] = -1e30f;
    for (int k = 0; k < p.n_verts; ++k) {
        d.qverts[k][0] = p.verts[k][0] / res;
        d.qverts[k][1] = p.verts[k][1] / res;
        d.qbox[0] = std::min(d.qbox[0], (float)d.qverts[k][0]);
        d.qbox[1] = std::max(d.qbox[1], (float)d.qverts[k][0]);
        d.qbox[2] = std::min(d.qbox[2], (float)d.qverts[k][1]);
        d.qbox[3] = std::max(d.qbox[3], (float)d.qverts[k][1]);
    }
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
    d.inv_dt = 1.0 / p.dt;                       // (correctly rounded: what div_by_const needs)
    d.inv_L = 1.0 / p.front_wheel_from_axis;
    for (int k = 0; k < 6; ++k) d.alpha[k] = p.alpha[k];
    d.sp = p.spatial_precision;
    d.ap = p.angular_precision;
    d.progress_mult = p.spatial_progress_multiplier;
    d.par_thr = -p.spatial_precision / 9;
    d.ap_cos_min = p.angular_precision >= 3.14159265358979 ? -2.0f : (float)(std::cos(p.angular_precision) - 1e-4);
    d.sp_prune = std::nextafter(std::nextafter(p.spatial_precision, INFINITY), INFINITY);
    d.sp2_lo = p.spatial_precision * p.spatial_precision * (1.0 - 1e-13);
    d.sp2_hi = p.spatial_precision * p.spatial_precision * (1.0 + 1e-13);
    d.reward_provider = p.reward_provider;
    d.control_delay = p.control_delay;
    d.pose_delay = p.pose_delay;
    d.state_delay = p.state_delay;
    const double res = h->resolution > 0 ? h->resolution : 1.0;
    scale_footprint(d, p, res);
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
    {   // div_by_const (bcp_device.h) divides by dt and by the wheel base through their reciprocals; the sequence is the IEEE
        // quotient for every divisor but those whose significand is all ones (0.99999999999999989 and its like)
        const auto all_ones = [](double d) {
            uint64_t bits;
            memcpy(&bits, &d, sizeof(bits));
            return (bits & 0x000FFFFFFFFFFFFFull) == 0x000FFFFFFFFFFFFFull;
        };
        if (all_ones(params->dt) || (params->model == BCP_MODEL_TRICYCLE && all_ones(params->front_wheel_from_axis)))
            return fail(BCP_E_INVALID, "bcp_create: dt / front_wheel_from_axis with an all-ones significand is not supported");
    }
    if (params->model == BCP_MODEL_DIFFDRIVE && params->noise_on && !(params->options & BCP_OPT_DIFFDRIVE_NOISE))
        return fail(BCP_E_INVALID, "bcp_create: the reference's DiffDriveRobot raises IndexError with noise_parameters "
                                   "(differential_drive.py:73); set BCP_OPT_DIFFDRIVE_NOISE in bcp_params.options to opt in to "
                                   "the unpinned 1-pose analogue");
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
    h->edt_in_lds = 1;
    h->near_dilate = 1;
    h->device = device;
    h->env_id_base = env_id_base;
    h->seed = 0;
    h->exact_mode = 0;
    h->dense_threshold = 6;
    h->adaptive = 1;
    h->cull_enabled = 1;
    h->defer = 1;
    h->fused = 1;
    h->ego_sparse = 1;
    h->ego_cells_max = -1;
    h->static_dirty = true;
    h->near_shift = -1;
    if (const char* e = getenv("BCP_NEAR_SHIFT")) {   // (default of BCP_TUNE_NEAR_SHIFT for every handle of the process)
        const int v = atoi(e);
        if (v >= 0 && v <= 3) h->near_shift = v;
    }
    if (const char* e = getenv("BCP_LOCAL_PAIRS")) {   // (default of BCP_TUNE_LOCAL_PAIRS for every handle of the process)
        const int v = atoi(e);
        if (v == 1 || v == 2 || v == 4) h->local_pairs = v;
    }
    fill_dev_params(h);
    if (hipMalloc((void**)&h->tick, kTickWords * sizeof(uint64_t)) != hipSuccess || hipMemset(h->tick, 0, kTickWords * sizeof(uint64_t)) != hipSuccess) {
        if (h->tick) (void)hipFree(h->tick);
        delete h;
        return fail(BCP_E_HIP, "bcp_create: cannot allocate device memory");
    }
    *out = h;
    return BCP_OK;
}

extern "C" int bcp_destroy(bcp_handle* h)
{
    if (!h) return BCP_OK;
    (void)hipSetDevice(h->device);
    if (h->bitmap) (void)hipFree(h->bitmap);
    if (h->map_tiles) (void)hipFree(h->map_tiles);
    if (h->near_coarse) (void)hipFree(h->near_coarse);
    if (h->path5) (void)hipFree(h->path5);
    if (h->path_pre) (void)hipFree(h->path_pre);
    if (h->path_bbox) (void)hipFree(h->path_bbox);
    if (h->path_index) (void)hipFree(h->path_index);
    if (h->edt) (void)hipFree(h->edt);
    if (h->edt_col) (void)hipFree(h->edt_col);
    if (h->near) (void)hipFree(h->near);
    if (h->edt_stale) (void)hipFree(h->edt_stale);
    if (h->edt_stale_list) (void)hipFree(h->edt_stale_list);
    if (h->pending) (void)hipFree(h->pending);
    if (h->tick) (void)hipFree(h->tick);
    if (h->pending_count) (void)hipFree(h->pending_count);
    if (h->adapt) (void)hipFree(h->adapt);
    if (h->dev_static) (void)hipFree(h->dev_static);
    if (h->ego_bins) (void)hipFree(h->ego_bins);
    if (h->ego_order) (void)hipFree(h->ego_order);
    if (h->side_stream) (void)hipStreamDestroy(h->side_stream);
    if (h->ego_cells) (void)hipFree(h->ego_cells);
    if (h->ego_cell_counts) (void)hipFree(h->ego_cell_counts);
    if (h->parked_slots) (void)hipFree(h->parked_slots);
    if (h->refresh_done) (void)hipEventDestroy(h->refresh_done);
    if (h->waits_event) (void)hipEventDestroy(h->waits_event);
    if (h->waits_host) (void)hipHostFree(h->waits_host);
    if (h->ring) (void)hipFree(h->ring);
    delete h;
    return BCP_OK;
}

extern "C" int bcp_seed(bcp_handle* h, uint64_t seed)
{
    if (!h) return fail(BCP_E_INVALID, "bcp_seed: null handle");
    h->seed = seed;
    // The noise stream restarts: step counter 0 again.  The alternating counter sets of the parking scheme are keyed
    // by the counter's parity, so they are re-armed with it (rare call: synchronous).
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipDeviceSynchronize());
    const uint64_t tick[4] = {0, 0, seed, 0};
    HIP_TRY(hipMemcpy(h->tick, tick, sizeof(tick), hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(h->tick + kTickLocalTicket, 0, sizeof(uint64_t)));
    if (h->pending_count) HIP_TRY(hipMemset(h->pending_count, 0, 2 * kShards * sizeof(int32_t)));
    if (h->adapt) {
        HIP_TRY(hipMemset(h->adapt, 0, (2 + 2 * kShards) * sizeof(int32_t)));
        const int32_t init[2] = {h->dense_threshold, h->dense_threshold};
        HIP_TRY(hipMemcpy(h->adapt, init, sizeof(init), hipMemcpyHostToDevice));
    }
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
            if (value < 0 || value > 3) return fail(BCP_E_INVALID, "bcp_set_tuning: exact mode must be 0, 1, 2 or 3");
            h->exact_mode = value;
            return BCP_OK;
        case BCP_TUNE_DENSE_THRESHOLD:
            h->dense_threshold = value;
            h->adaptive = 0;   // an explicit threshold is taken as is
            return BCP_OK;
        case BCP_TUNE_DEFER:
            h->defer = value ? 1 : 0;
            return BCP_OK;
        case BCP_TUNE_EDT_LDS:
            h->edt_in_lds = value ? 1 : 0;
            return BCP_OK;
        case BCP_TUNE_NEAR_DILATE:
            if (value < 0 || value > 2) return fail(BCP_E_INVALID, "bcp_set_tuning: BCP_TUNE_NEAR_DILATE takes 0, 1 or 2");
            h->near_dilate = value;
            return BCP_OK;
        case BCP_TUNE_EGO_SPARSE:
            if (value < 0) return fail(BCP_E_INVALID, "bcp_set_tuning: BCP_TUNE_EGO_SPARSE takes 0, 1 or a limit of cells per map");
            if (value != h->ego_sparse) h->ego_cells_built = false;   // (the lists are sized for the limit in force)
            h->ego_sparse = value;
            return BCP_OK;
        case BCP_TUNE_EGO_LIST_STRIDE:
            if (value < 0 || (value & 63)) return fail(BCP_E_INVALID, "bcp_set_tuning: BCP_TUNE_EGO_LIST_STRIDE takes 0 or a multiple of 64");
            if (value != h->ego_stride) h->ego_cells_built = false;
            h->ego_stride = value;
            return BCP_OK;
        case BCP_TUNE_FUSED:
            h->fused = value ? 1 : 0;
            return BCP_OK;
        case BCP_TUNE_NEAR_SHIFT:
            if (value < -1 || value > 3) return fail(BCP_E_INVALID, "bcp_set_tuning: BCP_TUNE_NEAR_SHIFT takes -1 (default), 0, 1, 2 or 3");
            h->near_shift = value;   // (in force from the next bcp_set_costmaps on)
            return BCP_OK;
        case BCP_TUNE_LOCAL_PAIRS:
            if (value != 0 && value != 1 && value != 2 && value != 4)
                return fail(BCP_E_INVALID, "bcp_set_tuning: BCP_TUNE_LOCAL_PAIRS takes 0 (default), 1, 2 or 4");
            h->local_pairs = value;
            return BCP_OK;
        case BCP_TUNE_CULL:
            h->cull_enabled = value ? 1 : 0;
            h->cull.on = (value && h->cull.edt) ? 1 : 0;
            return BCP_OK;
        default:
            return fail(BCP_E_INVALID, "bcp_set_tuning: unknown key %d", key);
    }
}

// Derived map data (1-bit lethal mask, distance field) and path data (cos/sin columns, bounding boxes, bucket index)
// of the selected entries; `max_entries` bounds sel.size() and only sizes the grids.
static void launch_ego_cells(bcp_handle* h, EntrySelect sel, int64_t max_entries, hipStream_t s)
{
    const MapDesc& m = h->map;
    hipLaunchKernelGGL(ego_cells_kernel, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(max_entries, 8192))), dim3(256), 0, s,
                       h->map_data, sel, m.rows, m.cols, h->map_valid_rows, h->map_valid_cols, h->ego_cell_cap, h->ego_cells,
                       h->ego_cell_counts, h->ego_cell_counts + h->ego_cells_entries);
    h->ego_cells_max = -1;
}

static void launch_pack_bitmap(bcp_handle* h, EntrySelect sel, int64_t max_entries, hipStream_t s)
{
    const MapDesc& m = h->map;
    hipLaunchKernelGGL(pack_bitmap_kernel, dim3(stride_grid(max_entries * m.rows * m.wpr, 256, sel.list != nullptr)), dim3(256), 0, s, h->map_data,
                       h->bitmap, h->map_tiles, sel, m.rows, m.cols, m.wpr, h->map_valid_rows, h->map_valid_cols);
    // the cell lists of the sparse egocentric views follow the maps: all of them are rebuilt lazily after a re-bind
    // (sel.list == nullptr), the re-sampled entries of a pool refresh right here, in stream order
    if (h->ego_cells_built) {
        if (sel.list && h->ego_cell_counts) launch_ego_cells(h, sel, max_entries, s);
        else h->ego_cells_built = false;
    }
}

static void launch_near_tiles(bcp_handle* h, EntrySelect sel, int64_t max_entries, hipStream_t s)
{
    const CullDesc& C = h->cull;
    if (!C.near) return;
    const int tiles_y = C.near_words / (32 * C.near_tx);
    hipLaunchKernelGGL(near_tiles_kernel, dim3(stride_grid(max_entries * C.near_words, 256, sel.list != nullptr)), dim3(256), 0, s,
                       h->edt, sel, C.width, C.height, C.near_tx, tiles_y, C.t_out, h->near);
}

// near_dilate_kernel serves these maps: radius within a word, rows + margins in LDS
static size_t near_dilate_lds(const bcp_handle* h)
{
    const CullDesc& C = h->cull;
    if (!C.near || C.t_out < 1 || C.t_out > 32 || C.pad < C.t_out - 1) return 0;
    const int tiles_y = C.near_words / (32 * C.near_tx);
    const size_t bytes = ((size_t)(tiles_y * 32 + 2 * (C.t_out - 1)) * (C.near_tx + 2) + C.t_out) * sizeof(uint32_t);
    return bytes <= 64 * 1024 ? bytes : 0;
}

static void launch_near_dilate(bcp_handle* h, EntrySelect sel, int64_t max_entries, uint8_t* stale, hipStream_t s)
{
    const MapDesc& m = h->map;
    const CullDesc& C = h->cull;
    const size_t lds = near_dilate_lds(h);
    const int64_t blocks = std::min<int64_t>(max_entries, sel.list ? 4096 : 16384);
    hipLaunchKernelGGL(near_dilate_kernel, dim3((unsigned)std::max<int64_t>(blocks, 1)), dim3(256), lds, s, h->bitmap, sel, m.rows,
                       m.cols, m.wpr, C.pad, C.t_out, C.width, C.height, C.near_tx, C.near_words / (32 * C.near_tx), h->near, stale);
}

static void launch_edt(bcp_handle* h, EntrySelect sel, int64_t max_entries, hipStream_t s)
{
    const MapDesc& m = h->map;
    const CullDesc& C = h->cull;
    const size_t lds = ((size_t)m.rows * m.wpr + (size_t)C.height * ((C.width + 3) / 4)) * sizeof(uint32_t) +
                       (((size_t)C.clamp * C.clamp + 1 + 3) & ~(size_t)3);
    // (one workgroup per map: worth it from a few dozen maps on; a lone shared map keeps the two wide kernels)
    if (C.clamp <= 60 && lds <= kMaxDynamicLds && h->edt_in_lds && max_entries >= 32) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(edt_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        const int64_t blocks = std::min<int64_t>(max_entries, sel.list ? 2048 : 16384);
        hipLaunchKernelGGL(edt_lds_kernel, dim3((unsigned)std::max<int64_t>(blocks, 1)), dim3(256), lds, s, h->bitmap, sel,
                           m.rows, m.cols, m.wpr, C.pad, C.clamp, h->edt);
        return;
    }
    hipLaunchKernelGGL(edt_columns_kernel, dim3(stride_grid(max_entries * C.width, 64, sel.list != nullptr)), dim3(64), 0, s, h->bitmap, sel, m.rows,
                       m.cols, m.wpr, C.pad, C.clamp, h->edt_col);
    hipLaunchKernelGGL(edt_rows_kernel, dim3(stride_grid(max_entries * C.width * C.height, 256, sel.list != nullptr)), dim3(256), 0, s, h->edt_col,
                       sel, C.width, C.height, C.clamp, h->edt);
}

// CullDesc::step_near of private maps unless BCP_TUNE_NEAR_SHIFT says otherwise: a quarter of the resolution (measured on one
// box, shift 0 / 1 / 2: one private 64 x 64 world per env 821 / 751 / 719 bytes of memory traffic per env-step and 21.1 / 20.6 /
// 20.6 us per step; 65 536 private 256 x 141 aisle maps 2.58 / 2.53 / 2.60e9 env-steps/s -- profiles/r04_near_shift.txt;
// shift 3, an eighth: 687 bytes, but 19.6 against 19.1 us and the aisle maps 2.24e9 -- more poses go to the exact test)
constexpr int kNearShiftPrivate = 2;

static void launch_near_coarse(bcp_handle* h, EntrySelect sel, int64_t max_entries, hipStream_t s)
{
    const CullDesc& C = h->cull;
    if (!C.near || C.step_near_shift == 0) return;
    const int tiles_y = C.near_words / (32 * C.near_tx);
    const int cty = (int)(C.step_near_stride / (32 * C.step_near_tx));   // (private maps only: the stride is an entry's words)
    hipLaunchKernelGGL(near_coarsen_kernel, dim3(stride_grid(max_entries * C.step_near_tx * cty * 32, 256, sel.list != nullptr)), dim3(256),
                       0, s, h->near, sel, C.near_tx, tiles_y, C.step_near_shift, C.step_near_tx, cty, h->near_coarse);
}

// Distance field + tiles of the selected entries.  `tiles_only`: the caller's consumers read nothing but the tiles (a pool
// refresh under the single-launch step) -- when near_dilate_kernel can serve the maps, the uint8 field is left stale and
// marked so; ensure_fields() brings it up to date for whoever asks for it later.
static int launch_distance_field(bcp_handle* h, EntrySelect sel, int64_t max_entries, hipStream_t s, bool tiles_only = false)
{
    if (tiles_only && h->near_dilate >= 1 && near_dilate_lds(h) && h->edt_stale && h->edt_stale_cap >= n_slots(h)) {
        launch_near_dilate(h, sel, max_entries, h->edt_stale, s);
        launch_near_coarse(h, sel, max_entries, s);
        h->edt_lazy = true;
        return BCP_OK;
    }
    launch_edt(h, sel, max_entries, s);
    launch_near_tiles(h, sel, max_entries, s);
    if (h->near_dilate == 2 && near_dilate_lds(h)) launch_near_dilate(h, sel, max_entries, nullptr, s);
    launch_near_coarse(h, sel, max_entries, s);
    return BCP_OK;
}

// (One scratch list per handle: the readers of the uint8 field of ONE handle must share a stream, like everything else a
// handle does -- include/bcplan.h, "a handle is not thread-safe".)
// Before anything reads the uint8 field (two-launch and single-kernel step forms, bcp_pose_collides,
// bcp_get_distance_field): the transform of the entries a tiles-only refresh has left stale, on the reader's stream.  An
// entry is marked at the end of its refresh, in the refresh's stream order, so a refresh still running on another stream
// is simply picked up by the next call; the flag of the handle stays up for as long as such refreshes may be in flight.
static int ensure_fields(bcp_handle* h, hipStream_t s)
{
    if (!h->edt_lazy || !h->cull.edt || !h->edt_stale) return BCP_OK;
    // Has every tiles-only refresh issued so far finished?  Then this pass leaves no stale field behind and later calls can
    // skip their three launches until the next such refresh (which raises the flag again).
    bool settled = true;   // (no refresh ever issued: the stale marks come from bcp_set_costmaps, in stream order)
    if (h->refresh_recorded) {
        settled = hipEventQuery(h->refresh_done) == hipSuccess;
        if (!settled) (void)hipGetLastError();   // (hipErrorNotReady)
    }
    const int64_t entries = h->edt_stale_cap;
    int32_t* count = h->edt_stale_list + entries;
    HIP_TRY(hipMemsetAsync(count, 0, sizeof(int32_t), s));
    hipLaunchKernelGGL(stale_fields_list_kernel, dim3(stride_grid(entries, 256)), dim3(256), 0, s, h->edt_stale, entries,
                       h->edt_stale_list, count);
    const EntrySelect sel = {h->edt_stale_list, count, entries};
    launch_edt(h, sel, entries, s);
    HIP_TRY(hipGetLastError());
    if (settled) h->edt_lazy = false;
    return BCP_OK;
}

// the costmap origins into the path records of private paths (kBoxOrigin): needs both the costmaps and the paths
static void launch_world_records(bcp_handle* h, EntrySelect sel, int64_t max_entries, hipStream_t s)
{
    if (!h->path_bbox || h->path.shared || !h->map.bits) return;
    hipLaunchKernelGGL(world_record_kernel, dim3(stride_grid(max_entries, 256, sel.list != nullptr)), dim3(256), 0, s, sel,
                       h->map.origins, h->map.ox, h->map.oy, h->path_bbox);
}

static void launch_path_data(bcp_handle* h, EntrySelect sel, int64_t max_entries, hipStream_t s)
{
    const PathDesc& p = h->path;
    hipLaunchKernelGGL(path_bbox_kernel, dim3(stride_grid(max_entries, 256, sel.list != nullptr)), dim3(256), 0, s, h->path_src, p.lens, p.max_len,
                       sel, h->dev.sp_prune, p.shared ? kPathBuckets : kPathBucketsCompact, h->path_bbox);
    hipLaunchKernelGGL(path_trig_kernel, dim3(stride_grid(max_entries * p.max_len, 256, sel.list != nullptr)), dim3(256), 0, s, h->path_src, h->path5,
                       p.shared ? nullptr : h->path_pre, h->path_bbox, sel, p.max_len);
    if (p.shared)
        hipLaunchKernelGGL(path_index_kernel, dim3(stride_grid(max_entries * 2 * kPathBuckets, 256, sel.list != nullptr)), dim3(256), 0, s, h->path_src,
                           p.lens, p.max_len, sel, h->dev.sp_prune, h->path_bbox, h->path_index);
    else   // (private paths: compact tables inside the records)
        hipLaunchKernelGGL(path_index_compact_kernel, dim3(stride_grid(max_entries * 2 * kPathBucketsCompact, 256, sel.list != nullptr)), dim3(256), 0,
                           s, h->path_src, p.lens, p.max_len, sel, h->dev.sp_prune, h->path_bbox);
    launch_world_records(h, sel, max_entries, s);
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
    const size_t tile_bytes = (size_t)n_maps * map_tile_words(rows, wpr) * sizeof(uint32_t);
    if (tile_bytes > h->map_tiles_bytes) {
        if (h->map_tiles) HIP_TRY(hipFree(h->map_tiles));
        h->map_tiles = nullptr;
        h->map_tiles_bytes = 0;
        HIP_TRY(hipMalloc((void**)&h->map_tiles, tile_bytes));
        h->map_tiles_bytes = tile_bytes;
    }
    h->resolution = resolution;
    h->map_data = data;
    h->map_valid_rows = valid_rows;
    h->map_valid_cols = valid_cols;
    fill_dev_params(h);
    MapDesc& m = h->map;
    m.bits = h->bitmap;
    m.tiles = h->map_tiles;
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
    const EntrySelect all_maps = {nullptr, nullptr, n_maps};
    launch_pack_bitmap(h, all_maps, n_maps, s);
    HIP_TRY(hipGetLastError());
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
        // the 1-bit form for the outer test (near_tiles_kernel)
        const int tiles_x = (W + 31) / 32, tiles_y = (H + 31) / 32;
        const size_t near_bytes = (size_t)n_maps * tiles_x * tiles_y * 32 * sizeof(uint32_t);
        if (near_bytes > h->near_bytes) {
            if (h->near) HIP_TRY(hipFree(h->near));
            h->near = nullptr;
            h->near_bytes = 0;
            HIP_TRY(hipMalloc((void**)&h->near, near_bytes));
            h->near_bytes = near_bytes;
        }
        C.near = h->near;
        C.near_tx = tiles_x;
        C.near_words = tiles_x * tiles_y * 32;
        C.near_stride = shared ? 0 : (int64_t)C.near_words;
        // what the single-launch step reads: the tiles themselves for a shared map (it stays in cache), a coarser copy for
        // private maps -- see CullDesc::step_near
        const int shift = shared ? 0 : (h->near_shift >= 0 ? h->near_shift : kNearShiftPrivate);
        C.step_near = h->near;
        C.step_near_stride = C.near_stride;
        C.step_near_tx = tiles_x;
        C.step_near_shift = 0;
        if (shift > 0) {
            const int cw = (W + (1 << shift) - 1) >> shift, ch = (H + (1 << shift) - 1) >> shift;
            const int ctx = (cw + 31) / 32, cty = (ch + 31) / 32;
            const size_t coarse_bytes = (size_t)n_maps * ctx * cty * 32 * sizeof(uint32_t);
            if (coarse_bytes > h->near_coarse_bytes) {
                if (h->near_coarse) HIP_TRY(hipFree(h->near_coarse));
                h->near_coarse = nullptr;
                h->near_coarse_bytes = 0;
                HIP_TRY(hipMalloc((void**)&h->near_coarse, coarse_bytes));
                h->near_coarse_bytes = coarse_bytes;
            }
            C.step_near = h->near_coarse;
            C.step_near_stride = (int64_t)ctx * cty * 32;
            C.step_near_tx = ctx;
            C.step_near_shift = shift;
        }
        C.edt = h->edt;
        C.width = W;
        C.height = H;
        C.clamp = clamp;
        C.env_stride = shared ? 0 : (int64_t)W * H;
        C.on = C.t_out <= clamp ? 1 : 0;
        // the stale marks of tiles-only refreshes (launch_distance_field): every field is rebuilt below, so none is stale
        h->edt_lazy = false;
        if (!shared && n_maps > h->edt_stale_cap) {
            if (h->edt_stale) HIP_TRY(hipFree(h->edt_stale));
            if (h->edt_stale_list) HIP_TRY(hipFree(h->edt_stale_list));
            h->edt_stale = nullptr;
            h->edt_stale_list = nullptr;
            h->edt_stale_cap = 0;
            HIP_TRY(hipMalloc((void**)&h->edt_stale, (size_t)n_maps));
            HIP_TRY(hipMalloc((void**)&h->edt_stale_list, (size_t)(n_maps + 1) * sizeof(int32_t)));
            h->edt_stale_cap = n_maps;
        }
        if (h->edt_stale) HIP_TRY(hipMemsetAsync(h->edt_stale, 0, (size_t)h->edt_stale_cap, s));
        {
            // Many private maps under the single-launch step: only the 1-bit tiles are read, so they are made directly from the
            // lethal masks (near_dilate_kernel) and the uint8 fields are left to whoever asks for them (ensure_fields) -- what a
            // pool refresh has done since round 3.  65 536 maps of 256 x 256: 75 ms of edt_lds_kernel -> a few ms (round 4).
            const bool tiles_only = !shared && n_maps >= 32 && h->fused && h->adaptive && C.on && h->near_dilate == 1;
            const int rc = launch_distance_field(h, all_maps, n_maps, s, tiles_only);
            if (rc != BCP_OK) return rc;
        }
        HIP_TRY(hipGetLastError());
        if (!h->pending) {
            const int64_t blocks = (h->n + kBlock - 1) / kBlock;
            h->pending_cap = (int32_t)(((blocks + kShards - 1) / kShards) * kBlock);  // every env of a shard's blocks
            HIP_TRY(hipMalloc(&h->pending, (size_t)kShards * h->pending_cap * sizeof(Pending)));
            HIP_TRY(hipMalloc((void**)&h->pending_count, 2 * kShards * sizeof(int32_t)));
            HIP_TRY(hipMemsetAsync(h->pending_count, 0, 2 * kShards * sizeof(int32_t), s));
            // [2] thresholds (alternating by step parity), then [2][kShards] in-place counters
            HIP_TRY(hipMalloc((void**)&h->adapt, (2 + 2 * kShards) * sizeof(int32_t)));
            HIP_TRY(hipMemsetAsync(h->adapt, 0, (2 + 2 * kShards) * sizeof(int32_t), s));
            const int32_t init[2] = {h->dense_threshold, h->dense_threshold};
            HIP_TRY(hipMemcpyAsync(h->adapt, init, sizeof(init), hipMemcpyHostToDevice, s));
            HIP_TRY(hipStreamSynchronize(s));   // (`init` is on the stack)
        }
    }
    h->have_map = true;
    h->static_dirty = true;
    if (h->have_path && !h->path.shared) {   // the path records carry the origins
        const int64_t n_paths = n_slots(h);
        const EntrySelect all_paths = {nullptr, nullptr, n_paths};
        launch_world_records(h, all_paths, n_paths, s);
        HIP_TRY(hipGetLastError());
    }
    return BCP_OK;
}

extern "C" int bcp_get_distance_field(bcp_handle* h, int64_t first_entry, int64_t n_entries, uint8_t* out, int32_t* shape,
                                      void* stream)
{
    if (!h || !shape) return fail(BCP_E_INVALID, "bcp_get_distance_field: null argument");
    if (!h->have_map || !h->cull.edt) return fail(BCP_E_STATE, "bcp_get_distance_field: no distance field (no costmap, or culling off)");
    const CullDesc& C = h->cull;
    shape[0] = C.height;
    shape[1] = C.width;
    shape[2] = C.pad;
    shape[3] = C.clamp;
    if (!out) return BCP_OK;
    const int64_t n_maps = h->map.shared ? 1 : n_slots(h);
    if (first_entry < 0 || n_entries <= 0 || first_entry + n_entries > n_maps)
        return fail(BCP_E_INVALID, "bcp_get_distance_field: entries out of range");
    HIP_TRY(hipSetDevice(h->device));
    { const int rc = ensure_fields(h, (hipStream_t)stream); if (rc != BCP_OK) return rc; }
    const size_t per = (size_t)C.width * C.height;
    HIP_TRY(hipMemcpyAsync(out, h->edt + first_entry * per, n_entries * per, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return BCP_OK;
}

extern "C" int bcp_get_near_field(bcp_handle* h, int64_t first_entry, int64_t n_entries, uint32_t* out, int32_t* shape,
                                  void* stream)
{
    if (!h || !shape) return fail(BCP_E_INVALID, "bcp_get_near_field: null argument");
    if (!h->have_map || !h->cull.near) return fail(BCP_E_STATE, "bcp_get_near_field: no distance field (no costmap, or culling off)");
    const CullDesc& C = h->cull;
    shape[0] = C.near_words / (32 * C.near_tx);
    shape[1] = C.near_tx;
    shape[2] = C.t_out;
    if (!out) return BCP_OK;
    const int64_t n_maps = h->map.shared ? 1 : n_slots(h);
    if (first_entry < 0 || n_entries <= 0 || first_entry + n_entries > n_maps)
        return fail(BCP_E_INVALID, "bcp_get_near_field: entries out of range");
    HIP_TRY(hipSetDevice(h->device));
    const size_t per = (size_t)C.near_words;
    HIP_TRY(hipMemcpyAsync(out, h->near + first_entry * per, n_entries * per * sizeof(uint32_t), hipMemcpyDeviceToDevice,
                           (hipStream_t)stream));
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
    const size_t pre_bytes = shared ? 0 : (size_t)total * 2 * sizeof(uint32_t);
    if (pre_bytes > h->path_pre_bytes) {
        if (h->path_pre) HIP_TRY(hipFree(h->path_pre));
        h->path_pre = nullptr;
        h->path_pre_bytes = 0;
        HIP_TRY(hipMalloc((void**)&h->path_pre, pre_bytes));
        h->path_pre_bytes = pre_bytes;
    }
    const int64_t n_paths = shared ? 1 : n_slots(h);
    const size_t bb_bytes = (size_t)n_paths * kBoxDoubles * sizeof(double);
    if (bb_bytes > h->path_bbox_bytes) {
        if (h->path_bbox) HIP_TRY(hipFree(h->path_bbox));
        h->path_bbox = nullptr;
        h->path_bbox_bytes = 0;
        HIP_TRY(hipMalloc((void**)&h->path_bbox, bb_bytes));
        h->path_bbox_bytes = bb_bytes;
    }
    const size_t ix_bytes = (size_t)4 * kPathBuckets * sizeof(int16_t);   // (a shared path's tables; private ones live in the records)
    if (ix_bytes > h->path_index_bytes) {
        if (h->path_index) HIP_TRY(hipFree(h->path_index));
        h->path_index = nullptr;
        h->path_index_bytes = 0;
        HIP_TRY(hipMalloc((void**)&h->path_index, ix_bytes));
        h->path_index_bytes = ix_bytes;
    }
    if (max_len > 32766) return fail(BCP_E_INVALID, "bcp_set_paths: paths longer than 32766 way points are not supported");
    h->path.pts = h->path5;
    h->path.pre = shared ? nullptr : h->path_pre;
    h->path.bbox = h->path_bbox;
    h->path.index = h->path_index;
    h->path.lens = shared ? nullptr : lens;
    h->path.max_len = max_len;
    h->path.shared = shared ? 1 : 0;
    h->path_src = xytheta;
    const EntrySelect all_paths = {nullptr, nullptr, n_paths};
    launch_path_data(h, all_paths, n_paths, s);
    HIP_TRY(hipGetLastError());
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
// no delay queues and the continuous reward provider: the step kernels compile both out (their PLAIN variants)
static bool step_is_plain(const bcp_handle* h)
{
    const bcp_params& p = h->params;
    return p.control_delay == 0 && p.pose_delay == 0 && p.state_delay == 0 && p.reward_provider == BCP_REWARD_CONTINUOUS;
}

static bool step_uses_deferral(const bcp_handle* h)
{
    return h->defer && h->cull.on && h->exact_mode == 0 && h->pending != nullptr;
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
    S.dense_threshold = h->dense_threshold;   // (a negative value settles every undecided pose inside kernel 1)
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

// The parity-keyed parking / adaptation counters are only maintained by the two-kernel step (kernel 1 zeroes the NEXT
// step's set).  Whenever the step form changes (bcp_set_tuning between steps, a costmap without distance field, ...)
// both sets are re-armed on the stream of the steps, so the two-kernel step never resumes on stale counts.
static int rearm_parking(bcp_handle* h, hipStream_t s)
{
    if (h->pending_count) HIP_TRY(hipMemsetAsync(h->pending_count, 0, 2 * kShards * sizeof(int32_t), s));
    if (h->adapt) {
        HIP_TRY(hipMemsetAsync(h->adapt + 2, 0, 2 * kShards * sizeof(int32_t), s));
        HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)h->adapt, h->dense_threshold, 2, s));
    }
    return BCP_OK;
}

// step_local_kernel<WIDE, PLAIN, PAIRS, ROLL>: variant = WIDE << 1 | PLAIN; the rollout form exists for the 16-wave workgroup
static const void* local_step_fn(int variant, int pairs, bool roll = false)
{
    if (roll) {
        switch (variant) {
            case 3: return (const void*)step_local_kernel<true, true, 4, true>;
            case 2: return (const void*)step_local_kernel<true, false, 4, true>;
            case 1: return (const void*)step_local_kernel<false, true, 4, true>;
            default: return (const void*)step_local_kernel<false, false, 4, true>;
        }
    }
#define BCP_LOCAL_FN(W, P) (pairs == 4 ? (const void*)step_local_kernel<W, P, 4> : pairs == 2 ? (const void*)step_local_kernel<W, P, 2> \
                                                                                              : (const void*)step_local_kernel<W, P, 1>)
    switch (variant) {
        case 3: return BCP_LOCAL_FN(true, true);
        case 2: return BCP_LOCAL_FN(true, false);
        case 1: return BCP_LOCAL_FN(false, true);
        default: return BCP_LOCAL_FN(false, false);
    }
#undef BCP_LOCAL_FN
}

// Size of step_local_kernel's workgroups for this handle: BCP_TUNE_LOCAL_PAIRS, or (0) the default of the configuration.
static int local_pairs(const bcp_handle* h)
{
    if (h->local_pairs == 1 || h->local_pairs == 2 || h->local_pairs == 4) return h->local_pairs;
    return kLocalPairsDefault;
}

// rollout_steps > 1: only the single-launch form (step_local_kernel<.., ROLL = true>) takes several steps per launch; the
// caller (bcp_rollout) steps the other forms one launch at a time.
static int launch_step(bcp_handle* h, const bcp_step_io* io, uint32_t flags, hipStream_t s, bool first_only = false,
                       int32_t rollout_steps = 1)
{
    if (h->static_dirty) {
        const int rc = upload_step_static(h, s);
        if (rc != BCP_OK) return rc;
    }
    const StepStatic& S = h->host_static;
    // (an explicit BCP_TUNE_DENSE_THRESHOLD asks for poses to be settled inside the stepping wave: the two-launch form has that path)
    const bool fused = S.pending && h->fused && h->adaptive;
    const int32_t form = fused ? 3 : (S.pending ? 2 : 1);
    if (!fused && h->edt_lazy) {   // these forms read the uint8 field
        const int rc = ensure_fields(h, s);
        if (rc != BCP_OK) return rc;
    }
    if (form != h->last_step_form) {
        if (form == 2 && h->last_step_form != 0) {
            const int rc = rearm_parking(h, s);
            if (rc != BCP_OK) return rc;
        }
        h->last_step_form = form;
    }
    StepArgs a;
    a.S = h->dev_static;
    StepHot& hot = a.hot;
    hot.st = S.st;
    hot.n = S.n;
    hot.geom_of_env = S.geom_of_env;
    hot.path_pts = S.path.pts;
    hot.path_pre = S.path.pre;
    hot.path_bbox = S.path.bbox;
    hot.path_index = S.path.index;
    hot.pending = S.pending;
    hot.map_bits = S.map.bits;
    hot.map_env_stride = S.map.env_stride;
    hot.model = S.P.model;
    hot.lds_path_doubles = S.lds_path_doubles;
    hot.path_shared = S.path.shared;
    hot.pending_cap = S.pending_cap;
    hot.map_rows = S.map.rows;
    hot.map_cols = S.map.cols;
    hot.map_wpr = S.map.wpr;
    hot.map_shared = S.map.shared;
    hot.path_max_len = S.path.max_len;
    hot.near = S.cull.on ? S.cull.step_near : nullptr;
    hot.noise_on = S.P.noise_on;
    hot.n_verts = S.P.n_verts;
    hot.control_delay = S.P.control_delay;
    hot.pose_delay = S.P.pose_delay;
    hot.state_delay = S.P.state_delay;
    hot.dynamic_model = S.P.dynamic_model;
    hot.noise_slot0 = (S.P.alpha[0] > 0.0 || S.P.alpha[1] > 0.0) ? 1 : 0;
    hot.model_front_column_pid = S.P.model_front_column_pid;
    hot.env_id_base = S.env_id_base;
    hot.qverts = &h->dev_static->P.qverts[0][0];
    hot.map_origins = S.map.origins;
    a.actions = io->actions;
    a.noise_z = io->noise_z;
    a.noise_z_out = io->noise_z_out;
    a.reward = io->reward;
    a.done = io->done;
    a.collided_now = io->collided_now;
    a.err = io->err;
    a.flags = flags;
    // the step counter and the noise seed are read on the device (StepArgs::tick); the kernels resolve these themselves
    a.seed = a.step_counter = 0;
    a.pending_count = a.pending_next = nullptr;
    a.threshold_now = nullptr;
    a.threshold_next = a.inplace_count = a.inplace_next = nullptr;
    const bool adapt = h->adaptive && h->adapt && S.pending && S.dense_threshold >= 0;
    a.tick = h->tick;
    a.parked_slots = nullptr;
    a.map_tiles = S.map.tiles;
    a.rollout_steps = 1;
    a.pending_base = h->pending_count;
    a.adapt_base = adapt ? h->adapt : nullptr;
    const int blocks = (int)((h->n + kBlock - 1) / kBlock);
    if (fused) {
        // the whole step as one launch: 256 envs per workgroup of 16 waves; undecided poses are handed over in LDS and
        // settled by all the workgroup's waves (step_local_kernel)
        const int variant = (S.wide ? 2 : 0) | (step_is_plain(h) ? 1 : 0);
        const bool roll = rollout_steps > 1;
        const int pairs = roll ? 4 : local_pairs(h);
        const int pslot = roll ? 3 : (pairs == 4 ? 2 : (pairs == 2 ? 1 : 0));
        const void* fn = local_step_fn(variant, pairs, roll);
        a.rollout_steps = rollout_steps;
        const int64_t bitmap_words = (int64_t)S.map.rows * S.map.wpr;
        const size_t lds = local_step_lds_bytes(h->params.n_verts, S.lds_path_doubles,
                                                (S.map.shared && bitmap_words <= kLocalMapWords) ? (int)bitmap_words : 0,
                                                step_is_plain(h), pairs);
        // The attribute belongs to the FUNCTION on a device, not to a handle: the largest size any handle of this process
        // has asked for stays set (two live handles with different staging sizes would otherwise lower it under each other).
        {
            static std::mutex lds_mutex;
            static int32_t lds_max[64][4][4];   // [device][variant][workgroup size | rollout form], zero-initialised
            std::lock_guard<std::mutex> lock(lds_mutex);
            int32_t& cur = lds_max[h->device & 63][variant][pslot];
            if ((int32_t)lds > cur) {
                HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                cur = (int32_t)lds;
            }
        }
        a.flags |= kStepAdvances;
        hot.io_flags = a.flags;   // (the prologue's copies, next to the rest of what it fetches)
        hot.io_actions = a.actions;
        hot.io_noise_z = a.noise_z;
        hot.io_tick = a.tick;
        const int envs_per_group = pairs * kBlock;
        const dim3 grid((unsigned)((h->n + envs_per_group - 1) / envs_per_group)), block(4 * pairs * kBlock);
        if (!h->parked_slots) {   // (sized for the smallest workgroup: the size may change between steps)
            h->parked_cap = (h->n + kBlock - 1) / kBlock;
            HIP_TRY(hipMalloc((void**)&h->parked_slots, (size_t)h->parked_cap * sizeof(uint64_t)));
            HIP_TRY(hipMemsetAsync(h->parked_slots, 0, (size_t)h->parked_cap * sizeof(uint64_t), s));
        }
        a.parked_slots = h->parked_slots;
        void* kargs[] = {(void*)&a};
        HIP_TRY(hipLaunchKernel(fn, grid, block, kargs, lds, s));
    } else if (rollout_steps > 1) {
        return fail(BCP_E_STATE, "launch_step: only the single-launch step form takes several steps per launch");
    } else if (S.pending) {
        // kernel 1 settles every env the distance field decides; kernel 2 rasterises the parked rest
        const size_t lds1 = ((size_t)h->params.n_verts * 2 + S.lds_path_doubles) * sizeof(double);
        const size_t lds2 = (size_t)2 * 4 * (S.wide ? 8 : 3) * 64 * sizeof(uint32_t);
        const int waves = 2048;  // a multiple of kShards: 32 teams per shard, so that a shard rarely needs a second round
        const bool second = !first_only && S.dense_threshold >= 0;  // (threshold < 0: everything settled in place)
        if (!second) a.flags |= kStepAdvances;   // kernel 1 is the whole step
        // kernel 1 runs with two wavefronts per 64 envs (mover + scorer, step_fast_pair_kernel)
        const size_t lds1p = lds1 + ((size_t)6 * kBlock + 8) * sizeof(double) + 2 * kBlock * sizeof(uint32_t);
        const dim3 g1(blocks), b1(2 * kBlock), g2(waves), b2(kBlock * kPendingWaves);
        const int variant = (S.wide ? 2 : 0) | (step_is_plain(h) ? 1 : 0);
        switch (variant) {
            case 3:
                hipLaunchKernelGGL((step_fast_pair_kernel<true, true>), g1, b1, lds1p, s, a);
                if (second) hipLaunchKernelGGL((step_pending_kernel<true, true>), g2, b2, lds2, s, a);
                break;
            case 2:
                hipLaunchKernelGGL((step_fast_pair_kernel<true, false>), g1, b1, lds1p, s, a);
                if (second) hipLaunchKernelGGL((step_pending_kernel<true, false>), g2, b2, lds2, s, a);
                break;
            case 1:
                hipLaunchKernelGGL((step_fast_pair_kernel<false, true>), g1, b1, lds1p, s, a);
                if (second) hipLaunchKernelGGL((step_pending_kernel<false, true>), g2, b2, lds2, s, a);
                break;
            default:
                hipLaunchKernelGGL((step_fast_pair_kernel<false, false>), g1, b1, lds1p, s, a);
                if (second) hipLaunchKernelGGL((step_pending_kernel<false, false>), g2, b2, lds2, s, a);
                break;
        }
    } else {
        const size_t lds = collision_lds_bytes(h->params.n_verts, h->map.in_lds, h->map.rows, h->map.wpr);
        a.flags |= kStepAdvances;
        hipLaunchKernelGGL(step_kernel, dim3(blocks), dim3(kBlock), lds, s, a);
    }
    return BCP_OK;
}

// Flags a caller may pass.  The ablation switches of bcp_step.h (timing experiments, results wrong by construction)
// exist only in a -DBCP_DIAG build (tools/); kStepAdvances is internal and never accepted.
#ifdef BCP_DIAG
constexpr uint32_t kCallerFlags = BCP_STEP_AUTO_RESET | BCP_STEP_ACTIONS_F32 | kAblateNoCollision | kAblateNoReward |
                                  kAblateNoCoop | kAblateNoPark | kAblateNoClassify | kDiagWithholdVerdicts;
#else
constexpr uint32_t kCallerFlags = BCP_STEP_AUTO_RESET | BCP_STEP_ACTIONS_F32;
#endif

static int check_step(bcp_handle* h, const bcp_step_io* io, uint32_t flags, const char* who)
{
    if (!h || !io) return fail(BCP_E_INVALID, "%s: null argument", who);
    if (flags & ~kCallerFlags) return fail(BCP_E_INVALID, "%s: undefined flag bits 0x%x", who, flags & ~kCallerFlags);
    if (!h->have_map || !h->have_path || !h->have_state)
        return fail(BCP_E_STATE, "%s: costmaps, paths and state must be set first", who);
    if ((flags & BCP_STEP_AUTO_RESET) && !h->have_init)
        return fail(BCP_E_STATE, "%s: BCP_STEP_AUTO_RESET needs bcp_bind_initial_state", who);
    if (!io->actions || !io->reward || !io->done) return fail(BCP_E_INVALID, "%s: actions/reward/done are required", who);
    return BCP_OK;
}

// A wait of step_local_kernel that gives up lets its envs finish as free (bcp_step.h: BCP_ERR_INTERNAL): a training loop
// that never calls bcp_expired_waits would not notice.  So bcp_step itself looks, without ever waiting for the GPU: every
// kWatchdogSteps calls the counter is copied to pinned host memory behind the step just launched, and a later call, once
// that copy has landed, compares it with what was seen before.
constexpr uint32_t kWatchdogSteps = 256;

static int step_watchdog(bcp_handle* h, hipStream_t s)
{
    if (h->last_step_form != 3) return BCP_OK;   // (only step_local_kernel has such waits)
    if (h->waits_in_flight) {
        const hipError_t q = hipEventQuery(h->waits_event);
        if (q == hipSuccess) {
            h->waits_in_flight = false;
            const uint64_t now = *h->waits_host;
            if (now > h->waits_seen) {
                const uint64_t fresh = now - h->waits_seen;
                h->waits_seen = now;
                return fail(BCP_E_INTERNAL, "bcp_step: %llu bounded wait(s) of the step kernel gave up during earlier steps "
                                            "(BCP_ERR_INTERNAL in err[] marks the envs; their verdicts are unreliable)",
                            (unsigned long long)fresh);
            }
        } else {
            (void)hipGetLastError();   // hipErrorNotReady is not an error here
        }
        return BCP_OK;
    }
    if (++h->steps_since_probe < kWatchdogSteps) return BCP_OK;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return BCP_OK;   // (a captured step is replayed without this function: the caller asks bcp_expired_waits)
    }
    if (!h->waits_host) {
        HIP_TRY(hipHostMalloc((void**)&h->waits_host, sizeof(uint64_t), hipHostMallocDefault));
        *h->waits_host = 0;
        HIP_TRY(hipEventCreateWithFlags(&h->waits_event, hipEventDisableTiming));
    }
    HIP_TRY(hipMemcpyAsync(h->waits_host, h->tick + 4, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipEventRecord(h->waits_event, s));
    h->waits_in_flight = true;
    h->steps_since_probe = 0;
    return BCP_OK;
}

extern "C" int bcp_step(bcp_handle* h, const bcp_step_io* io, uint32_t flags, void* stream)
{
    int rc = check_step(h, io, flags, "bcp_step");
    if (rc != BCP_OK) return rc;
    HIP_TRY(hipSetDevice(h->device));
    rc = launch_step(h, io, flags, (hipStream_t)stream);
    if (rc != BCP_OK) return rc;
    HIP_TRY(hipGetLastError());
    return step_watchdog(h, (hipStream_t)stream);
}

// K steps per call for callers that hold the actions of a whole rollout (Monte-Carlo rollouts from one state, the use the
// reference documents: /root/reference/README.md "many rollouts from one state"; StepEnvRoller's 128-step rollouts once the
// policy is open-loop).  With the single-launch step form the K steps are ONE launch of step_local_kernel<.., ROLL = true>;
// otherwise K launches.  Either way: the states and outputs of K calls of bcp_step with row k of the arrays, bit for bit.
extern "C" int bcp_rollout(bcp_handle* h, const bcp_step_io* io, int32_t n_steps, uint32_t flags, void* stream)
{
    int rc = check_step(h, io, flags, "bcp_rollout");
    if (rc != BCP_OK) return rc;
    if (n_steps <= 0) return fail(BCP_E_INVALID, "bcp_rollout: n_steps must be positive");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    if (h->static_dirty) {
        rc = upload_step_static(h, s);
        if (rc != BCP_OK) return rc;
    }
    const StepStatic& S = h->host_static;
    const bool fused = S.pending && h->fused && h->adaptive;
    if (fused && n_steps > 1) {
        rc = launch_step(h, io, flags, s, false, n_steps);
        if (rc != BCP_OK) return rc;
        HIP_TRY(hipGetLastError());
        return BCP_OK;
    }
    const int64_t n = h->n;
    const size_t act = (flags & BCP_STEP_ACTIONS_F32) ? 8 : 16;
    for (int32_t k = 0; k < n_steps; ++k) {
        bcp_step_io row = *io;
        row.actions = (const char*)io->actions + (size_t)k * n * act;
        if (io->noise_z) row.noise_z = io->noise_z + (size_t)k * n * 3;
        if (io->noise_z_out) row.noise_z_out = io->noise_z_out + (size_t)k * n * 3;
        row.reward = io->reward + (size_t)k * n;
        row.done = io->done + (size_t)k * n;
        if (io->collided_now) row.collided_now = io->collided_now + (size_t)k * n;
        if (io->err) row.err = io->err + (size_t)k * n;
        rc = launch_step(h, &row, flags, s);
        if (rc != BCP_OK) return rc;
    }
    HIP_TRY(hipGetLastError());
    return BCP_OK;
}

extern "C" int bcp_side_stream(bcp_handle* h, int32_t cu_percent, void** stream)
{
    if (!h || !stream) return fail(BCP_E_INVALID, "bcp_side_stream: null argument");
    if (cu_percent < 1 || cu_percent > 100) return fail(BCP_E_INVALID, "bcp_side_stream: cu_percent must be 1 .. 100");
    HIP_TRY(hipSetDevice(h->device));
    if (h->side_stream && h->side_share != cu_percent) {
        HIP_TRY(hipStreamSynchronize(h->side_stream));
        HIP_TRY(hipStreamDestroy(h->side_stream));
        h->side_stream = nullptr;
    }
    if (!h->side_stream) {
        int cus = 0;
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device));
        // one bit per CU; the enabled ones are spread evenly (every k-th bit, whatever order the driver numbers the CUs of
        // the shader engines and XCDs in, every one of them keeps the same share)
        const int words = (cus + 31) / 32;
        std::vector<uint32_t> mask((size_t)std::max(words, 1), 0u);
        int enabled = 0;
        for (int c = 0; c < cus; ++c)
            if ((int64_t)(c + 1) * cu_percent / 100 > (int64_t)c * cu_percent / 100) {
                mask[(size_t)c / 32] |= 1u << (c % 32);
                ++enabled;
            }
        if (enabled == 0) mask[0] |= 1u;
        hipStream_t s = nullptr;
        if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
            (void)hipGetLastError();
            return fail(BCP_E_HIP, "bcp_side_stream: the runtime refused a CU-masked stream");
        }
        h->side_stream = s;
        h->side_share = cu_percent;
    }
    *stream = (void*)h->side_stream;
    return BCP_OK;
}

extern "C" int bcp_expired_waits(bcp_handle* h, int64_t* count, void* stream)
{
    if (!h || !count) return fail(BCP_E_INVALID, "bcp_expired_waits: null argument");
    HIP_TRY(hipSetDevice(h->device));
    uint64_t v = 0;
    HIP_TRY(hipMemcpyAsync(&v, h->tick + 4, sizeof(v), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    *count = (int64_t)v;
    return BCP_OK;
}

extern "C" int bcp_parked_poses(bcp_handle* h, int64_t* count, void* stream)
{
    if (!h || !count) return fail(BCP_E_INVALID, "bcp_parked_poses: null argument");
    HIP_TRY(hipSetDevice(h->device));
    *count = 0;
    if (!h->parked_slots) return BCP_OK;
    std::vector<uint64_t> slots((size_t)h->parked_cap);
    HIP_TRY(hipMemcpyAsync(slots.data(), h->parked_slots, slots.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    uint64_t sum = 0;
    for (uint64_t v : slots) sum += v;
    *count = (int64_t)sum;
    return BCP_OK;
}

extern "C" int bcp_step_form(bcp_handle* h)
{
    if (!h) return fail(BCP_E_INVALID, "bcp_step_form: null handle");
    if (!h->have_map || !h->have_path || !h->have_state) return fail(BCP_E_STATE, "bcp_step_form: costmaps, paths and state must be set first");
    if (!step_uses_deferral(h)) return 0;
    if (h->dense_threshold < 0) return 1;
    return (h->fused && h->adaptive) ? 3 : 2;
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
    for (int k = 0; k < steps && rc == BCP_OK; ++k) rc = launch_step(h, io, flags, s);
    HIP_TRY(hipEventRecord(e1, s));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    HIP_TRY(hipEventDestroy(e0));
    HIP_TRY(hipEventDestroy(e1));
    HIP_TRY(hipGetLastError());
    if (rc != BCP_OK) return rc;
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
    int rc = BCP_OK;
    for (int k = 0; k < steps && rc == BCP_OK; ++k) rc = launch_step(h, io, flags, s, first_only);
    HIP_TRY(hipEventRecord(e1, s));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    HIP_TRY(hipEventDestroy(e0));
    HIP_TRY(hipEventDestroy(e1));
    HIP_TRY(hipGetLastError());
    if (rc != BCP_OK) return rc;
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
    // full steps first (the state advances), then the same number of kernel-1-only launches on the reached state:
    // envs parked by a lone step_kernel are never finished, so every launch of that loop sees the same batch.
    float full = 0, first = 0;
    rc = time_loop(h, io, flags, steps, s, false, &full);
    if (rc != BCP_OK) return rc;
    if (bcp_step_form(h) != 2) {   // the step is ONE launch (step_local_kernel, step_kernel): nothing to split, and no second loop
        kernel_ms[0] = full;
        kernel_ms[1] = 0.0f;
        return BCP_OK;
    }
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

static int pose_collides_launch(bcp_handle* h, const double* poses, int64_t n, uint8_t* out, void* stream, int origin_in_map,
                                const char* who)
{
    if (!h || !poses || !out || n <= 0) return fail(BCP_E_INVALID, "%s: bad argument", who);
    if (!h->have_map) return fail(BCP_E_STATE, "%s: costmaps not set", who);
    HIP_TRY(hipSetDevice(h->device));
    { const int rc = ensure_fields(h, (hipStream_t)stream); if (rc != BCP_OK) return rc; }
    const int blocks = (int)((n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(pose_collides_kernel, dim3(blocks), dim3(kBlock),
                       collision_lds_bytes(h->params.n_verts, h->map.in_lds, h->map.rows, h->map.wpr),
                       (hipStream_t)stream, h->dev, h->map, h->cull, h->exact_mode, h->dense_threshold, h->wide, poses, n,
                       h->n, h->geom_of_env, out, origin_in_map, h->map_valid_rows, h->map_valid_cols);
    HIP_TRY(hipGetLastError());
    return BCP_OK;
}

extern "C" int bcp_pose_collides(bcp_handle* h, const double* poses, int64_t n, uint8_t* out, void* stream)
{
    return pose_collides_launch(h, poses, n, out, stream, 0, "bcp_pose_collides");
}

extern "C" int bcp_is_robot_colliding(bcp_handle* h, const double* poses, int64_t n, uint8_t* out, void* stream)
{
    return pose_collides_launch(h, poses, n, out, stream, 1, "bcp_is_robot_colliding");
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
    scale_footprint(P, h->params, resolution);
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

// done masks as bits: word w, bit b = mask[32 w + b] != 0 (a sharded job sends its masks over xGMI in this form)
__global__ void pack_mask_bits_kernel(const uint8_t* __restrict__ mask, int64_t n, uint32_t* __restrict__ bits)
{
    const int64_t words = (n + 31) / 32;
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < words; w += (int64_t)gridDim.x * blockDim.x) {
        uint32_t word = 0;
        if (32 * w + 32 <= n && ((uintptr_t)(mask + 32 * w) & 15) == 0) {
            const uint4* src = reinterpret_cast<const uint4*>(mask + 32 * w);
            const uint4 lo = src[0], hi = src[1];
            const uint32_t q[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) word |= (uint32_t)(((q[k] >> (8 * j)) & 255u) != 0) << (4 * k + j);
        } else {
            for (int b = 0; b < 32 && 32 * w + b < n; ++b) word |= (uint32_t)(mask[32 * w + b] != 0) << b;
        }
        bits[w] = word;
    }
}

__global__ void unpack_mask_bits_kernel(const uint32_t* __restrict__ bits, int64_t n, uint8_t* __restrict__ mask)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        mask[i] = (uint8_t)((bits[i >> 5] >> (i & 31)) & 1u);
}

extern "C" int bcp_pack_mask_bits(const uint8_t* mask, int64_t n, uint32_t* bits, void* stream)
{
    if (!mask || !bits || n <= 0) return fail(BCP_E_INVALID, "bcp_pack_mask_bits: bad argument");
    hipLaunchKernelGGL(pack_mask_bits_kernel, dim3(stride_grid((n + 31) / 32, 256)), dim3(256), 0, (hipStream_t)stream, mask, n, bits);
    HIP_TRY(hipGetLastError());
    return BCP_OK;
}

extern "C" int bcp_unpack_mask_bits(const uint32_t* bits, int64_t n, uint8_t* mask, void* stream)
{
    if (!mask || !bits || n <= 0) return fail(BCP_E_INVALID, "bcp_unpack_mask_bits: bad argument");
    hipLaunchKernelGGL(unpack_mask_bits_kernel, dim3(stride_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, bits, n, mask);
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


// ---- reward-provider / path-tools operator seams ---------------------------------------------------------------
static int ready_static(bcp_handle* h, hipStream_t s, const char* who)
{
    if (!h->have_path) return fail(BCP_E_STATE, "%s: paths must be set first", who);
    if (h->static_dirty) return upload_step_static(h, s);
    return BCP_OK;
}

extern "C" int bcp_reward(bcp_handle* h, const double* poses, int64_t n, double* min_spat_dist_so_far, int32_t* target_idx,
                          const uint8_t* robot_collided, double* reward, uint8_t* goal_reached, void* stream)
{
    if (!h || !poses || !min_spat_dist_so_far || !target_idx || !reward || n <= 0)
        return fail(BCP_E_INVALID, "bcp_reward: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const int rc = ready_static(h, s, "bcp_reward");
    if (rc != BCP_OK) return rc;
    hipLaunchKernelGGL(reward_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, h->dev_static, poses, n,
                       min_spat_dist_so_far, target_idx, robot_collided, reward, goal_reached);
    HIP_TRY(hipGetLastError());
    return BCP_OK;
}

extern "C" int bcp_find_last_reached(bcp_handle* h, const double* poses, int64_t n, int32_t* out, void* stream)
{
    if (!h || !poses || !out || n <= 0) return fail(BCP_E_INVALID, "bcp_find_last_reached: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const int rc = ready_static(h, s, "bcp_find_last_reached");
    if (rc != BCP_OK) return rc;
    hipLaunchKernelGGL(find_last_reached_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, h->dev_static, poses, n,
                       out);
    HIP_TRY(hipGetLastError());
    return BCP_OK;
}

extern "C" int bcp_path_velocity(bcp_handle* h, const double* path_txyth, int64_t n_rows, double* v, double* w, int32_t* err,
                                 void* stream)
{
    if (!h || !path_txyth || !v || !w || n_rows < 2) return fail(BCP_E_INVALID, "bcp_path_velocity: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    hipLaunchKernelGGL(path_velocity_kernel, dim3((unsigned)((n_rows - 1 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       path_txyth, n_rows, v, w, err);
    HIP_TRY(hipGetLastError());
    return BCP_OK;
}

extern "C" int bcp_is_footprint_colliding(bcp_handle* h, const uint8_t* image_slices, const uint8_t* blit_masks, int64_t n,
                                          int32_t rows, int32_t cols, uint8_t lethal, uint8_t* out, void* stream)
{
    if (!h || !image_slices || !blit_masks || !out || n <= 0 || rows <= 0 || cols <= 0)
        return fail(BCP_E_INVALID, "bcp_is_footprint_colliding: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    hipLaunchKernelGGL(footprint_colliding_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       image_slices, blit_masks, n, (int64_t)rows * cols, (uint32_t)lethal, out);
    HIP_TRY(hipGetLastError());
    return BCP_OK;
}


extern "C" int bcp_device_normals(bcp_handle* h, int64_t first_env, int64_t n_envs, uint64_t first_step, int32_t n_steps,
                                  double* out, void* stream)
{
    if (!h || !out || n_envs <= 0 || n_steps <= 0 || first_env < 0) return fail(BCP_E_INVALID, "bcp_device_normals: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    hipLaunchKernelGGL(device_normals_kernel, dim3(stride_grid(n_envs * n_steps, 256)), dim3(256), 0, (hipStream_t)stream,
                       h->seed, h->env_id_base + first_env, n_envs, first_step, n_steps, out);
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

// The cost model of the sparse route (tools/bench_ego_cells.py measures both sides on the box): per image the fill-and-patch
// kernel pays ~0.4 instructions per listed cell for the culling pass and ~2.5 per cell that meets the window, the sampling
// kernels ~0.1 per destination pixel when the map is staged in LDS whole and five times that when every workgroup stages the
// part of the map its window sees.  BCP_TUNE_EGO_SPARSE >= 2 is an explicit limit (tests, sweeps).
static int32_t ego_sparse_limit(int32_t tuning, int64_t pixels, bool fits_lds)
{
    if (tuning >= 2) return tuning;
    const int64_t lim = fits_lds ? pixels / 8 : pixels / 2;
    return (int32_t)std::max<int64_t>(kEgoCellCapMin, std::min<int64_t>(lim, 16384));
}

extern "C" int bcp_egocentric_route(bcp_handle* h, int32_t* info4)
{
    if (!h || !info4) return fail(BCP_E_INVALID, "bcp_egocentric_route: null argument");
    for (int k = 0; k < 4; ++k) info4[k] = h->ego_route[k];
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
    const size_t map_bytes = ((size_t)(a.rows + 2) * (a.cols + 2) + 7) & ~(size_t)7;   // LDS copy with a border ring
    const size_t row_bytes = ((size_t)a.drows * 2 + kEgoBoundInts) * sizeof(int32_t);   // one table: row terms, row bounds
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
    // Sparse maps and a zero border (extract_egocentric_costmap's default): zero fill + one patch per non-zero source cell
    // (ego_sparse_kernel).  Decided per call from the counts of non-zero cells: a counting pass over the maps on the first
    // such call after the maps were (re)bound, one read-back of the largest count, lists sized from it; a pool refresh keeps
    // counts and lists of the entries it re-samples up to date.  Maps with more cells than the cost model's limit (or a
    // non-zero border) keep the sampling kernels below.
    const bool fits_lds = map_bytes + 4 * row_bytes <= 150 * 1024;
    h->ego_route[0] = h->ego_route[1] = h->ego_route[2] = h->ego_route[3] = 0;
    if (border_value == 0 && a.rows <= 4095 && a.cols <= 4095 && !h->ego_cells_refused && h->ego_sparse &&
        ego_sparse_lds_bytes(a.drows, a.dcols, kEgoWaves) <= 64 * 1024) {
        const int64_t entries = a.shared ? 1 : n_slots(h);
        const int32_t limit = ego_sparse_limit(h->ego_sparse, (int64_t)a.drows * a.dcols, fits_lds);
        if (h->refresh_recorded && (!h->ego_cells_built || h->ego_cells_max < 0))
            HIP_TRY(hipStreamWaitEvent(st, h->refresh_done, 0));   // (a refresh on another stream may still be writing the maps / counts)
        if (h->ego_cells_entries != entries || !h->ego_cell_counts) {
            if (h->ego_cells) (void)hipFree(h->ego_cells);
            if (h->ego_cell_counts) (void)hipFree(h->ego_cell_counts);
            h->ego_cells = nullptr;
            h->ego_cell_counts = nullptr;
            h->ego_cells_entries = 0;
            h->ego_cell_cap = 0;
            h->ego_cells_built = false;
            if (hipMalloc((void**)&h->ego_cell_counts, (size_t)(entries + 1) * sizeof(int32_t)) != hipSuccess) {
                (void)hipGetLastError();
                h->ego_cell_counts = nullptr;
                h->ego_cells_refused = true;   // (no room: not an error, the sampling kernels take over)
            } else {
                h->ego_cells_entries = entries;
            }
        }
        if (h->ego_cell_counts && !h->ego_cells_built) {
            // counting pass -> largest count -> stride of the lists -> lists
            const EntrySelect all = {nullptr, nullptr, entries};
            if (h->ego_cells) (void)hipFree(h->ego_cells);
            h->ego_cells = nullptr;
            h->ego_cell_cap = 0;
            HIP_TRY(hipMemsetAsync(h->ego_cell_counts + entries, 0, sizeof(int32_t), st));
            launch_ego_cells(h, all, entries, st);
            HIP_TRY(hipMemcpyAsync(&h->ego_cells_max, h->ego_cell_counts + entries, sizeof(int32_t), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            h->ego_cells_built = true;
            if (h->ego_cells_max <= limit) {
                // pool entries change under a refresh: leave room for a world with more cells than today's largest
                int64_t cap = std::max<int64_t>(kEgoCellCapMin, ((int64_t)h->ego_cells_max + 63) & ~(int64_t)63);
                const int64_t budget = (int64_t)1 << 30;   // bytes of lists per handle
                if (entries * cap * 4 > budget) cap = ((int64_t)h->ego_cells_max + 63) & ~(int64_t)63;
                if (h->ego_stride > 0) cap = h->ego_stride;   // (tests: entries with more cells than this are drawn pixel by pixel)
                if (cap > 0 && entries * cap * 4 <= budget &&
                    hipMalloc((void**)&h->ego_cells, (size_t)entries * cap * sizeof(uint32_t)) == hipSuccess) {
                    h->ego_cell_cap = (int32_t)cap;
                    const int32_t counted = h->ego_cells_max;
                    launch_ego_cells(h, all, entries, st);   // (the same counts again, and the lists)
                    h->ego_cells_max = counted;
                } else {
                    (void)hipGetLastError();
                    h->ego_cells = nullptr;
                    if (cap > 0) h->ego_cells_refused = true;
                }
            }
        }
        if (h->ego_cell_counts && h->ego_cells_built && h->ego_cells_max < 0) {   // (a refresh re-counted some entries)
            HIP_TRY(hipMemcpyAsync(&h->ego_cells_max, h->ego_cell_counts + entries, sizeof(int32_t), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
        }
        h->ego_route[1] = h->ego_cells_max;
        h->ego_route[2] = h->ego_cell_cap;
        h->ego_route[3] = limit;
        if (h->ego_cells && h->ego_cells_built && h->ego_cells_max >= 0 && h->ego_cells_max <= limit) {
            // One image per wave, eight per workgroup: 8 192 short workgroups for 65 536 images.  (Round 3 first ran this kernel
            // persistently -- as many workgroups as the chip holds, 64 images per wave, the lanes sharing the transforms' float64
            // arithmetic: 11 % slower on the same box, 0.249 against 0.222 ms.  Stores from many short workgroups drain faster than
            // from a few long-lived ones, tools/fill_rate.hip; the arithmetic saved was never the bottleneck, VALU busy 17 %.
            // Also measured: an image split over 2 / 4 waves of a workgroup (+- 0 / 14 % slower), a plain one-image kernel with
            // 48 instead of 83 registers (3 - 8 % slower), fewer workgroups per CU by way of unused LDS (within the noise).)
            const dim3 wide(64 * kEgoWaves);
            const dim3 grid((unsigned)((n + kEgoWaves - 1) / kEgoWaves));
            const size_t lds = ego_sparse_lds_bytes(a.drows, a.dcols, kEgoWaves);   // (<= 64 KB: checked above)
            hipLaunchKernelGGL(ego_sparse_kernel, grid, wide, lds, st, a, h->ego_cells, h->ego_cell_counts, h->ego_cell_cap);
            HIP_TRY(hipGetLastError());
            h->ego_route[0] = BCP_EGO_SPARSE;
            return BCP_OK;
        }
    }
    if (!a.shared && fits_lds && n < ((int64_t)1 << 31)) {
        h->ego_route[0] = BCP_EGO_BINNED;
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
        if (lds > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, lds));
        const dim3 grid((unsigned)std::min<int64_t>(n, (int64_t)std::max(per_cu, 1) * cus));
        a.stage_map = 1;
        if (px8) hipLaunchKernelGGL((ego_costmap_binned_kernel<8>), grid, block, lds, st, a, bin_start, bin_count, order);
        else hipLaunchKernelGGL((ego_costmap_binned_kernel<4>), grid, block, lds, st, a, bin_start, bin_count, order);
    } else {
        // shared map (staged in LDS when it fits) or maps too large for LDS: persistent workgroups, as many as are
        // resident at once
        // (gfx950 gives a workgroup up to 160 KB of LDS; a big copy costs occupancy, but LDS sampling still wins)
        a.stage_map = (a.shared && fits_lds) ? 1 : 0;
        // too large: each workgroup stages just the part of the map its window can see -- at most the window's
        // diagonal (+ 2 px of rounding, + ring) squared
        const double diag = std::sqrt((double)a.drows * a.drows + (double)a.dcols * a.dcols);
        const size_t side = (size_t)std::ceil(diag) + 5;
        const size_t win_bytes = (side * side + 7) & ~(size_t)7;
        if (!a.stage_map && win_bytes + row_bytes <= 60 * 1024) {
            a.win_lds_bytes = (int32_t)win_bytes;
            const size_t lds = win_bytes + row_bytes;
            const void* fn = px8 ? (const void*)ego_costmap_window_kernel<8> : (const void*)ego_costmap_window_kernel<4>;
            int per_cu = 0;
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, lds));
            const dim3 grid((unsigned)std::min<int64_t>(n, (int64_t)std::max(per_cu, 1) * cus));
            if (px8) hipLaunchKernelGGL((ego_costmap_window_kernel<8>), grid, block, lds, st, a);
            else hipLaunchKernelGGL((ego_costmap_window_kernel<4>), grid, block, lds, st, a);
            HIP_TRY(hipGetLastError());
            h->ego_route[0] = BCP_EGO_WINDOW;
            return BCP_OK;
        }
        const int waves = kEgoWaves;
        h->ego_route[0] = a.stage_map ? BCP_EGO_STAGED : BCP_EGO_GLOBAL;
        const size_t lds = waves * row_bytes + (a.stage_map ? map_bytes : 0);
        const void* fn = a.stage_map ? (px8 ? (const void*)ego_costmap_kernel<true, 8> : (const void*)ego_costmap_kernel<true, 4>)
                                     : (px8 ? (const void*)ego_costmap_kernel<false, 8> : (const void*)ego_costmap_kernel<false, 4>);
        if (lds > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64 * waves, lds));
        const dim3 grid((unsigned)std::min<int64_t>((n + waves - 1) / waves, (int64_t)std::max(per_cu, 1) * cus));
        const dim3 wide(64 * waves);
        if (a.stage_map) {
            if (px8) hipLaunchKernelGGL((ego_costmap_kernel<true, 8>), grid, wide, lds, st, a);
            else hipLaunchKernelGGL((ego_costmap_kernel<true, 4>), grid, wide, lds, st, a);
        } else {
            if (px8) hipLaunchKernelGGL((ego_costmap_kernel<false, 8>), grid, wide, lds, st, a);
            else hipLaunchKernelGGL((ego_costmap_kernel<false, 4>), grid, wide, lds, st, a);
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

extern "C" int bcp_goal_direction_state(bcp_handle* h, const double* world_size, double* out, void* stream)
{
    if (!h || !world_size || !out) return fail(BCP_E_INVALID, "bcp_goal_direction_state: null argument");
    if (!h->have_path || !h->have_state)
        return fail(BCP_E_STATE, "bcp_goal_direction_state: paths and state must be set first");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    if (h->static_dirty) {
        const int rc = upload_step_static(h, s);
        if (rc != BCP_OK) return rc;
    }
    hipLaunchKernelGGL(goal_direction_state_kernel, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, s, h->dev_static,
                       world_size[0], world_size[1], out);
    HIP_TRY(hipGetLastError());
    return BCP_OK;
}

// ---- RandomMiniEnv worlds sampled on the device ----------------------------------------------------------------
extern "C" int bcp_mini_world_seed(bcp_handle* h, const int64_t* seeds, int64_t n_chains, uint32_t* mt_state, void* stream)
{
    if (!h || !seeds || !mt_state || n_chains <= 0) return fail(BCP_E_INVALID, "bcp_mini_world_seed: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    hipLaunchKernelGGL(mt_seed_kernel, dim3((unsigned)((n_chains + 255) / 256)), dim3(256), 0, (hipStream_t)stream, seeds,
                       n_chains, mt_state);
    HIP_TRY(hipGetLastError());
    return BCP_OK;
}

static int sample_mini_worlds(bcp_handle* h, const bcp_mini_world_params* p, uint32_t* mt_state, int64_t n_chains,
                              int32_t episodes, int32_t rows, int32_t cols, const int32_t* counts, const int64_t* first_world,
                              double* worlds, uint8_t* maps, int32_t* status, void* stream)
{
    if (!(p->resolution > 0) || !check_kernel_size(h->params, p->resolution))
        return fail(BCP_E_INVALID, "bcp_sample_mini_worlds: bad resolution for this footprint");
    if ((int)(0.05 / p->resolution) > 1)   // Wall.render: thickness = max(1, int(width / resolution))
        return fail(BCP_E_INVALID, "bcp_sample_mini_worlds: walls thicker than one pixel are not supported");
    const int wpr = (cols + 31) / 32;
    const size_t lds = sample_lds_words(rows, wpr) * sizeof(uint32_t);
    if (rows <= 0 || cols <= 0 || lds > 60 * 1024) return fail(BCP_E_INVALID, "bcp_sample_mini_worlds: unsupported map shape");
    HIP_TRY(hipSetDevice(h->device));
    DevParams P = h->dev;
    scale_footprint(P, h->params, p->resolution);
    MiniWorldParams mp;
    mp.inner_h = p->inner_h;
    mp.inner_w = p->inner_w;
    mp.mid_margin = p->mid_margin;
    mp.out_margin = p->out_margin;
    mp.min_obstacle_angle = p->min_obstacle_angle;
    mp.max_obstacle_angle = p->max_obstacle_angle;
    mp.lim_euc_dist = p->lim_euc_dist;
    mp.lim_ang_dist = p->lim_ang_dist;
    mp.angular_pose_noise_scale = p->angular_pose_noise_scale;
    mp.resolution = p->resolution;
    mp.goal_spat_dist = p->goal_spat_dist;
    mp.goal_ang_dist = p->goal_ang_dist;
    if (footprint_is_wide(h->params, p->resolution))
        hipLaunchKernelGGL(mini_world_sample_kernel<true>, dim3((unsigned)n_chains), dim3(64), lds, (hipStream_t)stream, P, mp,
                           mt_state, n_chains, (int)episodes, (int)rows, (int)cols, counts, first_world, worlds, maps, status);
    else
        hipLaunchKernelGGL(mini_world_sample_kernel<false>, dim3((unsigned)n_chains), dim3(64), lds, (hipStream_t)stream, P, mp,
                           mt_state, n_chains, (int)episodes, (int)rows, (int)cols, counts, first_world, worlds, maps, status);
    HIP_TRY(hipGetLastError());
    return BCP_OK;
}

extern "C" int bcp_sample_mini_worlds(bcp_handle* h, const bcp_mini_world_params* p, uint32_t* mt_state, int64_t n_chains,
                                      int32_t episodes, int32_t rows, int32_t cols, double* worlds, uint8_t* maps,
                                      int32_t* status, void* stream)
{
    if (!h || !p || !mt_state || !worlds || !maps || !status || n_chains <= 0 || episodes <= 0)
        return fail(BCP_E_INVALID, "bcp_sample_mini_worlds: bad argument");
    return sample_mini_worlds(h, p, mt_state, n_chains, episodes, rows, cols, nullptr, nullptr, worlds, maps, status, stream);
}

static int check_ring(const bcp_handle* h, int32_t episodes, const char* who)
{
    if (episodes < 2 || h->n_geoms <= 0 || (int64_t)h->n_geoms != h->n * episodes || !h->next_geom)
        return fail(BCP_E_STATE, "%s: needs a geometry pool of n_envs x episodes (>= 2) entries with next_geom", who);
    if (!h->have_map || !h->have_path || !h->have_init || h->map.shared || h->path.shared || h->map_valid_rows ||
        h->map_valid_cols)
        return fail(BCP_E_STATE, "%s: pool costmaps, paths and initial state must be set first", who);
    return BCP_OK;
}

extern "C" int bcp_plan_mini_worlds(bcp_handle* h, int32_t episodes, int64_t* generated, int32_t* info, void* stream)
{
    if (!h || !generated || !info) return fail(BCP_E_INVALID, "bcp_plan_mini_worlds: null argument");
    const int rc = check_ring(h, episodes, "bcp_plan_mini_worlds");
    if (rc != BCP_OK) return rc;
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = h->n, G = h->n_geoms;
    const size_t bytes = (size_t)n * sizeof(int64_t) + (size_t)(n + G + 4) * sizeof(int32_t);
    if (bytes > h->ring_bytes) {
        if (h->ring) HIP_TRY(hipFree(h->ring));
        h->ring = nullptr;
        h->ring_bytes = 0;
        HIP_TRY(hipMalloc((void**)&h->ring, bytes));
        h->ring_bytes = bytes;
    }
    int64_t* first_world = (int64_t*)h->ring;
    int32_t* counts = (int32_t*)(first_world + n);
    int32_t* dirty = counts + n;
    int32_t* tally = dirty + G;
    HIP_TRY(hipMemsetAsync(tally, 0, 4 * sizeof(int32_t), s));
    hipLaunchKernelGGL(mini_world_ring_plan_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, (int)episodes,
                       h->geom_of_env, const_cast<int32_t*>(h->next_geom), generated, counts, first_world, dirty, tally);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(info, tally, 4 * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    h->ring_episodes = episodes;
    h->ring_planned = true;
    return BCP_OK;
}

extern "C" int bcp_refresh_mini_worlds(bcp_handle* h, const bcp_mini_world_params* p, uint32_t* mt_state, double* worlds,
                                       uint8_t* maps, double* paths, int32_t* lens, double* init, double path_delta,
                                       int32_t* status, int32_t* path_status, void* stream)
{
    if (!h || !p || !mt_state || !worlds || !maps || !paths || !lens || !init || !status || !path_status || !(path_delta > 0))
        return fail(BCP_E_INVALID, "bcp_refresh_mini_worlds: bad argument");
    if (!h->ring_planned) return fail(BCP_E_STATE, "bcp_refresh_mini_worlds: call bcp_plan_mini_worlds first");
    const int32_t episodes = h->ring_episodes;
    const int rc0 = check_ring(h, episodes, "bcp_refresh_mini_worlds");
    if (rc0 != BCP_OK) return rc0;
    if (maps != h->map_data || paths != h->path_src || lens != h->path.lens)
        return fail(BCP_E_INVALID, "bcp_refresh_mini_worlds: maps / paths / lens are not the arrays this handle was given");
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = h->n, G = h->n_geoms;
    const int64_t* first_world = (const int64_t*)h->ring;
    const int32_t* counts = (const int32_t*)(first_world + n);
    const int32_t* dirty = counts + n;
    const int32_t* tally = dirty + G;
    const int rc = sample_mini_worlds(h, p, mt_state, n, episodes, h->map.rows, h->map.cols, counts, first_world, worlds, maps,
                                      status, stream);
    if (rc != BCP_OK) return rc;
    const EntrySelect sel = {dirty, tally, G};
    hipLaunchKernelGGL(mini_world_paths_kernel, dim3(stride_grid(G, 128, true)), dim3(128), 0, s, worlds, sel, path_delta,
                       h->params.spatial_precision, h->params.angular_precision,
                       (int)(h->params.reward_provider == BCP_REWARD_PURE_PURSUIT), (int)h->path.max_len, paths, lens, init,
                       path_status);
    launch_pack_bitmap(h, sel, G, s);
    if (h->cull.edt) {
        // under the single-launch step nothing reads the uint8 fields: tiles only, the fields follow on demand
        const bool tiles_only = h->pending && h->fused && h->adaptive && h->cull.on;
        const int rc2 = launch_distance_field(h, sel, G, s, tiles_only);
        if (rc2 != BCP_OK) return rc2;
    }
    launch_path_data(h, sel, G, s);
    hipLaunchKernelGGL(pool_initial_state_kernel, dim3(stride_grid(G, 256, true)), dim3(256), 0, s, sel, paths,
                       (int)h->path.max_len, init, h->init);
    HIP_TRY(hipGetLastError());
    if (!h->refresh_done) HIP_TRY(hipEventCreateWithFlags(&h->refresh_done, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(h->refresh_done, s));
    h->refresh_recorded = true;
    h->ring_planned = false;
    h->ring_refreshed = true;
    return BCP_OK;
}

extern "C" int bcp_release_mini_worlds(bcp_handle* h, void* stream)
{
    if (!h) return fail(BCP_E_INVALID, "bcp_release_mini_worlds: null handle");
    if (!h->ring || !h->ring_refreshed || (int64_t)h->n_geoms != h->n * h->ring_episodes || !h->next_geom)
        return fail(BCP_E_STATE, "bcp_release_mini_worlds: no bcp_refresh_mini_worlds to complete");
    HIP_TRY(hipSetDevice(h->device));
    const int64_t n = h->n;
    const int64_t* first_world = (const int64_t*)h->ring;
    const int32_t* counts = (const int32_t*)(first_world + n);
    hipLaunchKernelGGL(mini_world_ring_release_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n,
                       (int)h->ring_episodes, counts, first_world, const_cast<int32_t*>(h->next_geom));
    HIP_TRY(hipGetLastError());
    h->ring_refreshed = false;
    return BCP_OK;
}

extern "C" int bcp_mini_world_paths(bcp_handle* h, const double* worlds, int64_t n_worlds, double path_delta, int32_t max_len,
                                    double* paths, int32_t* lens, double* init, int32_t* status, void* stream)
{
    if (!h || !worlds || !paths || !lens || !init || !status || n_worlds <= 0 || max_len < 2 || !(path_delta > 0))
        return fail(BCP_E_INVALID, "bcp_mini_world_paths: bad argument");
    HIP_TRY(hipSetDevice(h->device));
    const EntrySelect all = {nullptr, nullptr, n_worlds};
    hipLaunchKernelGGL(mini_world_paths_kernel, dim3(stride_grid(n_worlds, 128)), dim3(128), 0, (hipStream_t)stream, worlds, all,
                       path_delta, h->params.spatial_precision, h->params.angular_precision,
                       (int)(h->params.reward_provider == BCP_REWARD_PURE_PURSUIT), (int)max_len, paths, lens, init, status);
    HIP_TRY(hipGetLastError());
    return BCP_OK;
}
