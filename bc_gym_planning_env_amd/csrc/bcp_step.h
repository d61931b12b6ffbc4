// bcp_step.h -- device code of the batched PlanEnv.step(): the parameter blocks shared with the host, the one-time
// path kernels, the reward providers, the delay queues, and the step kernels (general / fast / pending).  Included by
// bcplan.hip, which holds the handle and the C entry points.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <type_traits>

#include "bcp_device.h"
#include "bcp_raster.h"
#include "bcp_coop.h"

using namespace bcp;

struct DevState {
    double *x, *y, *angle, *v, *w, *steer, *wheel, *min_dist;
    int32_t *target_idx, *cur_iter;
    uint8_t* collided;
    double *pose_seen, *state_seen;         // [3][n] / [7][n], delays > 0 only
    double *control_q, *pose_q, *state_q;   // [delay][width][n]
};

struct MapDesc {
    const uint32_t* bits;  // lethal bitmap, [rows][wpr] shared or [N][rows][wpr]
    const uint32_t* tiles; // the same bits in tiles of 32 x 32 cells, [tile_words] per entry (pack_bitmap_kernel, CoopCollisionSink)
    int32_t rows, cols, wpr;
    int32_t shared;
    int32_t in_lds;        // shared bitmap small enough to be staged in LDS
    int64_t env_stride;    // words per env (0 when shared)
    const double* origins; // device [N,2] when per-env, else NULL
    double ox, oy, inv_res;
};

struct PathDesc {
    const double* pts;   // [len][5] = x, y, theta, cos(theta), sin(theta); shared or [N][max_len][5]
    const uint32_t* pre; // private paths: [N][max_len][2] = x, y (uint16 each, in steps from the box corner) and cos, sin of theta
                         // (int16, / 32767) -- the 8-byte prefilter record of
                         // the way-point scan (16 bytes: four way points per 64-byte sector); nullptr for a shared path
    const double* bbox;  // [kBoxDoubles] per path, one 128-byte record: box of the way points and bucket grid as eight f32,
                         // the costmap origin, the path length and, for private paths, the bucket tables -- see kBoxDoubles
    const int16_t* index; // shared path: [2 axes][kPathBuckets][2] = first / last way point index that can be reached from a bucket
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

// What a wave needs before anything else -- where its state, its path and (kernel 2) its parked poses are -- travels
// with the kernel arguments: the first vector loads then depend on the argument fetch alone and overlap the (cold, the
// kernel boundary emptied the L2) scalar fetch of *S instead of queueing behind it.  A copy of the corresponding *S fields.
struct StepHot {
    DevState st;               // (x .. collided first: what every configuration loads; the five delay-queue arrays last)
    // ---- the rest of what step_local_kernel's prologue uses, contiguous (fetched in one go: fetch / pin_words)
    int64_t n;
    int64_t env_id_base;
    int32_t* geom_of_env;
    const double* path_pts;
    const uint32_t* path_pre;
    const double* path_bbox;
    const int16_t* path_index;
    const uint32_t* map_bits;
    const double* qverts;      // = &S->P.qverts[0][0]
    const double* map_origins; // per-entry map origins [.,2] or nullptr
    int32_t model, lds_path_doubles, path_shared, path_max_len;
    int32_t map_rows, map_cols, map_wpr, map_shared;
    int32_t noise_on, n_verts, control_delay, pose_delay, state_delay;
    int32_t dynamic_model, model_front_column_pid;
    int32_t noise_slot0;       // alpha1 > 0 or alpha2 > 0: the noise model can consume slot 0 (differential_drive.py:62)
    int32_t pending_cap;
    // (copies of StepArgs::flags / actions / noise_z / tick: with them here everything step_local_kernel's prologue fetches
    //  lies in the first five 64-byte lines of the launch arguments instead of being spread over seven -- a cold argument
    //  fetch costs ~160 cycles per further line, at the very start of every wave)
    uint32_t io_flags;
    const void* io_actions;
    const double* io_noise_z;
    uint64_t* io_tick;
    // ---- used later in a step
    const uint32_t* near;
    int64_t map_env_stride;
    struct Pending* pending;
};
constexpr int kHotStateWords = 11 * 2;                                   // x .. collided
constexpr int kHotPrologueWords = (10 * 8 + 18 * 4 + 3 * 8) / 4;        // n .. io_tick
static_assert(offsetof(DevState, collided) == 10 * 8 && offsetof(StepHot, st) == 0, "x .. collided lead DevState");
static_assert(offsetof(StepHot, io_tick) + 8 - offsetof(StepHot, n) == kHotPrologueWords * 4, "the prologue block of StepHot");

constexpr int kShards = 64;  // a wave parks into shard (block index % kShards)

// Per-launch kernel arguments (small).
struct StepArgs {
    const StepStatic* S;
    StepHot hot;
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
    int32_t* inplace_count;    // [kShards] undecided poses settled inside kernel 1 this step (the parked ones are in
                               // pending_count); sharded like the parking counters: no hot atomic
    int32_t* inplace_next;     // [kShards] next step's counters; kernel 1 zeroes them
    uint64_t seed, step_counter;
    uint32_t flags;
    // The step counter (parity of the alternating counter sets above, counter of the noise stream) and the noise seed
    // live on the DEVICE, so that the launch arguments of a step never change and a captured hipGraph of steps can be
    // replayed: the kernels fill in the eight fields above themselves (resolve_step).  tick[0] = counter as kernel 1 (or
    // the single-kernel step) reads it, tick[1] = as kernel 2 reads it, tick[2] = seed, tick[3] = ticket of the
    // single-kernel step.  Two-kernel step: kernel 1 copies tick[0] to tick[1], kernel 2 stores tick[1] + 1 to tick[0]
    // -- each word is only written while no kernel that reads it is running.  Single-kernel step: the last workgroup
    // to finish (ticket) advances tick[0].  tick[4] counts the bounded waits of step_local_kernel that gave up (bcp_expired_waits).
    // step_local_kernel draws its ticket from tick[8], a cache line of its own (see there) -- tick has 16 words.
    uint64_t* tick;
    int32_t* pending_base;     // [2][kShards] or nullptr
    int32_t* adapt_base;       // [2] thresholds + [2][kShards] in-place counters, or nullptr (no adaptation)
    uint64_t* parked_slots;    // [workgroups of step_local_kernel] parked poses so far, a word per workgroup (bcp_parked_poses)
    int32_t rollout_steps;     // step_local_kernel<.., ROLL = true> (bcp_rollout): steps per launch; actions / noise_z / outputs are [steps][N]..
    const uint32_t* map_tiles; // MapDesc::tiles: what step_local_kernel's exact tests read when the map is not staged in LDS
};
constexpr int kTickWords = 16;
constexpr int kTickLocalTicket = 8;   // step_local_kernel's ticket: not on the line the prologues read the counter and the seed from

constexpr uint32_t kStepAdvances = 1u << 24;   // internal flag: this launch is the last kernel of its step

// which: 0 = kernel 1 / single-kernel step, 1 = kernel 2
__device__ __forceinline__ StepArgs resolve_step(const StepArgs& in, int which)
{
    StepArgs a = in;
    const uint64_t step = in.tick[which];
    a.step_counter = step;
    a.seed = in.tick[2];
    const int p = (int)(step & 1u);
    a.pending_count = in.pending_base ? in.pending_base + p * kShards : nullptr;
    a.pending_next = in.pending_base ? in.pending_base + (p ^ 1) * kShards : nullptr;
    a.threshold_now = in.adapt_base ? in.adapt_base + p : nullptr;
    a.threshold_next = in.adapt_base ? in.adapt_base + (p ^ 1) : nullptr;
    a.inplace_count = in.adapt_base ? in.adapt_base + 2 + p * kShards : nullptr;
    a.inplace_next = in.adapt_base ? in.adapt_base + 2 + (p ^ 1) * kShards : nullptr;
    return a;
}

// end of a kernel that is the last one of its step (flag kStepAdvances): the last workgroup to get here moves the
// counter on.  Every workgroup has read tick[0] long before it takes its ticket.
__device__ __forceinline__ void advance_step_by_ticket(const StepArgs& a)
{
    if (!(a.flags & kStepAdvances) || threadIdx.x != 0) return;
    unsigned int* ticket = reinterpret_cast<unsigned int*>(a.tick + 3);
    __threadfence();
    if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
        *ticket = 0u;
        a.tick[0] = a.step_counter + 1;
    }
}

// Ablation switches in the upper half of the step flags (tools/ablate*.py time the step with stages removed; results
// are then WRONG by construction).  Not part of the ABI: bcplan.h only defines bits 0-1.
enum : uint32_t {
    kAblateNoCollision = 1u << 16,   // skip pose_collides altogether
    kAblateNoReward = 1u << 17,      // skip the reward scan
    kAblateNoCoop = 1u << 19,        // kernel 2: skip the cooperative rasteriser
    kAblateNoPark = 1u << 21,        // kernel 1: do not park undecided envs
    kAblateNoClassify = 1u << 22     // kernel 1: skip the distance-field lookups
};
// The ablation switches exist in -DBCP_DIAG builds only (tools/libbcplan_diag.so): the shipping kernels neither fetch nor test them.
#ifdef BCP_DIAG
#define ABLATED(a, bits) (((a).flags & (bits)) != 0)
#else
#define ABLATED(a, bits) false
#endif


constexpr int kBlock = 64;  // one wavefront per workgroup

// Which entries of a non-shared array (costmaps, paths, ...) a set-up kernel derives data for: all n of them, or the
// *count entries listed in `list` (both on the device: a pool that is topped up between steps, bcp_refresh_mini_worlds).
// The kernels walk `size() * units-per-entry` work items in a grid-stride loop, so their grids never depend on *count.
struct EntrySelect {
    const int32_t* list;
    const int32_t* count;
    int64_t n;
    __device__ __forceinline__ int64_t size() const { return list ? (int64_t)*count : n; }
    __device__ __forceinline__ int64_t entry(int64_t k) const { return list ? (int64_t)list[k] : k; }
};

// uint8 costmap -> 1-bit lethal mask.  One thread per 32-bit output word.
// (tiles: the same words once more in tiles of 32 x 32 cells, map_tile_words() per entry -- see CoopCollisionSink)
__host__ __device__ __forceinline__ int64_t map_tile_words(int rows, int wpr) { return (int64_t)((rows + 31) & ~31) * wpr; }

__global__ void pack_bitmap_kernel(const uint8_t* __restrict__ data, uint32_t* __restrict__ bits, uint32_t* __restrict__ tiles,
                                   EntrySelect sel, int rows, int cols, int wpr, const int32_t* __restrict__ valid_rows,
                                   const int32_t* __restrict__ valid_cols)
{
    const int64_t total = sel.size() * rows * wpr;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(it % wpr);
        const int64_t t = it / wpr;
        const int r = (int)(t % rows);
        const int64_t m = sel.entry(t / rows);
        const int64_t idx = (m * rows + r) * wpr + w;
        const int vr = valid_rows ? valid_rows[m] : rows;
        const int vc = valid_cols ? valid_cols[m] : cols;
        uint32_t word = 0;
        if (r < vr) {
            const uint8_t* src = data + (m * rows + r) * (int64_t)cols + (int64_t)w * 32;
            const int lim = min(32, vc - w * 32);
            if (lim == 32) {
                // the 32 cells as four 8-byte loads (any alignment: the rows of a 183-column map start anywhere) instead of 32
                // byte loads -- the kernel was bound by its load instructions, not by bytes (a pool refresh: 0.11 - 0.4 ms of
                // this kernel for ~7000 maps) --; "byte == LETHAL" for eight bytes at once: the exact zero-byte test of
                // x ^ LETHAL.., then the eight flags gathered into a byte by a multiplication
                typedef uint64_t __attribute__((aligned(1))) PackU64Unaligned;
                constexpr uint64_t kOnes = 0x0101010101010101ull, kLow7 = 0x7F7F7F7F7F7F7F7Full;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint64_t t = reinterpret_cast<const PackU64Unaligned*>(src)[q] ^ ((uint64_t)BCP_LETHAL * kOnes);
                    const uint64_t hit = ~((((t & kLow7) + kLow7) | t) | kLow7);   // 0x80 in every byte of t that is zero
                    word |= (uint32_t)(((hit >> 7) * 0x0102040810204080ull) >> 56) << (8 * q);
                }
            } else {
                for (int b = 0; b < lim; ++b) word |= (uint32_t)(src[b] == BCP_LETHAL) << b;
            }
        }
        bits[idx] = word;
        if (tiles) tiles[m * map_tile_words(rows, wpr) + (((r >> 5) * wpr + w) << 5) + (r & 31)] = word;
    }
}

constexpr int kPathBuckets = 64;          // buckets per axis of a shared path's tables (PathDesc::index, staged in LDS)
constexpr int kPathBucketsCompact = 16;   // ... of a private path's, which live inside its record
// One 128-byte record per path entry -- ONE line of memory: a load that misses L2 fetches the whole line whatever it asks
// for (tools/sector_probe.hip: 128 bytes per lone 4-byte read, and no more for a second word 32 or 64 bytes further), so
// every small per-entry array an env reads from costs it a line per step.  With private paths all of them share this one:
//   floats  [0..3]    xmin, xmax, ymin, ymax of the way points, rounded OUTWARD to f32 (the box only ever prunes: a wider one
//                     hands a few more poses to the bucket tables, whose windows the exact test then walks)
//   floats  [4..7]    the bucket grid x0, 1/wx, y0, 1/wy, as f32 (the tables are built from these very values, below, so a
//                     look-up and the table it reads agree whatever their rounding)
//   doubles [4],[5]   origin of the entry's costmap (world_record_kernel)
//   int32   [12],[13] number of way points; shift of the compact tables' indices (0 unless the paths have > 255 way points)
//   floats  [14],[15] step of the quantised prefilter records (path_trig_kernel) and its reciprocal
//   bytes   [64..127] private paths: the bucket tables, [2 axes][kPathBucketsCompact] {first, last} >> shift as uint8
//                     (path_index_compact_kernel); until round 4 [2][64] int16 pairs in an array of their own, two lines per step
// (rounds 2-3: 128 bytes in two halves -- box and grid as doubles, origin and length -- beside 512 bytes of tables)
constexpr int kBoxDoubles = 16;
constexpr int kBoxOrigin = 4;
constexpr int kBoxLenWord = 12;      // (int32 index; [13]: the index shift)
constexpr int kBoxQuantStep = 14;    // (float index)
constexpr int kBoxIndexU16 = 32;     // (uint16 index of the compact tables: {first | last << 8} per bucket)
constexpr int kQuantSteps = 65000;   // the longer side of the box in steps (uint16 coordinates)
constexpr int kQuantReach = 16384;   // spatial precision in steps, at most: poses the window lets through stay below 2^17 steps

// path [.,3] -> [.,5] with cos/sin of the heading (utilities/path_tools.py:405)
// (pre: the quantised prefilter record of private paths, last_reached_prefiltered -- needs the entry's record, `bbox`,
//  which path_bbox_kernel writes: launched behind it)
__global__ void path_trig_kernel(const double* __restrict__ xyt, double* __restrict__ out, uint32_t* __restrict__ pre,
                                 const double* __restrict__ bbox, EntrySelect sel, int max_len)
{
    const int64_t total = sel.size() * max_len;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = sel.entry(it / max_len) * max_len + it % max_len;
        const double th = xyt[3 * i + 2];
        out[5 * i + 0] = xyt[3 * i + 0];
        out[5 * i + 1] = xyt[3 * i + 1];
        out[5 * i + 2] = th;
        const double c = cos(th), sn = sin(th);
        out[5 * i + 3] = c;
        out[5 * i + 4] = sn;
        if (pre) {
            const float* rec = reinterpret_cast<const float*>(bbox + (i / max_len) * kBoxDoubles);
            const double ox = (double)rec[0], oy = (double)rec[2], step = (double)rec[kBoxQuantStep];
            const uint32_t qx = (uint32_t)fmin(fmax(rint((xyt[3 * i + 0] - ox) / step), 0.0), 65535.0);
            const uint32_t qy = (uint32_t)fmin(fmax(rint((xyt[3 * i + 1] - oy) / step), 0.0), 65535.0);
            const int32_t qc = (int32_t)rint(c * 32767.0), qs = (int32_t)rint(sn * 32767.0);
            pre[2 * i + 0] = qx | (qy << 16);
            pre[2 * i + 1] = ((uint32_t)qc & 0xFFFFu) | ((uint32_t)qs << 16);
        }
    }
}

__device__ __forceinline__ float f32_below(double v)
{
    float f = (float)v;
    return (double)f > v ? nextafterf(f, -INFINITY) : f;
}

__device__ __forceinline__ float f32_above(double v)
{
    float f = (float)v;
    return (double)f < v ? nextafterf(f, INFINITY) : f;
}

// Per path: bounding box of the way points and a 1-D grid of `buckets` buckets per axis over [min - sp, max + sp].
__device__ __forceinline__ void path_bbox_one(const double* __restrict__ xyt, const int32_t* __restrict__ lens, int max_len,
                                              int64_t p, double sp_prune, int buckets, double* __restrict__ bbox)
{
    const int m = lens ? lens[p] : max_len;
    const double* q = xyt + p * (int64_t)max_len * 3;
    double x0 = INFINITY, x1 = -INFINITY, y0 = INFINITY, y1 = -INFINITY;
    for (int j = 0; j < m; ++j) {
        x0 = fmin(x0, q[3 * j]);
        x1 = fmax(x1, q[3 * j]);
        y0 = fmin(y0, q[3 * j + 1]);
        y1 = fmax(y1, q[3 * j + 1]);
    }
    float* o = reinterpret_cast<float*>(bbox + kBoxDoubles * p);
    o[0] = f32_below(x0);
    o[1] = f32_above(x1);
    o[2] = f32_below(y0);
    o[3] = f32_above(y1);
    const double wx = fmax((x1 - x0 + 2.0 * sp_prune) / buckets, 1e-9);
    const double wy = fmax((y1 - y0 + 2.0 * sp_prune) / buckets, 1e-9);
    o[4] = (float)(x0 - sp_prune);
    o[5] = (float)(1.0 / wx);
    o[6] = (float)(y0 - sp_prune);
    o[7] = (float)(1.0 / wy);
    // prefilter records: way points as uint16 steps from the (rounded-down) lower corner of the box
    double extent = fmax(x1 - (double)o[0], y1 - (double)o[2]);
    if (!(extent >= 0.0) || extent > 3e38) extent = 0.0;   // (no way points)
    const float step = (float)fmax(extent / kQuantSteps, sp_prune / kQuantReach);
    o[kBoxQuantStep] = step;
    o[kBoxQuantStep + 1] = 1.0f / step;
    int shift = 0;
    while (((max_len - 1) >> shift) > 254) ++shift;   // (255 marks an empty bucket)
    reinterpret_cast<int32_t*>(o)[kBoxLenWord] = m;
    reinterpret_cast<int32_t*>(o)[kBoxLenWord + 1] = shift;
}

__global__ void path_bbox_kernel(const double* __restrict__ xyt, const int32_t* __restrict__ lens, int max_len,
                                 EntrySelect sel, double sp_prune, int buckets, double* __restrict__ bbox)
{
    const int64_t total = sel.size();
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (int64_t)gridDim.x * blockDim.x)
        path_bbox_one(xyt, lens, max_len, sel.entry(it), sp_prune, buckets, bbox);
}

// the rest of a path entry's record: origin of the entry's costmap (or of the shared one)
__global__ void world_record_kernel(EntrySelect sel, const double* __restrict__ origins, double ox, double oy,
                                    double* __restrict__ bbox)
{
    const int64_t total = sel.size();
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = sel.entry(it);
        double* o = bbox + kBoxDoubles * p;
        o[kBoxOrigin] = origins ? origins[2 * p] : ox;
        o[kBoxOrigin + 1] = origins ? origins[2 * p + 1] : oy;
    }
}

// index[p][axis][b] = {first, last} way point whose coordinate lies within sp of bucket b (widened by a guard band
// that swallows the rounding of the bucket computation); {32767, -1} when there is none.  Any way point with
// |x_j - x| <= sp_prune for a query x that falls into bucket b is inside [first, last].
__device__ __forceinline__ void path_index_one(const double* __restrict__ xyt, const int32_t* __restrict__ lens, int max_len,
                                               int64_t t, double sp_prune, const double* __restrict__ bbox,
                                               int16_t* __restrict__ index)
{
    const int b = (int)(t % kPathBuckets);
    const int axis = (int)((t / kPathBuckets) % 2);
    const int64_t p = t / (2 * kPathBuckets);
    const int m = lens ? lens[p] : max_len;
    const double* q = xyt + p * (int64_t)max_len * 3;
    const float* grid = reinterpret_cast<const float*>(bbox + kBoxDoubles * p) + 4;
    const double o = (double)grid[2 * axis], w = 1.0 / (double)grid[2 * axis + 1];
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

__global__ void path_index_kernel(const double* __restrict__ xyt, const int32_t* __restrict__ lens, int max_len,
                                  EntrySelect sel, double sp_prune, const double* __restrict__ bbox,
                                  int16_t* __restrict__ index)
{
    const int64_t total = sel.size() * 2 * kPathBuckets;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (int64_t)gridDim.x * blockDim.x)
        path_index_one(xyt, lens, max_len, sel.entry(it / (2 * kPathBuckets)) * (2 * kPathBuckets) + it % (2 * kPathBuckets),
                       sp_prune, bbox, index);
}

// The same tables for a private path, inside its record: kPathBucketsCompact buckets per axis, {first, last} >> shift as
// uint8 ({255, 0} when there is none).  One thread per bucket.
__global__ void path_index_compact_kernel(const double* __restrict__ xyt, const int32_t* __restrict__ lens, int max_len,
                                          EntrySelect sel, double sp_prune, double* __restrict__ bbox)
{
    const int64_t total = sel.size() * 2 * kPathBucketsCompact;
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(it % kPathBucketsCompact), axis = (int)((it / kPathBucketsCompact) % 2);
        const int64_t p = sel.entry(it / (2 * kPathBucketsCompact));
        const int m = lens ? lens[p] : max_len;
        const double* q = xyt + p * (int64_t)max_len * 3;
        const float* rec = reinterpret_cast<const float*>(bbox + kBoxDoubles * p);
        const int shift = reinterpret_cast<const int32_t*>(rec)[kBoxLenWord + 1];
        const double o = (double)rec[4 + 2 * axis], w = 1.0 / (double)rec[5 + 2 * axis];
        const double guard = 1e-6 * w + 1e-12;
        const double lo = o + b * w - sp_prune - guard, hi = o + (b + 1) * w + sp_prune + guard;
        int first = -1, last = -1;
        for (int j = 0; j < m; ++j) {
            const double v = q[3 * j + axis];
            if (v >= lo && v <= hi) {
                if (first < 0) first = j;
                last = j;
            }
        }
        const uint32_t packed = first < 0 ? 255u : (uint32_t)(first >> shift) | ((uint32_t)(last >> shift) << 8);
        reinterpret_cast<uint16_t*>(bbox + kBoxDoubles * p)[kBoxIndexU16 + axis * kPathBucketsCompact + b] = (uint16_t)packed;
    }
}

// find_last_reached restricted to j >= target (utilities/path_tools.py:408-448): the reward only asks whether the
// LAST reached index is >= target_idx (envs/base/reward.py:234), so indices below target never matter.
// A way point can only be reached when |x_j - x| and |y_j - y| are both below spatial_precision, so the scan is
// confined to the index window the two bucket tables allow for this pose (usually a handful of way points).
// index window [lo, hi] of the way points that can be within spatial_precision of (x, y); empty when lo > hi
struct PathWindow {
    int lo, hi;
};

// bbox: the eight leading values of a path record -- the f32 words of the record itself, or a copy widened to doubles
// (the LDS copy of a shared path); either way the arithmetic below is on the same doubles
template <typename BoxPtr, typename IndexPtr>
__device__ __forceinline__ PathWindow path_window(const DevParams& P, BoxPtr bbox, IndexPtr index, double x, double y)
{
    PathWindow w;
    w.lo = 0;
    w.hi = -1;
    if (x < (double)bbox[0] - P.sp_prune || x > (double)bbox[1] + P.sp_prune || y < (double)bbox[2] - P.sp_prune ||
        y > (double)bbox[3] + P.sp_prune)
        return w;  // farther than spatial_precision from the bounding box of the whole path
    const int bx = min(max((int)floor((x - (double)bbox[4]) * (double)bbox[5]), 0), kPathBuckets - 1);
    const int by = min(max((int)floor((y - (double)bbox[6]) * (double)bbox[7]), 0), kPathBuckets - 1);
    const IndexPtr ix = index + 2 * bx;
    const IndexPtr iy = index + 2 * (kPathBuckets + by);
    w.lo = max((int)ix[0], (int)iy[0]);
    w.hi = min((int)ix[1], (int)iy[1]);
    return w;
}

// the same look-up in the compact tables of a private path (tables: the record's uint16 words from kBoxIndexU16 on;
// shift: its int32 word kBoxLenWord + 1) -- a superset of the window the 64-bucket tables would give
template <typename BoxPtr, typename TablePtr>
__device__ __forceinline__ PathWindow path_window_compact(const DevParams& P, BoxPtr bbox, TablePtr tables, int shift, double x,
                                                          double y)
{
    PathWindow w;
    w.lo = 0;
    w.hi = -1;
    if (x < (double)bbox[0] - P.sp_prune || x > (double)bbox[1] + P.sp_prune || y < (double)bbox[2] - P.sp_prune ||
        y > (double)bbox[3] + P.sp_prune)
        return w;
    const int bx = min(max((int)floor((x - (double)bbox[4]) * (double)bbox[5]), 0), kPathBucketsCompact - 1);
    const int by = min(max((int)floor((y - (double)bbox[6]) * (double)bbox[7]), 0), kPathBucketsCompact - 1);
    const uint32_t ix = tables[bx], iy = tables[kPathBucketsCompact + by];
    w.lo = (int)max(ix & 0xFFu, iy & 0xFFu) << shift;
    w.hi = (int)(((min(ix >> 8, iy >> 8) + 1u) << shift) - 1u);
    return w;
}

// the window of entry g's path for a pose, whichever tables the path has (everything but the step kernels' hot paths)
__device__ __forceinline__ PathWindow path_window_of(const DevParams& P, bool shared, const double* bbox, const int16_t* index,
                                                     int64_t g, double x, double y)
{
    if (shared) return path_window(P, reinterpret_cast<const float*>(bbox), index, x, y);
    const float* rec = reinterpret_cast<const float*>(bbox + g * kBoxDoubles);
    return path_window_compact(P, rec, reinterpret_cast<const uint16_t*>(rec) + kBoxIndexU16,
                               reinterpret_cast<const int32_t*>(rec)[kBoxLenWord + 1], x, y);
}

// one candidate way point (its five values already in registers) against the three reach conditions of
// path_tools.py:419-427; they are independent predicates, so the cheap ones go first
__device__ __forceinline__ bool way_point_reached(const DevParams& P, double sx, double sy, double sth, double sc, double ss,
                                                  double x, double y, double th)
{
    const double dx = sx - x, dy = sy - y;
    if (fabs(dx) > P.sp_prune || fabs(dy) > P.sp_prune) return false;   // then hypot(dx,dy) >= sp
    const double par = sc * (x - sx) + ss * (y - sy);                   // path_tools.py:405
    if (!(par >= P.par_thr)) return false;
    const double q = dx * dx + dy * dy;
    bool near = q < P.sp2_lo;
    if (!near && q <= P.sp2_hi) near = hypot(dx, dy) < P.sp;            // too close to call from q
    return near && fabs(normalize_angle(th - sth)) < P.ap;
}

// way points in LDS (shared path): one candidate at a time, first hit from the top wins
// (four at a time, like the global-memory form below, measured no faster: the scan is not bound by the LDS round trip)
template <typename PathPtr>
__device__ __forceinline__ int last_reached_from(const DevParams& P, PathPtr path, PathWindow w, int m, int target,
                                                 double x, double y, double th, int stride = 1)
{
    if (target > m - 1) return -1;
    const int lo = max(w.lo, target);
    const int hi = min(w.hi, m - 1);
    for (int j = hi; j >= lo; j -= stride) {
        const PathPtr s = path + 5 * j;
        // all five values of the way point are fetched up front (one latency instead of three dependent ones)
        if (way_point_reached(P, s[0], s[1], s[2], s[3], s[4], x, y, th)) return j;
    }
    return -1;
}

// way points in global memory (private / pooled paths, cold after the kernel boundary): four candidates are fetched
// together, so the scan pays one memory round trip per four way points instead of one each
// (TRIP candidates per round trip: four hold 40 vector registers while they are in flight -- where the caller is short of
//  registers, the finishing code of the configurations with delay queues, two)
template <int TRIP = 4>
__device__ __forceinline__ int last_reached_from(const DevParams& P, const double* __restrict__ path, PathWindow w, int m,
                                                 int target, double x, double y, double th)
{
    if (target > m - 1) return -1;
    const int lo = max(w.lo, target);
    const int hi = min(w.hi, m - 1);
    for (int j = hi; j >= lo; j -= TRIP) {
        double v[TRIP][5];
#pragma unroll
        for (int u = 0; u < TRIP; ++u) {
            const double* s = path + 5 * max(j - u, lo);   // (below lo: a harmless repeat of way point lo)
#pragma unroll
            for (int k = 0; k < 5; ++k) v[u][k] = s[k];
        }
#pragma unroll
        for (int u = 0; u < TRIP; ++u)
            if (j - u >= lo && way_point_reached(P, v[u][0], v[u][1], v[u][2], v[u][3], v[u][4], x, y, th)) return j - u;
    }
    return -1;
}

// find_last_reached for ONE pose by all 64 lanes of a wave (way points in LDS): lane l takes candidate hi - l, the first
// lane that reports "reached" holds the last reached index.  Wave-uniform arguments and result.
__device__ __forceinline__ int coop_last_reached(const DevParams& P, LdsF64 path, PathWindow w, int m, int target, double x,
                                                 double y, double th)
{
    if (target > m - 1) return -1;
    const int lo = max(w.lo, target), hi = min(w.hi, m - 1);
    const int ln = (int)__lane_id();
    for (int top = hi; top >= lo; top -= 64) {
        const int j = top - ln;
        bool ok = false;
        if (j >= lo) {
            const LdsF64 s = path + 5 * j;
            ok = way_point_reached(P, s[0], s[1], s[2], s[3], s[4], x, y, th);
        }
        const uint64_t mask = __ballot(ok);
        if (mask) return top - ((int)__ffsll((unsigned long long)mask) - 1);
    }
    return -1;
}

// way points in global memory with the quantised prefilter record of private paths (PathDesc::pre, 8 bytes: x, y as uint16
// steps from the corner of the path's box, cos / sin of the heading as int16 / 32767; path_trig_kernel): eight candidates per
// trip cost eight 8-byte loads out of one or two 64-byte sectors (float32 records of 16 bytes until round 4: four per trip
// out of the same sectors; float64 before that: twenty 8-byte loads out of three per four candidates), and float32
// arithmetic; only a candidate the prefilter cannot rule out fetches its float64 record for the exact test.
// The prefilter works in steps: xr = (x - corner) / step is below 2^17 in magnitude for every pose whose window is not empty
// (the box test of path_window, step >= sp_prune / kQuantReach), so xr as a float32 is good to 2^-8 step and qx - xr to as much again;
// the reciprocal of the step as float32 adds 6e-8 * 2^17 < 0.01 step, the way point's own rounding 0.5: each coordinate
// difference is within 0.52 step of the true one, the distance within 0.74 -- the position limits carry +1.  "Not behind the
// way point" (path_tools.py:405): cos and sin are off by 1.53e-5 each, times a difference of at most kQuantReach + 1 steps =
// 0.25 step per term, plus 0.52 (|cos| + |sin|) <= 0.74 for the differences and ~0.002 of float32 rounding: below 1.3 -- the
// limit carries -2.  The heading (path_tools.py:419, |normalize(th - th_j)| < ap  <=>  cos(th - th_j) > cos ap for ap < pi): the
// cosine of the difference from the record's cos / sin and float32 cos / sin of the pose's heading is within 2.5e-5 of the true
// one -- its limit, DevParams::ap_cos_min, carries -1e-4.  (Without it a pose that is near a way point but turned away from it
// went through the exact test of every such way point, one memory round trip each: the waves that scan the part of the window
// next to the target took 9 k cycles longer than the others on C4.)
// So whatever the prefilter rejects fails the float64 test too: the result is the exact scan's, bit for bit.
#ifndef BCP_PREFILTER_TRIP
#define BCP_PREFILTER_TRIP 8
#endif
__device__ __forceinline__ int last_reached_prefiltered(const DevParams& P, const double* __restrict__ path,
                                                        const uint32_t* __restrict__ pre, PathWindow w, int m, int target,
                                                        double x, double y, double th, double ox, double oy, float inv_step)
{
    if (target > m - 1) return -1;
    const int lo = max(w.lo, target);
    const int hi = min(w.hi, m - 1);
    const double inv = (double)inv_step;
    const float xr = (float)((x - ox) * inv), yr = (float)((y - oy) * inv);
    const float lim = (float)(P.sp * inv) + 1.0f, par_min = (float)(P.par_thr * inv) - 2.0f;
    const float q_max = lim * lim * 1.000001f;
    float sin_th, cos_th;
    sincosf((float)th, &sin_th, &cos_th);
    const float cos_min = P.ap_cos_min * 32767.0f;
    constexpr int TRIP = BCP_PREFILTER_TRIP;
    for (int j = hi; j >= lo; j -= TRIP) {
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        u32x2 p[TRIP];
#pragma unroll
        for (int u = 0; u < TRIP; ++u) p[u] = as_global(reinterpret_cast<const u32x2*>(pre))[max(j - u, lo)];   // (below lo: a repeat of lo)
        // the candidates the prefilter lets through, as a bit mask (straight-line code), then the exact test from the top
        uint32_t maybe = 0;
#pragma unroll
        for (int u = 0; u < TRIP; ++u) {
            const float dx = (float)(p[u].x & 0xFFFFu) - xr, dy = (float)(p[u].x >> 16) - yr;
            const float qc = (float)(int16_t)(p[u].y & 0xFFFFu), qs = (float)((int32_t)p[u].y >> 16);
            const float par = -(qc * dx + qs * dy) * (1.0f / 32767.0f);
            const bool out = fabsf(dx) > lim || fabsf(dy) > lim || dx * dx + dy * dy > q_max || par < par_min ||
                             qc * cos_th + qs * sin_th < cos_min || j - u < lo;
            maybe |= (uint32_t)!out << u;
        }
        while (maybe) {
            const int u = (int)__builtin_ctz(maybe);
            maybe &= maybe - 1;
            const double* s = path + 5 * (j - u);
            if (way_point_reached(P, s[0], s[1], s[2], s[3], s[4], x, y, th)) return j - u;
        }
    }
    return -1;
}

template <int TRIP = 4, typename PathPtr>
__device__ __forceinline__ double reward_step(const DevParams& P, PathPtr path, PathWindow w, int m, double x, double y,
                                              double th, double& min_dist, int& target)
{
    if (target > m - 1) return 0.0;
    int last;
    if constexpr (std::is_same<PathPtr, const double*>::value) last = last_reached_from<TRIP>(P, path, w, m, target, x, y, th);
    else last = last_reached_from(P, path, w, m, target, x, y, th);
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

// reward_step with the scan already done (`last` = last_reached_from's result for this pose and target): the rest of
// ContinuousRewardProvider.reward (envs/base/reward.py:234-259)
template <typename PathPtr>
__device__ __forceinline__ double reward_from_last(const DevParams& P, PathPtr path, int m, int last, double x, double y,
                                                   double& min_dist, int& target)
{
    if (target > m - 1) return 0.0;
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

// The same in two halves: fifo_peek() reads, before anything is pushed, the element the k-th push will hand back --
// for k > delay the one it displaces, for 1 < k <= delay the list's first element (slot 0) -- and fifo_push() stores
// the new element and returns the peeked one (the new element itself for k = 1).  So what State exposes at a step
// (k > 1) is known before the robot model runs, and pushing twice (kernel 1 optimistically, kernel 2 again with the
// rolled-back pose after a collision) lands in the same slot and hands back the same element.
template <int W>
__device__ __forceinline__ void fifo_peek(const double* __restrict__ q, int delay, int64_t n, int64_t i, int k, double (&peeked)[W])
{
    if (delay <= 0 || k <= 1) return;
    const double* cell = q + ((int64_t)(k > delay ? (k - 1) % delay : 0) * W) * n + i;
#pragma unroll
    for (int c = 0; c < W; ++c) peeked[c] = cell[c * n];
}

template <int W>
__device__ __forceinline__ void fifo_push(double* __restrict__ q, int delay, int64_t n, int64_t i, int k, double (&v)[W],
                                          const double (&peeked)[W])
{
    if (delay <= 0) return;
    double* cell = q + ((int64_t)((k - 1) % delay) * W) * n + i;
#pragma unroll
    for (int c = 0; c < W; ++c) {
        cell[c * n] = v[c];
        if (k > 1) v[c] = peeked[c];
    }
}

constexpr int kWavePerPoseFrom = 1;   // step_pending_kernel: more than this many rounds of parked poses per team -> a pose per wave
#ifdef BCP_DIAG
constexpr int kDiagBlocks = 4096;   // stamps of the first kDiagBlocks workgroups of a step kernel, 16 slots each
__device__ unsigned long long g_diag[kDiagBlocks * 16];
#define DIAG_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 512) g_diag[blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)   /* (rows 512 .. 2047 hold the per-wave tables) */
// lane 0 of wave `w` of the workgroup
#define DIAG_STAMP_W(w, k) do { if (threadIdx.x == (w) * 64 && blockIdx.x < kDiagBlocks) g_diag[blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
// maxima over the waves of a workgroup live in the upper half of the array (slot k of workgroup b: (2048 + b) * 16 + k)
#define DIAG_MAX(k, v) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 2048) atomicMax(&g_diag[(2048 + blockIdx.x) * 16 + (k)], (unsigned long long)(v)); } while (0)
#define DIAG_NOW() __builtin_amdgcn_s_memtime()
// wall-clock stamps (s_memrealtime: 100 MHz, the same counter chip-wide -- s_memtime is not comparable between compute units):
// rows 3072 + workgroup (grids of up to 1024 workgroups), slot 0 = earliest entry of a wave, slot 1 = latest exit
#define DIAG_REAL_ENTRY() do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024) atomicMax(&g_diag[(3072 + blockIdx.x) * 16 + 0], ~(unsigned long long)__builtin_amdgcn_s_memrealtime()); } while (0)   /* (the complement: bcp_diag_clear zeroes) */
#define DIAG_REAL_EXIT() do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024) atomicMax(&g_diag[(3072 + blockIdx.x) * 16 + 1], (unsigned long long)__builtin_amdgcn_s_memrealtime()); } while (0)
// further stamps of lane 0 of wave `w`, in the upper half next to the maxima: slot k = 1 .. 15 of workgroup b at (2048 + b) * 16 + k
#define DIAG_STAMP_U(w, k) do { if (threadIdx.x == (w) * 64 && blockIdx.x < 2048) g_diag[(2048 + blockIdx.x) * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define DIAG_WAIT_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// lane 0 of EVERY wave: row (base + workgroup), slot = wave number (base = 1024, 1536: arrival at barrier 0 / 1)
#define DIAG_STAMP_WAVES(base) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 256) g_diag[((base) + blockIdx.x) * 16 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memtime(); } while (0)
// maximum over ALL lanes of the workgroup (not only lane 0 of each wave)
#define DIAG_MAX_ALL(k, v) do { if (blockIdx.x < 2048) atomicMax(&g_diag[(2048 + blockIdx.x) * 16 + (k)], (unsigned long long)(v)); } while (0)
extern "C" int bcp_diag_read(unsigned long long* out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_diag), sizeof(g_diag));
}
extern "C" int bcp_diag_clear()
{
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_diag)) != hipSuccess) return -1;
    return (int)hipMemset(p, 0, sizeof(g_diag));
}

#else
#define DIAG_STAMP(k) do { } while (0)
#define DIAG_STAMP_W(w, k) do { } while (0)
#define DIAG_MAX(k, v) do { } while (0)
#define DIAG_NOW() 0ull
#define DIAG_REAL_ENTRY() do { } while (0)
#define DIAG_REAL_EXIT() do { } while (0)
#define DIAG_STAMP_U(w, k) do { } while (0)
#define DIAG_WAIT_VMEM() do { } while (0)
#define DIAG_STAMP_WAVES(base) do { } while (0)
#define DIAG_MAX_ALL(k, v) do { } while (0)
#endif


// dynamic LDS of the collision kernels:
//   [lethal bitmap words (when the shared map is staged)] [qverts: n_verts * 2 doubles] [vertex scratch of the
//   per-thread rasteriser: n_verts * 2 * kBlock words]
extern __shared__ uint32_t lds_dyn[];

struct CollisionLds {
    bool staged;      // the shared lethal bitmap sits at LDS offset 0
    LdsWords bits;
    LdsF64 qverts;
    VertLds scratch;
    LdsU32 cells;     // kSparseCap words: the cell list of coop_collides_sparse (exact mode 3)
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
    L.cells = lds + q_off + 4 * P.n_verts + 2 * P.n_verts * kBlock;   // kSparseLdsWords
    __syncthreads();
    return L;
}

static size_t collision_lds_bytes(int n_verts, int in_lds, int rows, int wpr)
{
    size_t words = in_lds ? (size_t)rows * wpr : 0;
    words = (words + 1) & ~(size_t)1;
    words += 4 * (size_t)n_verts;            // qverts (doubles)
    words += 2 * (size_t)n_verts * kBlock;   // per-thread vertex scratch
    words += kSparseLdsWords;                // cell list of the cell-by-cell exact test
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
    double c, s;
    cos_sin(th, c, s);
    int cls = active ? classify(cull, map.shared ? 0 : env, map.rows, map.cols, px, py, c, s) : kFree;
    bool hit = cls == kHit;
    uint64_t amb = __ballot(cls == kAmbiguous);
    if (amb == 0) return hit;
    const bool dense = exact_mode == 2 || (exact_mode == 0 && (int)__popcll(amb) > dense_threshold);   // (modes 1 and 3: cooperative)
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
        const int64_t env_ = ((int64_t)bcast_i((int)(env >> 32), src) << 32) | (uint32_t)bcast_i((int)env, src);
        const uint32_t* words = map.bits + (map.shared ? 0 : env_ * map.env_stride);
        int verdict = kSparseTooMany;
        if (exact_mode == 3) {   // the cell-by-cell form of the exact test (what step_local_kernel uses)
            if (L.staged)
                verdict = wide ? coop_collides_sparse<true>(P, L.qverts, c_, s_, px_, py_, L.bits, map.rows, map.cols, map.wpr, L.cells)
                               : coop_collides_sparse<false>(P, L.qverts, c_, s_, px_, py_, L.bits, map.rows, map.cols, map.wpr, L.cells);
            else
                verdict = wide ? coop_collides_sparse<true>(P, L.qverts, c_, s_, px_, py_, words, map.rows, map.cols, map.wpr, L.cells)
                               : coop_collides_sparse<false>(P, L.qverts, c_, s_, px_, py_, words, map.rows, map.cols, map.wpr, L.cells);
        }
        if (verdict != kSparseTooMany) {
            h = verdict == kSparseHit;
        } else if (L.staged) {
            h = coop_collides(P, vqx, vqy, c_, s_, px_, py_, L.bits, map.rows, map.cols, map.wpr, wide);
        } else {
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
    // delays > 0: the elements this step's pushes into the pose / robot-state FIFOs will hand back (fifo_peek: read with
    // the state, before anything is pushed), so that kernel 2 can redo the pushes of an env kernel 1 finished optimistically
    double popped_pose[3], popped_state[7];
};

// An env's initial state (entry k of the initial-state arrays), all loads in flight together
struct InitAhead {
    double x, y, th, v, w, steer, wheel, min_dist;
    int32_t target, iter, collided;
};

template <typename SP>
__device__ __forceinline__ InitAhead fetch_init_ahead(SP S, int64_t k, bool tri)
{
    InitAhead p;
    p.x = as_global(S->init.x)[k];
    p.y = as_global(S->init.y)[k];
    p.th = as_global(S->init.angle)[k];
    p.v = as_global(S->init.v)[k];
    p.w = as_global(S->init.w)[k];
    p.steer = tri ? as_global(S->init.steer)[k] : 0.0;
    p.wheel = tri ? as_global(S->init.wheel)[k] : 0.0;
    p.min_dist = as_global(S->init.min_dist)[k];
    p.target = as_global(S->init.target_idx)[k];
    p.iter = as_global(S->init.cur_iter)[k];
    p.collided = (int32_t)as_global(S->init.collided)[k];
    return p;
}

// reward-provider outcome for the pose as it is when nothing collides, computed ahead of the collision verdict
struct ScoredFree {
    double rew, min_dist;
    int target;
};

// A parked pose of the single-launch step (LDS): what the exact test needs, and its verdict.  The env's state stays in
// the registers of the mover lane that parked it; that lane finishes the env once the verdict is in.
struct ParkedPose {
    double c, s;             // cos / sin of the new heading
    int32_t px, py;          // world_to_pixel of the new position
    int32_t env_lo, env_hi;  // env index (private maps: the map entry)
    int32_t geom;            // geometry-pool entry
    int32_t verdict;         // 0 = pending, 1 = free, 2 = collides
    // private paths: what the reward provider returns for the rolled-back pose, should the verdict be "collides" -- worked
    // out by the env's mover while the scanners are still busy (step_local_kernel), valid when spec_ok != 0
    double spec_rew, spec_min;
    int32_t spec_target, spec_ok;
};

// entry of the non-shared map / path arrays that env i uses
template <typename SP>   // const StepStatic* in global memory, or step_local_kernel's copy of the block in LDS
__device__ __forceinline__ int64_t slot_of(SP S, int64_t i, const Pending& q)
{
    return S->geom_of_env ? (int64_t)q.geom : i;
}

// Everything of PlanEnv.step() that follows pose_collides(): rollback (env.py:458-459), bookkeeping and delay queues
// (:377-396), reward (:352), done (:407-419), outputs, optional reset, state write-back.  Runs on one lane for env i.
// PLAIN = true (no delays, continuous reward provider -- see step_is_plain) compiles the delay queues and the
// pure-pursuit branch out.
// (A: StepArgs by value, or by reference into the kernel-argument segment -- only a.S, the output pointers and a.flags are used)
template <bool PLAIN, typename A, typename SP>
__device__ __forceinline__ void finalize_env_from(const A& a, SP S, int64_t i, Pending& q, bool hit, LdsF64 lds_path = nullptr,
                                             const PathWindow* free_window = nullptr, bool have_score = false,
                                             ScoredFree score = ScoredFree(), int known_len = -1, bool score_fits_hit = false,
                                             int64_t out_base = 0)
{   // (the score travels by value: a pointer to a local made the compiler keep it in scratch memory;
    //  out_base: element offset of this step's rows in the output arrays -- step k of a rollout writes row k, bcp_rollout --;
    //  known_len >= 0: the caller already holds the length of this env's path;
    //  score_fits_hit: `score` was computed for the rolled-back pose of a colliding env -- continuous provider only)
    const DevParams& P = *(const DevParams*)&S->P;   // (S may point into LDS: step_local_kernel)
    const int pose_delay = PLAIN ? 0 : P.pose_delay, state_delay = PLAIN ? 0 : P.state_delay;
    const bool pure_pursuit = !PLAIN && P.reward_provider == BCP_REWARD_PURE_PURSUIT;
    const bool tri = P.model == BCP_MODEL_TRICYCLE;
    const int64_t n = S->n;
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
    if (pose_delay) fifo_push<3>(S->st.pose_q, pose_delay, n, i, iter, seen, q.popped_pose);
    if (state_delay) fifo_push<7>(S->st.state_q, state_delay, n, i, iter, seen_rs, q.popped_state);

    // shared path: uniform pointers (scalar cache); private paths: per-lane pointers
    double rew = 0.0;
    int m;
    bool goal;
    const int64_t g = slot_of(S, i, q);
    const double* pts = S->path.pts + (S->path.shared ? 0 : g * (int64_t)S->path.max_len * 5);
    m = known_len >= 0 ? known_len : (S->path.shared ? S->path.max_len : S->path.lens[g]);
    if (pure_pursuit) {
        if (have_score && !hit) {   // the scorer wave has already done it (no collision: the flag it assumed stands)
            rew = score.rew;
            min_dist = score.min_dist;
            target = score.target;
        } else if (!ABLATED(a, kAblateNoReward)) {
            rew = reward_pure_pursuit(pts, m, seen[0], seen[1], collided, min_dist, target);
        }
        goal = hypot(pts[5 * (m - 1)] - seen[0], pts[5 * (m - 1) + 1] - seen[1]) < 1.0;   // done(), reward.py:141-150
    } else {
        // the scorer wave has already done it (step_fast_pair_kernel) for the pose State exposes if nothing collides --
        // which, with a pose delay, is an earlier pose whatever this step's verdict (from the second step of an episode on)
        if (have_score && (!hit || score_fits_hit || (pose_delay && iter > 1))) {
            rew = score.rew;
            min_dist = score.min_dist;
            target = score.target;
        } else if (!ABLATED(a, kAblateNoReward)) {
            // way-point window of the pose: the caller may have looked it up already for the un-rolled-back pose
            const PathWindow w =
                (free_window && !hit && !pose_delay) ? *free_window : path_window_of(P, S->path.shared != 0, S->path.bbox, S->path.index, g, seen[0], seen[1]);
            if (lds_path && S->path.shared)  // way points staged in LDS by the step kernel
                rew = reward_step(P, lds_path, w, m, seen[0], seen[1], seen[2], min_dist, target);
            else
                rew = reward_step<PLAIN ? 4 : 2>(P, pts, w, m, seen[0], seen[1], seen[2], min_dist, target);
        }
        goal = target > m - 1;
    }
    const bool done = goal || (iter >= P.iteration_timeout) || collided;

    const int64_t o = out_base + i;
    as_global(a.reward)[o] = rew;
    as_global(a.done)[o] = (uint8_t)done;
    if (a.collided_now) as_global(a.collided_now)[o] = (uint8_t)hit;
    if (a.err) as_global(a.err)[o] = q.err;
    if (a.noise_z_out) {
        const double nan = __builtin_nan("");
        as_global(a.noise_z_out)[3 * o + 0] = (q.drawn & 1) ? q.z[0] : nan;
        as_global(a.noise_z_out)[3 * o + 1] = (q.drawn & 2) ? q.z[1] : nan;
        as_global(a.noise_z_out)[3 * o + 2] = (q.drawn & 4) ? q.z[2] : nan;
    }

    if (done && (a.flags & BCP_STEP_AUTO_RESET)) {  // PlanEnv.reset(): set_state(initial_state) (env.py:293-303)
        int64_t k = i;
        if (S->geom_of_env) {  // RandomMiniEnv.reset(): the env moves on to its next geometry (mini_env.py:469-481).
            // Computed from the entry the step started with, so kernel 2 redoing an env that kernel 1 already reset
            // lands on the same geometry (a hit always ends the episode, so both reset or neither does).
            k = S->next_geom ? as_global(S->next_geom)[g] : g;
            as_global(S->geom_of_env)[i] = (int32_t)k;
        }
        const InitAhead ahead = fetch_init_ahead(S, k, tri);
        r.p.x = ahead.x;
        r.p.y = ahead.y;
        r.p.th = ahead.th;
        r.v = ahead.v;
        r.w = ahead.w;
        if (tri) {
            r.steer = ahead.steer;
            r.wheel = ahead.wheel;
        }
        min_dist = ahead.min_dist;
        target = ahead.target;
        iter = ahead.iter;
        collided = ahead.collided != 0;
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
    // (the state pointers come with the kernel arguments -- StepHot -- not through *S: one dependent load less per array)
    as_global(a.hot.st.x)[i] = r.p.x;
    as_global(a.hot.st.y)[i] = r.p.y;
    as_global(a.hot.st.angle)[i] = r.p.th;
    as_global(a.hot.st.v)[i] = r.v;
    as_global(a.hot.st.w)[i] = r.w;
    if (tri) {
        as_global(a.hot.st.steer)[i] = r.steer;
        as_global(a.hot.st.wheel)[i] = r.wheel;
    }
    as_global(a.hot.st.min_dist)[i] = min_dist;
    as_global(a.hot.st.target_idx)[i] = target;
    as_global(a.hot.st.cur_iter)[i] = iter;
    as_global(a.hot.st.collided)[i] = (uint8_t)collided;
    if (pose_delay) {
#pragma unroll
        for (int c = 0; c < 3; ++c) S->st.pose_seen[c * n + i] = seen[c];
    }
    if (state_delay) {
#pragma unroll
        for (int c = 0; c < 7; ++c) S->st.state_seen[c * n + i] = seen_rs[c];
    }
}

// (the parameter block where the launch arguments point to: the two-launch and the general step kernels)
template <bool PLAIN, typename A>
__device__ __forceinline__ void finalize_env(const A& a, int64_t i, Pending& q, bool hit, LdsF64 lds_path = nullptr,
                                             const PathWindow* free_window = nullptr, bool have_score = false,
                                             ScoredFree score = ScoredFree(), int known_len = -1, bool score_fits_hit = false)
{
    finalize_env_from<PLAIN>(a, a.S, i, q, hit, lds_path, free_window, have_score, score, known_len, score_fits_hit);
}

// ---- loads shared by the step kernels ------------------------------------------------------------------------
template <bool PLAIN, typename A>
__device__ __forceinline__ void load_env(const A& a, uint64_t seed, uint64_t step_counter, int64_t i, bool active,
                                         Pending& q, double& cmd0, double& cmd1, bool draw_noise = true)
{
    // (no branch in here: the pointers are adjacent launch arguments, fetched by a few wide scalar loads in one go; an
    //  array a configuration does not have -- the tricycle's two values, the pool entry -- is read from a stand-in)
    Robot& r = q.r;
    const bool tri = a.hot.model == BCP_MODEL_TRICYCLE;
    const double* const p_steer = tri ? a.hot.st.steer : a.hot.st.x;
    const double* const p_wheel = tri ? a.hot.st.wheel : a.hot.st.x;
    const int32_t* const p_geom = a.hot.geom_of_env ? a.hot.geom_of_env : a.hot.st.target_idx;
    r.p.x = as_global(a.hot.st.x)[i];
    r.p.y = as_global(a.hot.st.y)[i];
    r.p.th = as_global(a.hot.st.angle)[i];
    r.v = as_global(a.hot.st.v)[i];
    r.w = as_global(a.hot.st.w)[i];
    const double v_steer = as_global(p_steer)[i], v_wheel = as_global(p_wheel)[i];
    q.min_dist = as_global(a.hot.st.min_dist)[i];
    q.target = as_global(a.hot.st.target_idx)[i];
    q.iter = as_global(a.hot.st.cur_iter)[i];
    q.collided = as_global(a.hot.st.collided)[i] != 0;
    const int32_t v_geom = as_global(p_geom)[i];
    r.steer = tri ? v_steer : 0.0;
    r.wheel = tri ? v_wheel : 0.0;
    q.geom = a.hot.geom_of_env ? v_geom : 0;
    // the action, float32 [N,2] or float64 [N,2], as two 8-byte words read without a branch (float32: the same word twice);
    // they are only looked at below, so the loads above and the caller's next ones are all in flight together
    const bool f32 = (a.flags & BCP_STEP_ACTIONS_F32) != 0;
    const GlobalPtr<const uint64_t> aw = as_global(reinterpret_cast<const uint64_t*>(a.actions)) + (f32 ? i : 2 * i);
    const uint64_t a_lo = aw[0], a_hi = aw[f32 ? 0 : 1];
    cmd0 = f32 ? (double)__uint_as_float((uint32_t)a_lo) : __longlong_as_double((long long)a_lo);
    cmd1 = f32 ? (double)__uint_as_float((uint32_t)(a_lo >> 32)) : __longlong_as_double((long long)a_hi);
    if (!PLAIN && a.hot.control_delay && active) {   // the robot executes the command given control_delay steps ago (env.py:371-373)
        double cmd[2] = {cmd0, cmd1};
        fifo_delay<2>(a.hot.st.control_q, a.hot.control_delay, a.hot.n, i, q.iter + 1, cmd);
        cmd0 = cmd[0];
        cmd1 = cmd[1];
    }
    if (!PLAIN) {   // what this step's pushes will displace (k = iter + 1)
        fifo_peek<3>(a.hot.st.pose_q, a.hot.pose_delay, a.hot.n, i, q.iter + 1, q.popped_pose);
        fifo_peek<7>(a.hot.st.state_q, a.hot.state_delay, a.hot.n, i, q.iter + 1, q.popped_state);
    }
    q.z[0] = q.z[1] = q.z[2] = 0.0;
    if (a.hot.noise_on) {
        if (a.noise_z) {
            q.z[0] = as_global(a.noise_z)[3 * i + 0];
            q.z[1] = as_global(a.noise_z)[3 * i + 1];
            q.z[2] = as_global(a.noise_z)[3 * i + 2];
        } else if (draw_noise) {
            device_normals(seed, (uint64_t)(a.hot.env_id_base + i), step_counter, q.z);
        }
    }
}

template <bool PLAIN>
__device__ __forceinline__ void load_env(const StepArgs& a, int64_t i, bool active, Pending& q, double& cmd0, double& cmd1)
{
    load_env<PLAIN>(a, a.seed, a.step_counter, i, active, q, cmd0, cmd1);
}

// General step kernel: robot model, collision settled in place by collides_wave (distance-field classification when
// there is one, then the cooperative / per-thread exact rasterisers), reward, done, write-back.  Used when there
// is no distance field, when the batch is too small to need load balancing, or when a mode is forced.
__global__ void __launch_bounds__(kBlock) step_kernel(const StepArgs launch_args)
{
    const StepArgs a = resolve_step(launch_args, 0);
    const DevParams& P = a.S->P;
    const int tid = threadIdx.x;
    const int64_t gi = (int64_t)blockIdx.x * kBlock + tid;
    const bool active = gi < a.hot.n;
    const int64_t i = active ? gi : a.hot.n - 1;  // inactive lanes of the last wave shadow env n-1 and never store

    const CollisionLds L = collision_lds_setup(P, a.S->map, tid);
    Pending q;
    double cmd0, cmd1;
    load_env<false>(a, i, active, q, cmd0, cmd1);

    // ---- _env_step (envs/base/env.py:442-461)
    q.old = q.r.p;
    q.drawn = 0;
    q.err = robot_step(P, q.r, cmd0, cmd1, q.z, q.drawn);
    bool hit = false;
    if (!ABLATED(a, kAblateNoCollision))
        hit = collides_wave(P, a.S->map, a.S->cull, L, a.S->exact_mode, a.S->dense_threshold, a.S->wide != 0, active,
                            slot_of(a.S, i, q), q.r.p.x, q.r.p.y, q.r.p.th);
    if (active) finalize_env<false>(a, i, q, hit);
    advance_step_by_ticket(a);
}


// Kernel 1 of the two-kernel step (needs a distance field).  A pose is cleared in O(1) by the outer test; the few
// envs it cannot clear are finished optimistically ("free") AND parked in `pending`, and kernel 2 redoes those that
// really collide.  Waves with many undecided lanes (robots hugging walls) settle them in place.  Memory operations are
// grouped so that independent round trips overlap: a wave has at most one partner on its SIMD, so an exposed L2 / HBM
// latency is nearly pure stall.
//
// It runs with TWO wavefronts per 64 envs.  A wave that runs alone on its SIMD is bound by the latency of its
// own dependent chain, and the two longest stretches after the robot model -- collision classification + parking on
// one side, the reward scan on the other -- only share the new pose.  So the "mover" wave (loads, robot model,
// classification, parking / in-place settling, write-back) hands the pose to the "scorer" wave through LDS, the
// scorer runs the reward provider for the free pose meanwhile, and the mover picks the result up for every env that
// did not collide in place (those redo the reward themselves for the rolled-back pose).
// LDS: [qverts][shared path][64 x {x, y, theta}][64 x {reward, min_dist, target}]
// PLAIN = false: delay queues and / or the pure-pursuit provider (finalize_env's general form; the scorer wave's
// result is then only used where it applies: continuous reward, no pose delay).
template <bool WIDE, bool PLAIN>
__global__ void __launch_bounds__(2 * kBlock) step_fast_pair_kernel(const StepArgs launch_args)
{
    const StepArgs a = resolve_step(launch_args, 0);
    const DevParams& P = a.S->P;
    const int tid = threadIdx.x, lane = tid & 63;
    const bool mover = tid < kBlock;
    const int64_t gi = (int64_t)blockIdx.x * kBlock + lane;
    const bool active = gi < a.hot.n;
    const int64_t i = active ? gi : a.hot.n - 1;

    // (1) staging loads (both waves), then the mover's state / action / noise and the scorer's two reward-state words
    __attribute__((address_space(3))) double* qv = (__attribute__((address_space(3))) double*)lds_dyn;
    const int nq = 2 * P.n_verts;
    const double my_q = tid < nq ? P.qverts[tid >> 1][tid & 1] : 0.0;
    double t[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int k = u * 2 * kBlock + tid;
        t[u] = k < a.hot.lds_path_doubles ? a.hot.path_pts[k] : 0.0;
    }
    Pending q;
    double cmd0 = 0.0, cmd1 = 0.0;
    if (mover) {
        load_env<PLAIN>(a, i, active, q, cmd0, cmd1);
        if (gi < kShards) a.pending_next[gi] = 0;  // arm the counters of the NEXT step (the two sets alternate)
        if (gi < kShards && a.inplace_next) a.inplace_next[gi] = 0;
    } else {
        q.min_dist = a.hot.st.min_dist[i];
        q.target = a.hot.st.target_idx[i];
        q.geom = a.hot.geom_of_env ? a.hot.geom_of_env[i] : 0;
        q.collided = PLAIN ? 0 : (int32_t)(a.hot.st.collided[i] != 0);   // (the pure-pursuit reward depends on it)
    }
    // the scorer also brings the bounding box and the bucket index of a shared path into LDS while the mover is busy
    // with the robot model (its window look-up then needs no global round trip); a private path's box is fetched now
    // as well -- it does not depend on the pose
    float box[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t index_words[2] = {0, 0};
    if (!mover) {
        if (a.hot.path_shared) {
            if (lane < 8) box[0] = reinterpret_cast<const float*>(a.hot.path_bbox)[lane];
            const uint32_t* iw = reinterpret_cast<const uint32_t*>(a.hot.path_index);   // [2][kPathBuckets][2] int16
            index_words[0] = iw[lane];
            index_words[1] = iw[kBlock + lane];
        } else {
            const int64_t g = a.hot.geom_of_env ? (int64_t)q.geom : i;
#pragma unroll
            for (int k = 0; k < 8; ++k) box[k] = reinterpret_cast<const float*>(a.hot.path_bbox + g * kBoxDoubles)[k];
        }
    }
    if (tid < nq) qv[tid] = my_q;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int k = u * 2 * kBlock + tid;
        if (k < a.hot.lds_path_doubles) qv[nq + k] = t[u];
    }
    for (int k = 512 + tid; k < a.hot.lds_path_doubles; k += 2 * kBlock) qv[nq + k] = a.hot.path_pts[k];  // long paths
    const LdsF64 lds_path = a.hot.lds_path_doubles ? (LdsF64)(qv + nq) : (LdsF64) nullptr;
    __attribute__((address_space(3))) double* hand_pose = qv + nq + a.hot.lds_path_doubles;
    __attribute__((address_space(3))) double* hand_score = hand_pose + 3 * kBlock;
    __attribute__((address_space(3))) double* lds_box = hand_score + 3 * kBlock;                    // [8]
    __attribute__((address_space(3))) uint32_t* lds_index = (__attribute__((address_space(3))) uint32_t*)(lds_box + 8);  // [128]
    if (!mover && a.hot.path_shared) {
        if (lane < 8) lds_box[lane] = (double)box[0];
        lds_index[lane] = index_words[0];
        lds_index[kBlock + lane] = index_words[1];
    }

    // (2) mover: _env_step's robot model (envs/base/env.py:442-461); the new pose goes to the scorer
    Robot& r = q.r;
    if (mover) {
        q.old = r.p;
        q.drawn = 0;
        q.err = robot_step(P, r, cmd0, cmd1, q.z, q.drawn);
        // the pose the reward provider will see: the new one, or -- with a pose delay -- the one fifo_peek fetched
        const bool delayed = !PLAIN && P.pose_delay > 0 && q.iter + 1 > 1;
        hand_pose[lane] = delayed ? q.popped_pose[0] : r.p.x;
        hand_pose[kBlock + lane] = delayed ? q.popped_pose[1] : r.p.y;
        hand_pose[2 * kBlock + lane] = delayed ? q.popped_pose[2] : r.p.th;
    }
    __syncthreads();

    bool hit = false;
    if (mover) {
        // (3a) collision: distance-field classification, then parking or in-place settling
        const int64_t g = slot_of(a.S, i, q);
        double ox = a.S->map.ox, oy = a.S->map.oy;
        if (a.S->map.origins) {
            ox = a.S->map.origins[2 * g + 0];
            oy = a.S->map.origins[2 * g + 1];
        }
        const int px = (int)rint((r.p.x - ox) * a.S->map.inv_res);  // world_to_pixel, coordinate_transformations.py:185-205
        const int py = (int)rint((r.p.y - oy) * a.S->map.inv_res);
        double c, s;
        cos_sin(r.p.th, c, s);
        const int64_t map_env = a.S->map.shared ? 0 : g;
        OuterLookups look;
        look.off_map = true;
        if (!ABLATED(a, kAblateNoCollision | kAblateNoClassify))
            look = outer_lookups_issue(a.S->cull, map_env, a.S->map.rows, a.S->map.cols, px, py, c, s);
        const int cls = active ? outer_lookups_verdict(a.S->cull, look) : kFree;
        const uint64_t amb = __ballot(cls == kAmbiguous);
        const int n_amb = (int)__popcll(amb);
        const int threshold = a.threshold_now ? *a.threshold_now : a.S->dense_threshold;
        if (n_amb > threshold) {
            if (lane == 0 && a.inplace_count) atomicAdd(a.inplace_count + (int)(blockIdx.x % kShards), n_amb);
            const bool inner = cls == kAmbiguous && classify_inner_hit(a.S->cull, map_env, px, py, c, s);
            hit = inner;
            uint64_t todo = __ballot(cls == kAmbiguous && !inner);
            const double vqx = lane < P.n_verts ? qv[2 * lane] : 0.0, vqy = lane < P.n_verts ? qv[2 * lane + 1] : 0.0;
            while (todo) {
                const int src = __ffsll((unsigned long long)todo) - 1;
                todo &= todo - 1;
                const int64_t env_ = ((int64_t)bcast_i((int)(g >> 32), src) << 32) | (uint32_t)bcast_i((int)g, src);
                const uint32_t* words = a.S->map.bits + (a.S->map.shared ? 0 : env_ * a.S->map.env_stride);
                const bool h = coop_collides<WIDE>(P, vqx, vqy, bcast_d(c, src), bcast_d(s, src), bcast_i(px, src),
                                                   bcast_i(py, src), words, a.S->map.rows, a.S->map.cols, a.S->map.wpr);
                if (lane == src) hit = h;
            }
        } else if (cls == kAmbiguous && !ABLATED(a, kAblateNoPark)) {
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
    } else if (!ABLATED(a, kAblateNoReward)) {
        // (3b) scorer: ContinuousRewardProvider.reward for the pose as it stands if nothing collides
        const double x = hand_pose[lane], y = hand_pose[kBlock + lane], th = hand_pose[2 * kBlock + lane];
        const int64_t g = slot_of(a.S, i, q);
        PathWindow win;
        if (a.S->path.shared)
            win = path_window(P, (LdsF64)lds_box, (const __attribute__((address_space(3))) int16_t*)lds_index, x, y);
        else
            win = path_window_compact(P, box, reinterpret_cast<const uint16_t*>(a.hot.path_bbox + g * kBoxDoubles) + kBoxIndexU16,
                                      reinterpret_cast<const int32_t*>(a.hot.path_bbox + g * kBoxDoubles)[kBoxLenWord + 1], x, y);
        const int m = a.S->path.shared ? a.S->path.max_len : a.S->path.lens[g];
        double min_dist = q.min_dist;
        int target = q.target;
        double rew;
        const double* gpath = a.S->path.pts + (a.S->path.shared ? 0 : g * (int64_t)a.S->path.max_len * 5);
        if (!PLAIN && P.reward_provider == BCP_REWARD_PURE_PURSUIT) {   // (no collision this step: the sticky flag as it is)
            if (lds_path) rew = reward_pure_pursuit(lds_path, m, x, y, q.collided != 0, min_dist, target);
            else rew = reward_pure_pursuit(gpath, m, x, y, q.collided != 0, min_dist, target);
        } else if (lds_path) {
            rew = reward_step(P, lds_path, win, m, x, y, th, min_dist, target);
        } else {
            rew = reward_step(P, gpath, win, m, x, y, th, min_dist, target);
        }
        hand_score[lane] = rew;
        hand_score[kBlock + lane] = min_dist;
        hand_score[2 * kBlock + lane] = (double)target;
    }
    __syncthreads();
    if (mover && active) {
        ScoredFree sc;
        sc.rew = hand_score[lane];
        sc.min_dist = hand_score[kBlock + lane];
        sc.target = (int)hand_score[2 * kBlock + lane];
        finalize_env<PLAIN>(a, i, q, hit, lds_path, nullptr, !ABLATED(a, kAblateNoReward), sc);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) a.tick[1] = a.step_counter;   // for kernel 2 (see StepArgs::tick)
    advance_step_by_ticket(a);   // (only when no kernel 2 follows)
}

// Kernel 2 of a step: kPendingWaves wavefronts per parked env: the lanes rasterise
// the footprint together (coop_collides, wave w takes the row chunks w, w+2, ...); on a collision thread 0 redoes
// the env's finalisation from the parked state.  The first entry is fetched speculatively, together with the
// counter that says whether it exists, so the two round trips overlap.
constexpr int kPendingWaves = 4;  // wave = 2 * (row-chunk slot) + (edge slot)
// Undecided poses per step up to which kernel 1 parks them all.  With a pose per wave (below) kernel 2 takes 8192 poses
// per round and spreads them evenly, which beat settling them inside kernel 1's waves in every workload measured (the
// aisle config, tens of thousands of undecided poses per step: 0.076 ms parked, 0.086 ms in place, 0.102 ms with the
// former limit of 8192); the in-place path remains for far denser cases and as BCP_TUNE_DENSE_THRESHOLD.
constexpr int kParkCapacity = 1 << 20;

template <bool WIDE, bool PLAIN>
__global__ void __launch_bounds__(kBlock * kPendingWaves) step_pending_kernel(const StepArgs launch_args)
{
    const StepArgs a = resolve_step(launch_args, 1);
    DIAG_STAMP(0);
    const DevParams& P = a.S->P;
    const int lane = threadIdx.x % kBlock, wave = threadIdx.x / kBlock;
    const int shard = (int)(blockIdx.x % kShards);
    const int stride = gridDim.x / kShards;
    const Pending* slots = a.hot.pending + shard;
    const double vqx = lane < P.n_verts ? P.qverts[lane][0] : 0.0, vqy = lane < P.n_verts ? P.qverts[lane][1] : 0.0;
    const int count = a.pending_count[shard];
    if (blockIdx.x == 0 && a.threshold_next && threadIdx.x < kShards) {
        // Undecided poses of this step, parked + settled in place.  Up to kParkCapacity the next step parks everything
        // (no wave is held up by its own unlucky lanes); beyond that every wave settles its own.
        int total = a.pending_count[threadIdx.x] + a.inplace_count[threadIdx.x];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o);
        if (threadIdx.x == 0) *a.threshold_next = total <= kParkCapacity ? 64 : 0;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) a.tick[0] = a.step_counter + 1;   // the next step's kernel 1 reads this
    if (count > kWavePerPoseFrom * stride) {
        // Many parked poses in this shard (private worlds with the robots near their walls: several rounds for every
        // team): throughput matters more than the latency of one pose, so every WAVE takes poses of its own -- four
        // times as many in flight, each about twice as long in the single-wave rasteriser.
        for (int idx = (blockIdx.x / kShards) * kPendingWaves + wave; idx < count; idx += stride * kPendingWaves) {
            const Pending* e = slots + (int64_t)idx * kShards;
            const int64_t i = ((int64_t)e->env_hi << 32) | (uint32_t)e->env_lo;
            const int64_t g = a.hot.geom_of_env ? (int64_t)e->geom : i;
            const uint32_t* words = a.hot.map_bits + (a.hot.map_shared ? 0 : g * a.hot.map_env_stride);
            bool hit = false;
            if (!ABLATED(a, kAblateNoCoop))
                hit = coop_collides<WIDE>(P, vqx, vqy, e->c, e->s, e->px, e->py, words, a.hot.map_rows, a.hot.map_cols,
                                          a.hot.map_wpr);
            if (hit && lane == 0) {
                Pending q = *e;
                finalize_env<PLAIN>(a, i, q, true);
            }
        }
        return;
    }
    for (int idx = blockIdx.x / kShards; idx < a.hot.pending_cap; idx += stride) {
        const Pending* e = slots + (int64_t)idx * kShards;   // in bounds whatever `count` says
        const double c = e->c, s = e->s;
        const int px = e->px, py = e->py;
        const int64_t i = ((int64_t)e->env_hi << 32) | (uint32_t)e->env_lo;
        const int64_t g = a.hot.geom_of_env ? (int64_t)e->geom : i;
        if (idx >= count) break;
        DIAG_STAMP(1);
        const uint32_t* words = a.hot.map_bits + (a.hot.map_shared ? 0 : g * a.hot.map_env_stride);
        // (no inner distance-field test here: nearly every parked pose is free, so the test would cost a dependent
        //  round trip per pose and almost never spare the rasteriser)
        bool hit = false;
        DIAG_STAMP(2);
        if (!ABLATED(a, kAblateNoCoop))
            hit = coop_collides_quad<WIDE>(P, vqx, vqy, c, s, px, py, words, a.hot.map_rows, a.hot.map_cols, a.hot.map_wpr, wave,
                                           (LdsU32)lds_dyn);
        DIAG_STAMP(3);
        hit = __syncthreads_or(hit);  // wave-uniform verdicts of the block's waves
        DIAG_STAMP(4);
        // kernel 1 already finished this env as "free"; only a collision changes anything
        if (hit && threadIdx.x == 0) {
            Pending q = *e;
            finalize_env<PLAIN>(a, i, q, true);
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// The step as ONE launch: step_local_kernel.  The hand-off of undecided poses stays inside the workgroup.
//
// Round 1 settled them in a second launch (step_pending_kernel: 13 of the step's 30 us on the metric workload -- kernel
// boundary, cold caches, ramp-up -- for ~1000 poses).  A global hand-off inside one launch (queues in HBM, write-through
// stores, sc1 polls, no fences) was built and measured in round 2: 13 - 25 x SLOWER, every parked pose costs device-scope
// atomics and polls on a few hot lines that the memory side serves one after the other (profiles/r02_queue_handoff_attempt.txt).
// So nothing leaves the CU here:
//   * a workgroup = 256 envs = 4 (mover, scorer) wave pairs as in step_fast_pair_kernel + 8 helper waves: 16 waves, one
//     workgroup per CU, four waves per SIMD;
//   * the waves that idle until the new pose exists do what needs no pose: the scorers draw the step's odometry noise, a
//     helper computes cos / sin of the old heading (barrier 0 hands both to the movers);
//   * movers park the POSES of their undecided envs in LDS (40 bytes each; the env's state stays in the mover lane's
//     registers); "barrier 2" is a pair of LDS counters: a mover waits for the three scan results of its pair, everybody
//     for all four movers' parked poses;
//   * then ALL 16 waves draw tickets (an LDS counter) and settle one parked pose each -- the exact test, cell by cell
//     (coop_collides_sparse) -- and post the verdict in the record, while the movers first finish their decided envs;
//   * a mover lane that parked an env waits for its verdict (a few hundred cycles, LDS) and finishes the env itself:
//     rollback on a hit, reward, done, in-kernel reset, stores -- in SIMD with the wave's other parked lanes.
//     With ~4 parked poses per 256 envs and 16 waves the exact tests of a workgroup run side by side, right after the
//     classification, on a warm CU.
// No queue, no atomic in global memory, no poll, no second launch; load balance comes from the workgroup being large
// (the sum of 256 envs' luck) and from the helper waves.
// The workgroup comes in three sizes (template parameter PAIRS, BCP_TUNE_LOCAL_PAIRS): 4 pairs = 16 waves = 256 envs (one
// workgroup per CU: rounds 2-3), 2 pairs = 8 waves = 128 envs (two co-resident per CU) and 1 pair = 4 waves = 64 envs (four).
// Every form has the same four waves per 64 envs and the same four waves per SIMD when the chip is full; what changes is the
// granule in which the CU's registers are handed out: a 16-wave workgroup shares a CU with NOTHING (112 VGPRs x 1024 threads),
// so one workgroup more than the chip holds, a sampler or an RCCL kernel resident on some CUs, costs a whole further round
// (profiles/r03_n_sweep.txt: 65 536 envs 11.8 us, 65 792 envs 17.8 us); an 8-wave one can start beside a leaving neighbour.
constexpr int kLocalPairsMax = 4;
constexpr int kLocalPairsDefault = 4;   // (bcplan.hip: local_pairs)

typedef const __attribute__((address_space(4))) StepArgs& KernArgs;   // the launch arguments where they lie: scalar loads on
                                                                      // demand instead of ~500 bytes pinned in SGPRs

constexpr int kScanItems = 1024;       // candidates of a pair's 64 envs that the flat scan list holds (bytes of LDS)
constexpr int kHandDoubles = 8;   // per pair and env: pose [3], score / noise [3], cos / sin of the old heading [2]
constexpr int kLocalMapWords = 4096;   // a shared lethal bitmap of up to 16 KB is staged in LDS for the exact tests

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int kStaticChunks = (int)((sizeof(StepStatic) + 15) / 16);   // *S in 16-byte pieces

static size_t local_step_lds_bytes(int n_verts, int lds_path_doubles, int staged_map_words, bool plain = true, int pairs = kLocalPairsMax)
{
    const int kLocalPairs = pairs, kLocalWaves = 4 * pairs, kLocalEnvs = pairs * kBlock;
    size_t bytes = ((size_t)2 * n_verts + lds_path_doubles + (size_t)kLocalPairs * kHandDoubles * kBlock + 8) * sizeof(double);
    bytes += 2 * kBlock * sizeof(uint32_t);          // bucket index of the shared path
    bytes += 16 * sizeof(int32_t);                   // parked count, ticket counter, movers parked, -, scans done per pair [4], scan lists ready per pair [4], - [4]
    bytes = (bytes + 15) & ~(size_t)15;
    bytes += (size_t)kLocalEnvs * sizeof(ParkedPose);
    bytes += (size_t)kLocalWaves * kSparseLdsWords * sizeof(uint32_t);   // a cell list per wave (coop_collides_sparse)
    bytes += (size_t)staged_map_words * sizeof(uint32_t);
    bytes = (bytes + 15) & ~(size_t)15;
    bytes += kStaticChunks * 16;   // the parameter block *S
    if (!plain) bytes += (size_t)10 * kLocalEnvs * sizeof(double);   // delay queues: what the step's pushes will hand back
    return bytes;
}

// A launch-argument value fetched NOW: the empty asm makes every 32-bit word of `v` an opaque scalar register at this
// point, so the compiler can neither sink the fetch into a later basic block nor fetch the field again.  At a kernel start
// every first touch of an argument line is a scalar-cache miss of ~450 cycles (measured: tools/diag_local.py), and hipcc
// fetches an argument where it is first used, block by block, each fetch with its own wait -- nine dependent round trips
// = 4.3 k cycles before the mover's last load was issued.  Fetched together (fetch_words: plain loads of adjacent words,
// merged into wide fetches) and then pinned (pin_words), the prologue's arguments arrive in ONE round trip.
template <int NW>
__device__ __forceinline__ void fetch_words(const __attribute__((address_space(4))) void* p, uint32_t (&w)[NW])
{
    const __attribute__((address_space(4))) uint32_t* src = (const __attribute__((address_space(4))) uint32_t*)p;
#pragma unroll
    for (int k = 0; k < NW; ++k) w[k] = src[k];
}

template <int NW>
__device__ __forceinline__ void pin_words(uint32_t (&w)[NW])
{
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        asm("" : "+s"(w[k]));   // (not volatile: a volatile asm counts as a store, and every later fetch of a uniform
                                //  address -- the parameter block *S -- would turn into a vector load)
        w[k] = (uint32_t)__builtin_amdgcn_readfirstlane((int)w[k]);   // (tells the compiler the word is wave-uniform; folds away)
    }
}

// what the prologue of step_local_kernel needs from the launch arguments (same member names as StepArgs: load_env takes either)
struct PrologueArgs {
    StepHot hot;
    const StepStatic* S;
    const void* actions;
    const double* noise_z;
    uint64_t* tick;
    uint32_t flags;
};

// a wave-uniform 64-bit value that arrived through a vector load
__device__ __forceinline__ uint64_t uniform_u64(uint64_t v)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

static_assert(kStaticChunks <= 128, "*S is staged by the last two waves of a workgroup");
// A spin on an LDS word another wave of the workgroup will set (`cond` true = keep waiting, wave-uniform).  Bounded: a wait
// that the hand-off protocol guarantees to end within ~20 k cycles gives up after kPollLimit trips (~10^7 cycles) and sets
// `expired`; the wave carries on with what it has and reports BCP_ERR_INTERNAL at the end.  (hipcc 7.2 has turned two
// shapes of exactly these loops into endless spins, DESIGN.md "Compiler notes": a miscompile must fail a test, not wedge the GPU.)
constexpr int kPollLimit = 1 << 16;
#define BOUNDED_POLL(cond, expired) do { int trips_ = 0; while (cond) { __builtin_amdgcn_s_sleep(1); if (++trips_ > kPollLimit) { (expired) = true; break; } } } while (0)
constexpr uint32_t kDiagWithholdVerdicts = 1u << 23;   // -DBCP_DIAG builds: parked poses are tested but their verdicts never posted

// ROLL (bcp_rollout): the workgroup takes its envs through StepArgs::rollout_steps steps inside ONE launch -- actions,
// optional normals and the outputs are [steps][N] arrays, the state goes through its usual arrays between steps.  The launch
// and the staging of footprint / path / lethal map / parameter block happen once, and -- the larger part -- a workgroup
// starts its next step when IT is done, not when the slowest workgroup of the chip is (envs are independent,
// envs/base/env.py:334-361: there is nothing to wait for).  Every trip fetches its launch arguments again (scalar-cache hits
// from the second on): carried across the loop the ~60 pinned words did not fit the scalar registers (600 spills).  Bit for
// bit the same states and outputs as that many launches of the one-step form (tests/test_gpu_rollout.py).  The one-step
// form is this code with the loop folded away.
typedef const __attribute__((address_space(4))) StepArgs* KernArgPtr;

// one step of the workgroup's envs: the whole of the one-step kernel, and one trip (`rs`) of a rollout
template <bool WIDE, bool PLAIN, int PAIRS, bool ROLL>
__device__ __forceinline__ void step_local_body(KernArgPtr kernarg, const int rs)
{
    static_assert(PAIRS == 1 || PAIRS == 2 || PAIRS == 4, "a workgroup holds 1, 2 or 4 (mover, scorer, helper, helper) quartets");
    constexpr int kLocalPairs = PAIRS, kLocalWaves = 4 * PAIRS, kLocalEnvs = PAIRS * kBlock;
    constexpr int kPairShift = PAIRS == 4 ? 2 : (PAIRS == 2 ? 1 : 0);
    constexpr int kStaticFrom = (kLocalWaves - 2) * kBlock;   // *S is staged by the last two waves
    constexpr int kCtlFrom = (kLocalWaves - 1) * kBlock;      // the control words are zeroed by the last wave
    [[maybe_unused]] constexpr int kWScorer = PAIRS, kWHelper1 = 2 * PAIRS, kWHelper2 = 3 * PAIRS;   // pair 0's waves (stamps)
    DIAG_STAMP_WAVES(512);    // every wave: first instruction
    DIAG_REAL_ENTRY();
    // (static issue priorities by role, s_setprio -- the movers above everybody else, or everybody else above the movers
    //  until barrier 0, or behind barrier 1 -- change nothing: +-0.5 % in every form tried)
    // The prologue is ONE memory round trip.  Everything a wave asks for first -- the mover's state, action and robot
    // constants, a scanner's target index, the old heading of the helper that takes its cos / sin, every wave's share of the
    // staging data (footprint, shared path + index, lethal bitmap, the parameter block *S) -- is addressed from the launch
    // arguments alone (StepHot, fetched together: fetch_words / pin_words) and issued back to back before anything is waited
    // for; what was loaded is stored to LDS only after the wave has done what it can do without it.  Nothing before barrier 0
    // reads *S through the scalar cache: those fetches take 1.3 - 2.2 k cycles at a kernel start, and a wave that waits for
    // one waits for all of them.  (Round 2 staged array by array -- load, wait, LDS store -- fetched each launch argument
    // where it was first used (~450 cycles per first touch, nine in a row) and only then issued the state loads: 4.1 k
    // cycles until a mover's state had landed, 5.1 k to barrier 0; tools/diag_local.py.)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool mover = wave < kLocalPairs, scorer = wave >= kLocalPairs && wave < 2 * kLocalPairs;
    const int pair = wave & (kLocalPairs - 1);
    KernArgs a = *kernarg;
    // (the launch arguments the prologue uses, fetched together before the first branch: pin_sgpr)
    PrologueArgs L;
    {
        constexpr int kStateWords = PLAIN ? kHotStateWords : (int)(sizeof(DevState) / 4);
        uint32_t w_st[kStateWords], w_hot[kHotPrologueWords];
        fetch_words(&a.hot.st, w_st);
        fetch_words(&a.hot.n, w_hot);
        pin_words(w_st);
        pin_words(w_hot);
        __builtin_memset(&L.hot, 0, sizeof(L.hot));
        __builtin_memcpy(&L.hot.st, w_st, sizeof(w_st));
        __builtin_memcpy(&L.hot.n, w_hot, sizeof(w_hot));
        L.actions = L.hot.io_actions;
        L.noise_z = L.hot.io_noise_z;
        L.tick = L.hot.io_tick;
        L.flags = L.hot.io_flags;
        L.S = a.S;   // (not pinned: a pointer that went through the asm is no longer known to be uniform, nor global)
        if (ROLL) {   // this step's rows of the [steps][N].. inputs
            const int64_t row = (int64_t)rs * L.hot.n;
            L.actions = (const char*)L.actions + row * ((L.flags & BCP_STEP_ACTIONS_F32) ? 8 : 16);
            L.noise_z = L.noise_z ? L.noise_z + 3 * row : nullptr;
        }
    }
    const int64_t out_base = ROLL ? (int64_t)rs * L.hot.n : 0;
    const int hot_n_verts = L.hot.n_verts, hot_npath = L.hot.lds_path_doubles, hot_path_shared = L.hot.path_shared;
    const int hot_map_rows = L.hot.map_rows, hot_map_cols = L.hot.map_cols, hot_map_wpr = L.hot.map_wpr;
    const int hot_map_shared = L.hot.map_shared, hot_noise_on = L.hot.noise_on, hot_max_len = L.hot.path_max_len;
    const int64_t hot_n = L.hot.n;
    // PLAIN (continuous reward, no delays): the backward scan for the last reached way point -- the longest stretch of
    // the reward provider -- is split three ways between the pair's scorer and its two helper waves (every third
    // candidate of the window each; contiguous thirds for paths in global memory); the mover takes the maximum and does
    // the rest of the provider itself.
    const bool scanner = PLAIN ? !mover : scorer;
    const int member = (wave >> kPairShift) - 1;   // 0 = scorer, 1 / 2 = helpers (PLAIN only)

    // ---- LDS
    __attribute__((address_space(3))) double* qv = (__attribute__((address_space(3))) double*)lds_dyn;
    const int nq = 2 * hot_n_verts;
    const int npath = hot_npath;
    const LdsF64 lds_path = npath ? (LdsF64)(qv + nq) : (LdsF64) nullptr;
    __attribute__((address_space(3))) double* hand_pose = qv + nq + npath + pair * kHandDoubles * kBlock;
    __attribute__((address_space(3))) double* hand_score = hand_pose + 3 * kBlock;
    __attribute__((address_space(3))) double* hand_heading = hand_pose + 6 * kBlock;
    __attribute__((address_space(3))) double* lds_box = qv + nq + npath + kLocalPairs * kHandDoubles * kBlock;   // [8]
    __attribute__((address_space(3))) uint32_t* lds_index = (__attribute__((address_space(3))) uint32_t*)(lds_box + 8);  // [128]
    __attribute__((address_space(3))) int32_t* ctl = (__attribute__((address_space(3))) int32_t*)(lds_index + 2 * kBlock);   // [16]
    const uint32_t rec_off = (uint32_t)((((size_t)(nq + npath + kLocalPairs * kHandDoubles * kBlock + 8) * sizeof(double) +
                                          2 * kBlock * sizeof(uint32_t) + 16 * sizeof(int32_t)) + 15) & ~(size_t)15);
    __attribute__((address_space(3))) ParkedPose* rec =
        (__attribute__((address_space(3))) ParkedPose*)((__attribute__((address_space(3))) char*)lds_dyn + rec_off);
    const LdsU32 cell_list = (LdsU32)(rec + kLocalEnvs) + wave * kSparseLdsWords;
    const LdsU32 lds_map = (LdsU32)(rec + kLocalEnvs) + kLocalWaves * kSparseLdsWords;
    const int map_words = ((hot_map_shared != 0) & (hot_map_rows * hot_map_wpr <= kLocalMapWords)) ? hot_map_rows * hot_map_wpr : 0;
    // the parameter block *S, copied into LDS by the prologue: behind barrier 0 every wave reads its parameters from there.
    // (Scalar fetches of *S take 1.3 - 2.2 k cycles at a kernel start, and a wave that holds the ~60 words it needs of the
    //  block in scalar registers from the start has none left for the launch arguments.)
    const uint32_t static_off = (uint32_t)(((size_t)((__attribute__((address_space(3))) char*)(lds_map + map_words) -
                                                     (__attribute__((address_space(3))) char*)lds_dyn) + 15) & ~(size_t)15);
    __attribute__((address_space(3))) u32x4* lds_static = (__attribute__((address_space(3))) u32x4*)((__attribute__((address_space(3))) char*)lds_dyn + static_off);
    const __attribute__((address_space(3))) StepStatic* SL = (const __attribute__((address_space(3))) StepStatic*)lds_static;
    // delay queues (PLAIN = false): the ten values this step's pushes into the pose / robot-state queues will hand back are
    // fetched with the state (fifo_peek) and needed when the env is finished; in between they wait in LDS, not in 20 vector
    // registers of a wave that has none to spare
    __attribute__((address_space(3))) double* fifo_stash = (__attribute__((address_space(3))) double*)(lds_static + kStaticChunks) + pair * kBlock + lane;

    DIAG_STAMP(0);
    DIAG_STAMP_WAVES(768);    // every wave: launch arguments fetched
    const int64_t gi = (int64_t)blockIdx.x * kLocalEnvs + pair * kBlock + lane;
    const bool active = gi < hot_n;
    const int64_t i = active ? gi : hot_n - 1;   // (inactive lanes of the last workgroup shadow env n-1 and never store)
    const bool noise_by_waves = (hot_noise_on != 0) & (L.noise_z == nullptr);   // the step's odometry noise is drawn here, by idle waves

    // (0) the step counter and the noise seed live on the device (StepArgs::tick): two dependent scalar fetches, first in
    //     line for the waves that draw the noise (the last wave also moves the counter on at the end); movers never read them
    //     (vector loads of a uniform address: a scalar fetch would share its counter with the fetches of *S below, and a
    //      wave waiting for the one waits for all of them)
    uint64_t tick_counter = 0, tick_seed = 0;
    if (!mover) {
        int zero;
        asm("v_mov_b32 %0, 0" : "=v"(zero));   // (opaque: keeps these loads on the vector side)
        const GlobalPtr<const uint64_t> t = as_global((const uint64_t*)L.tick) + zero;
        tick_counter = t[0];
        tick_seed = t[2];
    }
    // (1) every wave's own loads: the mover's state / action, a scanner's reward-state words, the old heading for the
    //     helper that takes its cos / sin
    Pending q;
    double cmd0 = 0.0, cmd1 = 0.0;
    float box[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // (a private path's box and bucket grid: the f32 words of its record)
    float quant_inv = 0.0f;                    // (... and the reciprocal step of its prefilter records)
    double old_angle = 0.0;
    double own_org_x = 0.0, own_org_y = 0.0;                  // (only ever written by the loads below: a later assignment
    uint64_t own_len_shift = 0;                                // (private path: way points | index shift << 32)
                                                               //  to a register a load is in flight to would wait for it)
    const bool own_origin = !hot_path_shared || L.hot.map_origins != nullptr;
    if (mover) {
        load_env<PLAIN>(L, 0, 0, i, active, q, cmd0, cmd1, false);
        if (!hot_path_shared) {   // private paths: origin and length of the entry share a line
            const int64_t g = L.hot.geom_of_env ? (int64_t)q.geom : i;
            own_org_x = as_global(L.hot.path_bbox)[g * kBoxDoubles + kBoxOrigin];
            own_org_y = as_global(L.hot.path_bbox)[g * kBoxDoubles + kBoxOrigin + 1];
            own_len_shift = as_global(reinterpret_cast<const uint64_t*>(L.hot.path_bbox))[g * kBoxDoubles + kBoxLenWord / 2];
        } else if (L.hot.map_origins) {   // (private maps with a shared path)
            const int64_t g = L.hot.geom_of_env ? (int64_t)q.geom : i;
            own_org_x = as_global(L.hot.map_origins)[2 * g + 0];
            own_org_y = as_global(L.hot.map_origins)[2 * g + 1];
        }
    } else {
        if (scanner) {
            if (!PLAIN) q.min_dist = as_global(L.hot.st.min_dist)[i];   // (PLAIN: the scan only needs the target index)
            q.target = as_global(L.hot.st.target_idx)[i];
            q.geom = L.hot.geom_of_env ? as_global(L.hot.geom_of_env)[i] : 0;
            q.collided = PLAIN ? 0 : (int32_t)(as_global(L.hot.st.collided)[i] != 0);
            if (!hot_path_shared) {
                const int64_t g = L.hot.geom_of_env ? (int64_t)q.geom : i;
#pragma unroll
                for (int k = 0; k < 8; ++k) box[k] = as_global(reinterpret_cast<const float*>(L.hot.path_bbox + g * kBoxDoubles))[k];
                quant_inv = as_global(reinterpret_cast<const float*>(L.hot.path_bbox + g * kBoxDoubles))[kBoxQuantStep + 1];
                own_len_shift = as_global(reinterpret_cast<const uint64_t*>(L.hot.path_bbox))[g * kBoxDoubles + kBoxLenWord / 2];
            }
        }
        if (member == 1) old_angle = as_global(L.hot.st.angle)[i];
    }
    //     ... and the mover the seven robot constants of the model's first half (dt .. p_gain, adjacent in DevParams), by
    //     vector loads of a uniform address: they arrive with the state, in vector registers, and scalar registers stay free
    //     (PLAIN only: with delay queues the first half runs behind barrier 0, on the constants in LDS)
    double rc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // (+ 1 / dt, 1 / L: div_by_const)
    if (PLAIN && mover) {   // (a rollout fetches them in every trip: seven loads, nothing to carry across the loop)
        int zero;
        asm("v_mov_b32 %0, 0" : "=v"(zero));   // (opaque: keeps these loads on the vector side)
        const GlobalPtr<const double> pc = as_global(&a.S->P.dt) + zero;
#pragma unroll
        for (int u = 0; u < 9; ++u) rc[u] = pc[u];
    }
    // (2) the parameter block *S (16 bytes per thread of the last two waves) is the only staging data anybody needs before
    //     barrier 1: the mover's second half reads its parameters from the LDS copy.  Footprint vertices, the shared path with
    //     its bounding box and bucket index and the shared lethal bitmap are first read behind barrier 1, so the waves that
    //     idle between the barriers fetch and store them there (below) -- a prologue that also addressed and issued ~8 staging
    //     loads per thread kept the SIMDs busy for 1 - 2 k cycles before the movers' own loads were even issued.
    //     (issued FIRST by those two waves it was measured slower, 11.75 against 11.70 us: it then competes with the movers' loads)
    u32x4 st_static = {0u, 0u, 0u, 0u};
    const bool first_trip = !ROLL || rs == 0;   // the staging data is copied into LDS once per launch
    if (first_trip && tid >= kStaticFrom && tid < kStaticFrom + kStaticChunks) st_static = as_global(reinterpret_cast<const u32x4*>(a.S))[tid - kStaticFrom];
    DIAG_STAMP_U(0, 10);   // mover: own + staging loads issued
    DIAG_STAMP_U(kWScorer, 11);   // scorer: the same
    DIAG_STAMP_WAVES(1280);   // every wave: prologue loads issued
    __builtin_amdgcn_sched_barrier(0);
    // (a rollout: the counter in memory stands still until the launch's last trip has drawn its ticket)
    const uint64_t step_counter = mover ? 0 : uniform_u64(tick_counter) + (ROLL ? (uint64_t)rs : 0ull), seed = mover ? 0 : uniform_u64(tick_seed);
    // (5) what needs no pose.  The odometry noise of this step needs nothing but seed, env id and step counter: Philox +
    //     float64 Box-Muller.  Slots 1 and 2 -- the ones PlanEnv's noise model draws -- are the two halves of one pair
    //     (device_normals): the scorer draws them; slot 0, when the model can consume it at all, the pair's second helper;
    //     (Splitting the pair -- radius on the scorer, direction on a helper, both running the Philox block -- was measured
    //      in round 3: the second copy of the generator costs the SIMD more issue slots than the shorter chain wins,
    //      11.87 - 12.0 against 11.72 us per step.)
    //     (So was drawing the NEXT step's pair at the end of a step, when the scorers idle, and fetching it here with a key
    //      {seed, step, env base}: the noise is then ready 0.9 k cycles earlier, but the ten extra loads in the scorers'
    //      prologue hold up everybody's own loads -- the movers reached barrier 0 0.4 k cycles LATER; 11.7 against 11.56 us.
    //      And fetching the scanners' target index / path box and the helper's old heading behind barrier 0 instead of
    //      here: 11.63 against 11.51 us.)
    if (noise_by_waves && scorer) {
        double z1, z2;
        device_normals_12(seed, (uint64_t)(L.hot.env_id_base + i), step_counter, z1, z2);
        hand_score[kBlock + lane] = z1;
        hand_score[2 * kBlock + lane] = z2;
        DIAG_STAMP_U(kWScorer, 1);   // scorer: noise drawn
    }
    if (noise_by_waves && member == 2) hand_score[lane] = L.hot.noise_slot0 ? device_normal_0(seed, (uint64_t)(L.hot.env_id_base + i), step_counter) : 0.0;
    //     (the first helper's cos / sin of the OLD heading is needed behind barrier 1 only: it follows barrier 0)
    //     the mover the first half of the robot model (_env_step, envs/base/env.py:442-461): front-wheel column, cos / sin
    //     of the new wheel angle, velocity model
    Robot& r = q.r;
    RobotDrive drive;
    drive.v = drive.w = 0.0;
    drive.noisy = false;
    if (mover) {
        q.old = r.p;
        q.drawn = 0;
#ifdef BCP_DIAG
        DIAG_WAIT_VMEM();
        DIAG_STAMP_U(0, 3);    // mover: state loads landed (diagnostic build only: the wait is not in the shipping kernel)
#endif
      if (PLAIN) {
        RobotConsts robot;
        robot.model = L.hot.model;
        robot.dynamic_model = L.hot.dynamic_model;
        robot.model_front_column_pid = L.hot.model_front_column_pid;
        robot.noise_on = hot_noise_on;
        robot.dt = rc[0];
        robot.L = rc[1];
        robot.max_wheel_angle = rc[2];
        robot.max_wheel_speed = rc[3];
        robot.max_lin_acc = rc[4];
        robot.max_ang_acc = rc[5];
        robot.p_gain = rc[6];
        robot.inv_dt = rc[7];
        robot.inv_L = rc[8];
        drive = robot_step_begin(robot, r, cmd0, cmd1);
        DIAG_STAMP_U(0, 12);   // mover: first half of the robot model done
      }
        if (!PLAIN) {
#pragma unroll
            for (int k = 0; k < 3; ++k) fifo_stash[k * kLocalEnvs] = q.popped_pose[k];
#pragma unroll
            for (int k = 0; k < 7; ++k) fifo_stash[(3 + k) * kLocalEnvs] = q.popped_state[k];
        }
    }
    // (6) the parameter block into LDS
    if (tid >= kCtlFrom && tid < kCtlFrom + 16) ctl[tid - kCtlFrom] = 0;
    if (first_trip && tid >= kStaticFrom && tid < kStaticFrom + kStaticChunks) lds_static[tid - kStaticFrom] = st_static;
    DIAG_STAMP_WAVES(1024);
    __syncthreads();   // barrier 0: noise, old heading and the parameter block are in LDS
    const DevParams& P = *(const DevParams*)&SL->P;
    const int map_rows = hot_map_rows, map_cols = hot_map_cols;
    const int my_len = hot_path_shared ? hot_max_len : (int)(uint32_t)own_len_shift;   // way points of this env's path
    // (7) mover: the second half of the robot model; the pose the reward provider will see goes to the scanning waves
    Pose new_pose;
    new_pose.x = new_pose.y = new_pose.th = 0.0;
    if (mover) {
        if (noise_by_waves) {
            q.z[0] = hand_score[lane];
            q.z[1] = hand_score[kBlock + lane];
            q.z[2] = hand_score[2 * kBlock + lane];
        }
        DIAG_STAMP(1);   // (the compiler may move loads across this: indicative only)
        if (!PLAIN) drive = robot_step_begin(P, r, cmd0, cmd1);
        // the new pose goes out as soon as it exists; the measured velocities (path_velocity: a square root, two divisions)
        // are the mover's own business, behind barrier 1
        new_pose = robot_step_pose(P, r, drive, q.z, q.drawn);
        const bool delayed = !PLAIN && P.pose_delay > 0 && q.iter + 1 > 1;
        hand_pose[lane] = delayed ? fifo_stash[0] : new_pose.x;
        hand_pose[kBlock + lane] = delayed ? fifo_stash[kLocalEnvs] : new_pose.y;
        hand_pose[2 * kBlock + lane] = delayed ? fifo_stash[2 * kLocalEnvs] : new_pose.th;
        DIAG_STAMP(2);
    }
    // (7a) the first helper: cos / sin of the OLD heading, which the robot model needs at its very end (path_velocity, behind
    //      barrier 1) -- before barrier 0 it made its wave the last one to arrive there
    if (!mover && member == 1) {
        double c0, s0;
        cos_sin(old_angle, c0, s0);
        hand_heading[lane] = c0;
        hand_heading[kBlock + lane] = s0;
        DIAG_STAMP_U(kWHelper1, 2);   // helper: cos / sin of the old heading
    }
    // (7b) everybody else stages what is read behind barrier 1: loads first, all of them in flight together, then the stores
    if (!mover && first_trip) {
        constexpr int kStagers = (kLocalWaves - kLocalPairs) * kBlock;            // 768 / 384 / 192 threads
        constexpr int kMapPerStager = (kLocalMapWords + kStagers - 1) / kStagers;   // 6 / 11 / 22 words each
        constexpr int kBoxAt = kStagers >= 448 ? 256 : kStagers - 8;              // who fetches the path's box (8 doubles) ...
        constexpr int kIndexAt = kStagers >= 448 ? 320 : kStagers - 136;          // ... and its bucket index (128 words)
        const int nm = tid - kLocalPairs * kBlock;
        double st_q = 0.0, st_path = 0.0;
        float st_box = 0.0f;
        uint32_t st_index = 0;
        uint32_t st_map[kMapPerStager] = {};
        if (nm < nq) st_q = as_global(L.hot.qverts)[nm];
        if (nm < npath) st_path = as_global(L.hot.path_pts)[nm];
        if (hot_path_shared) {
            if (nm >= kBoxAt && nm < kBoxAt + 8) st_box = as_global(reinterpret_cast<const float*>(L.hot.path_bbox))[nm - kBoxAt];
            if (nm >= kIndexAt && nm < kIndexAt + 128) st_index = as_global(reinterpret_cast<const uint32_t*>(L.hot.path_index))[nm - kIndexAt];
        }
#pragma unroll
        for (int u = 0; u < kMapPerStager; ++u) {
            const int k = u * kStagers + nm;
            if (k < map_words) st_map[u] = as_global(L.hot.map_bits)[k];
        }
        if (nm < nq) qv[nm] = st_q;
        if (nm < npath) qv[nq + nm] = st_path;
        for (int k = kStagers + nm; k < npath; k += kStagers) qv[nq + k] = as_global(L.hot.path_pts)[k];   // (long paths)
        if (hot_path_shared) {
            if (nm >= kBoxAt && nm < kBoxAt + 8) lds_box[nm - kBoxAt] = (double)st_box;
            if (nm >= kIndexAt && nm < kIndexAt + 128) lds_index[nm - kIndexAt] = st_index;
        }
#pragma unroll
        for (int u = 0; u < kMapPerStager; ++u) {
            const int k = u * kStagers + nm;
            if (k < map_words) lds_map[k] = st_map[u];
        }
    }
    DIAG_STAMP_WAVES(1536);
    __syncthreads();
    DIAG_STAMP(3);
    if (mover) {
        KnownHeading old_heading;   // (from the first helper wave, barrier 1)
        old_heading.c0 = hand_heading[lane];
        old_heading.s0 = hand_heading[kBlock + lane];
        old_heading.known = true;
        q.err = robot_step_measure(P, r, new_pose, old_heading);
    }
    bool park = false;
    bool poll_expired = false;   // (wave-uniform) one of the bounded waits below gave up
    if (mover) {
        // (3a) collision: distance-field classification; an undecided env is parked below
        // (its parameters are read from the LDS copy of *S here, where they are used: held from barrier 0 on they cost
        //  25 vector registers across the robot model)
        OuterParams outer = outer_params(*(const CullDesc*)&SL->cull);
        outer.near_tx = SL->cull.step_near_tx;
        outer.near_shift = SL->cull.step_near_shift;
        const double map_inv_res = SL->map.inv_res;
        const int64_t near_stride = SL->cull.step_near_stride;
        const double org_x = own_origin ? own_org_x : SL->map.ox, org_y = own_origin ? own_org_y : SL->map.oy;
        const int64_t g = slot_of(SL, i, q);
        const int px = (int)rint((r.p.x - org_x) * map_inv_res);  // world_to_pixel, coordinate_transformations.py:185-205
        const int py = (int)rint((r.p.y - org_y) * map_inv_res);
        double c, s;
        cos_sin(r.p.th, c, s);
        DIAG_STAMP_U(0, 4);    // mover: cos / sin of the new heading
        const int64_t map_env = a.hot.map_shared ? 0 : g;
        int cls = kFree;
#ifdef BCP_DIAG
        if (!ABLATED(a, kAblateNoCollision | kAblateNoClassify))
#endif
        {
            // (a shared field is 18 KB for the 183 x 183 map and stays in the CU's L1: a copy in LDS measured no faster)
            if (a.hot.near) {
                cls = classify_near(outer, as_global(a.hot.near) + map_env * near_stride, map_rows, map_cols, px, py, c, s);
            } else {
                const OuterLookups look = outer_lookups_issue(*(const CullDesc*)&SL->cull, map_env, map_rows, map_cols, px, py, c, s);
                cls = outer_lookups_verdict(*(const CullDesc*)&SL->cull, look);
            }
        }
        if (!active) cls = kFree;
        DIAG_STAMP_U(0, 5);    // mover: classified
        if (cls == kAmbiguous && !ABLATED(a, kAblateNoPark)) {
            park = true;
            q.c = c;
            q.s = s;
            q.px = px;
            q.py = py;
            q.env_lo = (int32_t)(uint32_t)i;
            q.env_hi = (int32_t)(i >> 32);
        }
    } else if (scanner && !ABLATED(a, kAblateNoReward)) {
        // (3b) the reward provider for the pose as it stands if nothing collides
        const double x = hand_pose[lane], y = hand_pose[kBlock + lane], th = hand_pose[2 * kBlock + lane];
        const int64_t g = slot_of(SL, i, q);
        const int m = my_len;
        const double* gpath = a.hot.path_pts + (a.hot.path_shared ? 0 : g * (int64_t)a.hot.path_max_len * 5);
        if (PLAIN && lds_path) {
            // Shared path in LDS: the candidates of the pair's 64 envs as ONE list, shared out evenly between the three
            // scanning waves.  More than half of the envs of a rollout are nowhere near the path (no candidate at all) while
            // a few have ten: with a lane per env the waves ran as many trips as their unluckiest lane (four) for an average
            // of less than one useful candidate per lane and trip.  The scorer looks up every env's window, numbers the
            // candidates (prefix sum over the wave), writes the list -- item k -> env -- and releases it; then every wave
            // takes its share of the items (64 per wave and round), tests way point lo(env) + (k - first(env)) against the env's
            // pose and keeps the largest reached index per env with an LDS maximum.  (`hand_score` is free by now: the noise
            // it carried was read behind barrier 0.)
            __attribute__((address_space(3))) int32_t* res = (__attribute__((address_space(3))) int32_t*)hand_score;        // [64]
            __attribute__((address_space(3))) uint32_t* lo_first = (__attribute__((address_space(3))) uint32_t*)(res + kBlock);   // [64]: lo | first << 16
            __attribute__((address_space(3))) uint8_t* items = (__attribute__((address_space(3))) uint8_t*)(res + 2 * kBlock);   // [kScanItems]
            int total;
            if (member == 0) {
                const PathWindow win = path_window(P, (LdsF64)lds_box, (const __attribute__((address_space(3))) int16_t*)lds_index, x, y);
                DIAG_STAMP_U(kWScorer, 6);    // scorer: candidate window known
                const int lo = max(win.lo, q.target), hi = min(win.hi, m - 1);
                const int cnt = max(hi - lo + 1, 0);
                // inclusive prefix sum over the wave: four DPP row shifts inside every 16-lane row (lanes shifted in from
                // outside a row read 0), then the totals of the rows below (three v_readlane) -- no LDS round trip
                int incl = cnt;
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, true);   // row_shr:1
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, true);   // row_shr:2
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, true);   // row_shr:4
                incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, true);   // row_shr:8
                const int t0 = bcast_i(incl, 15), t1 = bcast_i(incl, 31), t2 = bcast_i(incl, 47);
                incl += (lane >= 16 ? t0 : 0) + (lane >= 32 ? t1 : 0) + (lane >= 48 ? t2 : 0);
                total = bcast_i(incl, 63);
                const int first = incl - cnt;
                res[lane] = -1;
                if (total <= kScanItems) {
                    lo_first[lane] = (uint32_t)lo | ((uint32_t)first << 16);
                    for (int t = 0; t < cnt; ++t) items[first + t] = (uint8_t)lane;
                }
                DIAG_MAX(9, cnt);   // longest candidate window among the lanes 0 of the workgroup's waves
                if (lane == 0) __hip_atomic_store(&ctl[8 + pair], total + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                int ready;
                BOUNDED_POLL((ready = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&ctl[8 + pair], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP))) == 0, poll_expired);
                total = ready > 0 ? ready - 1 : 0;
            }
            if (total <= kScanItems) {
                // (the scorer, who has just made the list, takes the LAST share of every round: 64 * (3 r + 2) ...)
                // (the scorer taking the FIRST share -- on the metric workload the only one that is not empty -- straight after
                //  building the list: 11.64 against 11.54 us, round 3)
                for (int base = 64 * ((member + 2) % 3); base < total; base += 64 * 3) {
                    const int k = base + lane;
                    if (k < total) {
                        const int e = (int)items[k];
                        const uint32_t lf = lo_first[e];
                        const int j = (int)(lf & 0xFFFFu) + (k - (int)(lf >> 16));
                        const LdsF64 wp = lds_path + 5 * j;
                        if (way_point_reached(P, wp[0], wp[1], wp[2], wp[3], wp[4], hand_pose[e], hand_pose[kBlock + e], hand_pose[2 * kBlock + e]))
                            __hip_atomic_fetch_max(&res[e], j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            } else {
                // (a path so dense that the pair's windows hold more than kScanItems way points: a lane per env, every third
                //  candidate per wave, as for paths in global memory)
                const PathWindow win = path_window(P, (LdsF64)lds_box, (const __attribute__((address_space(3))) int16_t*)lds_index, x, y);
                PathWindow part;
                part.hi = min(win.hi, m - 1) - member;
                part.lo = max(win.lo, q.target);
                const int last = last_reached_from(P, lds_path, part, m, q.target, x, y, th, 3);
                if (last >= 0) __hip_atomic_fetch_max(&res[lane], last, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            DIAG_STAMP_U(kWScorer, 14);    // scorer of pair 0: scanned
            DIAG_STAMP_U(kWHelper1, 7);     // helper 1 of pair 0: scanned
            DIAG_STAMP_U(kWHelper2, 8);    // helper 2 of pair 0: scanned
        } else {
        PathWindow win;
        if (SL->path.shared)
            win = path_window(P, (LdsF64)lds_box, (const __attribute__((address_space(3))) int16_t*)lds_index, x, y);
        else
            win = path_window_compact(P, box, as_global(reinterpret_cast<const uint16_t*>(L.hot.path_bbox + g * kBoxDoubles)) + kBoxIndexU16,
                                      (int)(own_len_shift >> 32), x, y);
        DIAG_STAMP_U(kWScorer, 6);    // scorer: candidate window known
        if (PLAIN) {
            // way points in memory: this member's share of the candidate window [max(lo, target), min(hi, m - 1)] is a
            // contiguous third, counted from the top (four neighbours per round trip)
            const int lo = max(win.lo, q.target), hi = min(win.hi, m - 1);
            int last;
            {
                const int third = (max(hi - lo + 1, 0) + 2) / 3;
                PathWindow part;
                part.hi = hi - member * third;
                part.lo = max(lo, part.hi - third + 1);
                if (a.hot.path_pre) {
                    // (private paths: quantised prefilter records, in steps from the corner of this path's box)
                    last = last_reached_prefiltered(P, gpath, a.hot.path_pre + g * (int64_t)a.hot.path_max_len * 2, part, m,
                                                    q.target, x, y, th, (double)box[0], (double)box[2], quant_inv);
                } else {
                    last = last_reached_from(P, gpath, part, m, q.target, x, y, th);
                }
            }
            ((__attribute__((address_space(3))) int32_t*)hand_score)[member * kBlock + lane] = last;
            DIAG_STAMP_U(kWHelper1, 7);     // helper 1 of pair 0: scanned
            DIAG_STAMP_U(kWHelper2, 8);    // helper 2 of pair 0: scanned
            DIAG_MAX(9, max(hi - lo + 1, 0));   // longest candidate window among the lanes 0 of the workgroup's waves
        } else {
            double min_dist = q.min_dist;
            int target = q.target;
            double rew;
            if (P.reward_provider == BCP_REWARD_PURE_PURSUIT) {
                if (lds_path) rew = reward_pure_pursuit(lds_path, m, x, y, q.collided != 0, min_dist, target);
                else rew = reward_pure_pursuit(gpath, m, x, y, q.collided != 0, min_dist, target);
            } else if (lds_path) {
                rew = reward_step(P, lds_path, win, m, x, y, th, min_dist, target);
            } else {
                rew = reward_step(P, gpath, win, m, x, y, th, min_dist, target);
            }
            hand_score[lane] = rew;
            hand_score[kBlock + lane] = min_dist;
            hand_score[2 * kBlock + lane] = (double)target;
        }
        }
    }
    // (An env that ends its episode this step reloads its initial state in finalize_env_from, a dependent round trip at
    //  the end of the step.  Fetching it ahead for the lanes that may end -- time-out reached, pose not cleared -- was
    //  measured in round 3: the ~60 instructions it adds to every mover cost more than the round trip they hide,
    //  12.24 against 12.00 us per step.)
    // (4) movers park the undecided poses in LDS right away (one LDS atomic per wave hands out the slots)
    __attribute__((address_space(3))) ParkedPose* my_rec = rec;
    if (mover) {
        const uint64_t parking = __ballot(park);
        if (parking) {
            const int first = (int)__ffsll((unsigned long long)parking) - 1;
            int base = 0;
            if (lane == first) base = atomicAdd((int*)&ctl[0], (int)__popcll(parking));
            base = bcast_i(base, first);
            if (park) {
                my_rec = rec + base + (int)__popcll(parking & ((1ull << lane) - 1ull));
                my_rec->c = q.c;
                my_rec->s = q.s;
                my_rec->px = q.px;
                my_rec->py = q.py;
                my_rec->env_lo = q.env_lo;
                my_rec->env_hi = q.env_hi;
                my_rec->geom = q.geom;
                my_rec->verdict = 0;
                my_rec->spec_ok = 0;
            }
        }
    }
    DIAG_STAMP(4);        // mover: classified and parked
    DIAG_STAMP_W(kWScorer, 8);   // scorer of pair 0: scanned
    // "Barrier 2" is two counters in LDS instead of an s_barrier: a mover only needs the scan results of ITS pair, and the
    // other waves only need every mover's poses parked -- the waves that finish their scan first start on the parked
    // poses while the slowest scan of the workgroup is still running (release adds / acquire polls, workgroup scope).
    if (mover) {
        if (lane == 0) __hip_atomic_fetch_add(&ctl[2], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        // Private paths: a parked pose that turns out to collide is rolled back, and the reward provider then runs for
        // the OLD pose -- window lookup, way-point scan, distance to the target: five to ten dependent round trips to
        // memory by one lane at the very end of the step, with the rest of the workgroup idle (C4: the workgroups that
        // end last all hold such an env).  The mover has nothing to do until its pair's scans are in, so it works that
        // reward out now for every pose it parked; the verdict then only picks between two finished results.
        if (PLAIN && !lds_path && !hot_path_shared && a.hot.path_pre && park && !ABLATED(a, kAblateNoReward)) {
            const int64_t g = slot_of(SL, i, q);
            const GlobalPtr<const float> bx = as_global(reinterpret_cast<const float*>(L.hot.path_bbox + g * kBoxDoubles));
            float obox[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) obox[k] = bx[k];
            const PathWindow ow = path_window_compact(P, obox, as_global(reinterpret_cast<const uint16_t*>(L.hot.path_bbox + g * kBoxDoubles)) + kBoxIndexU16,
                                                      (int)(own_len_shift >> 32), q.old.x, q.old.y);
            const double* opath = SL->path.pts + g * (int64_t)SL->path.max_len * 5;
            const int olast = last_reached_prefiltered(P, opath, a.hot.path_pre + g * (int64_t)a.hot.path_max_len * 2, ow, my_len,
                                                       q.target, q.old.x, q.old.y, q.old.th, (double)obox[0], (double)obox[2],
                                                       bx[kBoxQuantStep + 1]);
            double omin = q.min_dist;
            int otarget = q.target;
            my_rec->spec_rew = reward_from_last(P, opath, my_len, olast, q.old.x, q.old.y, omin, otarget);
            my_rec->spec_min = omin;
            my_rec->spec_target = otarget;
            my_rec->spec_ok = 1;
        }
        const int scans = PLAIN ? 3 : 1;
        BOUNDED_POLL(__builtin_amdgcn_readfirstlane(__hip_atomic_load(&ctl[4 + pair], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) < scans, poll_expired);
    } else if (scanner && lane == 0) {
        __hip_atomic_fetch_add(&ctl[4 + pair], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // (everybody: the number of parked poses must be final before a wave draws tickets -- a ticket is consumed by the
    //  draw, so a wave that compared it with a stale count would drop a pose, and its owner would wait for ever)
    BOUNDED_POLL(__builtin_amdgcn_readfirstlane(__hip_atomic_load(&ctl[2], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) < kLocalPairs, poll_expired);
    DIAG_STAMP(5);
    // (scalar: the ticket loop below must stay wave-uniform)
    const int n_parked = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    // (5) movers: the rest of the reward provider, handed to the parked records, then the decided envs are finished;
    //     everybody else goes straight to the parked poses
    ScoredFree sc;
    sc.rew = 0.0;
    sc.min_dist = 0.0;
    sc.target = 0;
    if (mover) {
        if (PLAIN) {
            const __attribute__((address_space(3))) int32_t* found = (__attribute__((address_space(3))) int32_t*)hand_score;
            const int last = lds_path ? found[lane] : max(max(found[lane], found[kBlock + lane]), found[2 * kBlock + lane]);
            const int64_t g = slot_of(SL, i, q);
            const int m = my_len;
            sc.min_dist = q.min_dist;
            sc.target = q.target;
            if (lds_path)
                sc.rew = reward_from_last(P, lds_path, m, last, r.p.x, r.p.y, sc.min_dist, sc.target);
            else
                sc.rew = reward_from_last(P, SL->path.pts + (SL->path.shared ? 0 : g * (int64_t)SL->path.max_len * 5), m,
                                          last, r.p.x, r.p.y, sc.min_dist, sc.target);
        } else {
            sc.rew = hand_score[lane];
            sc.min_dist = hand_score[kBlock + lane];
            sc.target = (int)hand_score[2 * kBlock + lane];
        }
        DIAG_STAMP(6);    // mover: reward provider done
        // (Round 4: the envs are finished in ONE pass at the end, the decided ones together with the parked ones.  Rounds 2-3
        //  finished the decided lanes here and the parked ones behind their verdicts: two passes of the same ~2.5 k-cycle
        //  latency chain -- stores, the reset loads of lanes whose episode ends -- in the waves that end the step, and two
        //  inlined copies of finalize_env_from in a kernel larger than the instruction cache.)
    }
    DIAG_STAMP_W(kWHelper1, 9);   // helper: past the second barrier
    const double vqx = lane < P.n_verts ? qv[2 * lane] : 0.0, vqy = lane < P.n_verts ? qv[2 * lane + 1] : 0.0;   // (row-by-row fallback)
    // (6) every wave settles parked poses, a ticket at a time: exact test, verdict into the record.
    // (Control flow: a scalar loop condition and NO single-lane region around the draw.  With two `if (lane == 0)` regions
    //  per trip hipcc 7.2 threaded lane 0's path across the back edge and split the loop in two; lanes 1..63 then span in
    //  the inner one on ticket 0 for ever while lane 0 waited outside it.  Round 2 kept one such region per trip; round 3
    //  draws with every lane, so no edit or compiler update can bring that shape back.)
    DIAG_STAMP(7);        // mover: decided envs finished
    // (One pose tested by two or three waves -- every n-th row of the footprint's image each, the verdicts meeting in one LDS
    //  add -- when few poses are parked and most waves would find no ticket: measured in round 3, 11.78 against 11.55 us.  The
    //  edge set-up and the cell list are repeated by every share, and a test is not long because of its cells: with three
    //  shares the longest tests still took 6.5 k cycles.)
    // (the draw has no single-lane region: every lane issues an LDS add -- lane 0 adds 1 to the ticket counter, the others add
    //  0 to a word of their own in the wave's cell list, which changes nothing: 64 lanes on ONE word are 64 serialised
    //  atomics, and 16 waves drawing at once cost the step 4 us that way -- and lane 0's returned value is the ticket)
    __attribute__((address_space(3))) int* const ticket_word =
        lane == 0 ? (__attribute__((address_space(3))) int*)&ctl[1] : (__attribute__((address_space(3))) int*)(cell_list + lane);
    int ticket = __builtin_amdgcn_readfirstlane(__hip_atomic_fetch_add(ticket_word, lane == 0 ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    while (ticket < n_parked) {
        DIAG_STAMP_W(kWHelper1, 10);   // helper: has a ticket
        __attribute__((address_space(3))) ParkedPose* e = rec + ticket;
        const double c = e->c, s = e->s;
        const int px = e->px, py = e->py;
        const int64_t env = ((int64_t)e->env_hi << 32) | (uint32_t)e->env_lo;
        const int64_t g = a.hot.map_shared ? 0 : (a.hot.geom_of_env ? (int64_t)e->geom : env);
        const uint32_t* words = a.hot.map_bits + g * a.hot.map_env_stride;
        const uint32_t* tiles = a.map_tiles + g * map_tile_words(a.hot.map_rows, a.hot.map_wpr);
        bool h = false;
        [[maybe_unused]] const unsigned long long test_from = DIAG_NOW();
        [[maybe_unused]] int how = 0;
        if (!ABLATED(a, kAblateNoCoop)) {
            // the lethal cells under the image tested one by one; a map too dense for that is rasterised row by row
#ifdef BCP_DIAG
            unsigned long long phase[3] = {test_from, test_from, 0};
            unsigned long long* const phases = phase;
#else
            unsigned long long* const phases = nullptr;
#endif
            const int verdict = map_words
                ? coop_collides_sparse<WIDE>(P, (LdsF64)qv, c, s, px, py, (LdsWords)lds_map, a.hot.map_rows, a.hot.map_cols,
                                             a.hot.map_wpr, cell_list, phases)
                : coop_collides_sparse<WIDE, true>(P, (LdsF64)qv, c, s, px, py, as_global(tiles), a.hot.map_rows, a.hot.map_cols,
                                                   a.hot.map_wpr, cell_list, phases);
#ifdef BCP_DIAG
            // wave 8's test: cycles for the edge set-up, for listing the cells, for the rest, and the list's length
            if (threadIdx.x == kWHelper1 * 64 && blockIdx.x < 2048) {
                const unsigned long long now = DIAG_NOW();
                g_diag[(2048 + blockIdx.x) * 16 + 13] = ((phase[0] - test_from) << 40) | ((phase[1] - phase[0]) << 20) | (now - phase[1]);
                g_diag[(2048 + blockIdx.x) * 16 + 15] = phase[2];
            }
#endif
            h = verdict == kSparseHit;
            if (verdict == kSparseTooMany)
                h = coop_collides<WIDE>(P, vqx, vqy, c, s, px, py, words, a.hot.map_rows, a.hot.map_cols, a.hot.map_wpr);
            how = verdict;
        }
        DIAG_MAX(0, ((DIAG_NOW() - test_from) << 4) | (unsigned)how);   // the longest exact test of the workgroup, and its kind
        DIAG_STAMP_W(kWHelper1, 11);   // helper: verdict
        bool post = lane == 0;
#ifdef BCP_DIAG
        post = post && !(a.flags & kDiagWithholdVerdicts);
#endif
        if (post) __hip_atomic_store(&e->verdict, h ? 2 : 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        ticket = __builtin_amdgcn_readfirstlane(__hip_atomic_fetch_add(ticket_word, lane == 0 ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        DIAG_STAMP_W(kWHelper1, 12);   // helper: verdict posted
    }
    // (7) movers: the envs they parked are finished by the lane that holds their state, as soon as the verdicts are in
    //     (every parked pose has been claimed by now -- by this wave or by one that is working on it)
    if (mover) {
        int verdict = park ? 0 : 1;
        {
            int trips = 0;
            while (__ballot(verdict == 0)) {
                if (verdict == 0) verdict = __hip_atomic_load(&my_rec->verdict, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                __builtin_amdgcn_s_sleep(1);
                if (++trips > kPollLimit) {   // (a verdict that never comes: the env is finished as free, and the step says so)
                    poll_expired = true;
                    if (verdict == 0) {
                        verdict = 1;
                        q.err |= BCP_ERR_INTERNAL;
                    }
                    break;
                }
            }
        }
        // A pose that really collides is rolled back, and the reward provider runs for the OLD pose: the scan for its last
        // reached way point by the whole wave, a candidate per lane (one lane walking the window alone took ~5 k cycles, and
        // the step ends with the workgroups that hold such an env).  Shared path in LDS, continuous provider without delays;
        // otherwise finalize_env_from computes it itself.
        const bool hit_score = PLAIN && lds_path && a.hot.path_shared && !ABLATED(a, kAblateNoReward);
        int last_hit = -1;
        if (hit_score) {
            uint64_t hits = __ballot(park && verdict == 2);
            while (hits) {
                const int src = (int)__ffsll((unsigned long long)hits) - 1;
                hits &= hits - 1;
                const double ox = bcast_d(q.old.x, src), oy = bcast_d(q.old.y, src), oth = bcast_d(q.old.th, src);
                const int tg = bcast_i(q.target, src);
                const PathWindow w = path_window(P, (LdsF64)lds_box, (const __attribute__((address_space(3))) int16_t*)lds_index, ox, oy);
                const int found = coop_last_reached(P, lds_path, w, my_len, tg, ox, oy, oth);
                if (lane == src) last_hit = found;
            }
        }
        if (active) {
            const bool hit_now = park && verdict == 2;
            bool fits = hit_score && hit_now;
            if (fits) {
                sc.min_dist = q.min_dist;
                sc.target = q.target;
                sc.rew = reward_from_last(P, lds_path, my_len, last_hit, q.old.x, q.old.y, sc.min_dist, sc.target);
            } else if (PLAIN && hit_now && my_rec->spec_ok) {   // (private path: worked out ahead, see above)
                sc.rew = my_rec->spec_rew;
                sc.min_dist = my_rec->spec_min;
                sc.target = my_rec->spec_target;
                fits = true;
            }
            if (!PLAIN) {
#pragma unroll
                for (int k = 0; k < 3; ++k) q.popped_pose[k] = fifo_stash[k * kLocalEnvs];
#pragma unroll
                for (int k = 0; k < 7; ++k) q.popped_state[k] = fifo_stash[(3 + k) * kLocalEnvs];
            }
            finalize_env_from<PLAIN>(a, SL, i, q, hit_now, lds_path, nullptr, !ABLATED(a, kAblateNoReward), sc, my_len, fits, out_base);
        }
    }
    DIAG_STAMP(13);            // mover: out of tickets
    DIAG_STAMP_W(kWHelper1, 14);       // helper: out of tickets
#ifdef BCP_DIAG
    if (tid == 0 && blockIdx.x < kDiagBlocks) g_diag[blockIdx.x * 16 + 15] = (unsigned long long)n_parked;
#endif
    if (poll_expired && lane == 0) atomicAdd(reinterpret_cast<unsigned long long*>(a.tick + 4), 1ull);   // (bcp_expired_waits)
    DIAG_REAL_EXIT();
    // The last workgroup to get here moves the step counter on (every workgroup has read it long before it draws).  The ticket
    // has a cache line of its own, and it is drawn HERE, in the last instructions of a wave that is done before the movers are:
    // measured in round 4 (tools/step_time.py, builds side by side on one box) -- no ticket at all, workgroup 0 moving the
    // counter on: 11.7 us against 11.8 with the ticket, i.e. it hides behind the movers; the ticket drawn early (behind barrier
    // 0 or barrier 1, looked at here): 13.3 - 13.8 us -- 256 workgroups draw at the same moment, the returning atomics queue
    // at the memory side (~10 ns each on one address), and the drawing wave waits for its ticket at its next wait of any kind,
    // which holds up its share of the way-point scan and with it its pair's mover; a second atomic per workgroup on the
    // ticket's line (the parked-pose count, first form): 14.3 us.  The parked poses are counted in a word per workgroup instead.
    if ((a.flags & kStepAdvances) && tid == (kLocalWaves - 1) * kBlock) {
        if (n_parked) a.parked_slots[blockIdx.x] += (uint64_t)n_parked;   // (this workgroup's word: no atomic; bcp_parked_poses adds them up)
        if (!ROLL || rs == a.rollout_steps - 1) {
            const unsigned int ticket_drawn =
                __hip_atomic_fetch_add((GlobalPtr<unsigned int>)(a.tick + kTickLocalTicket), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ticket_drawn == gridDim.x - 1) {
                *reinterpret_cast<unsigned int*>(a.tick + kTickLocalTicket) = 0u;
                a.tick[0] = step_counter + 1;
            }
        }
    }
}

// A rollout's trips are CALLS of one out-of-line copy of the step: written as a loop around the inlined body the compiler
// hoisted the body's loop-invariant values -- launch arguments, addresses, the few hundred float64 constants of its
// transcendental functions -- in front of the loop and carried them across it (300 - 500 scalar and ~100 vector spills,
// 300 bytes of scratch per lane); a call boundary keeps every trip's code what the one-step kernel's is.
template <bool WIDE, bool PLAIN, int PAIRS>
__device__ __attribute__((noinline)) void step_local_trip(uint64_t kernarg_bits, int rs)
{
    // (arguments of a device function arrive in vector registers: the address of the launch arguments and the trip number are
    //  made scalar again.  The kernarg-segment builtin is no way to the arguments from inside a callee: the first form of
    //  this function used it and read address 0.)
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)kernarg_bits);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(kernarg_bits >> 32));
    step_local_body<WIDE, PLAIN, PAIRS, true>((KernArgPtr)(uintptr_t)(((uint64_t)hi << 32) | lo), __builtin_amdgcn_readfirstlane(rs));
}

template <bool WIDE, bool PLAIN, int PAIRS, bool ROLL = false>
__global__ void __attribute__((amdgpu_flat_work_group_size(4 * PAIRS * kBlock, 4 * PAIRS * kBlock), amdgpu_waves_per_eu(4)))
step_local_kernel(const StepArgs launch_args)
{
    KernArgPtr kernarg = (KernArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    if constexpr (!ROLL) {
        step_local_body<WIDE, PLAIN, PAIRS, false>(kernarg, 0);
    } else {
        const int n_steps = kernarg->rollout_steps;
        for (int rs = 0; rs < n_steps; ++rs) {
            step_local_trip<WIDE, PLAIN, PAIRS>((uint64_t)(uintptr_t)kernarg, rs);
            // the next step: the control words, the hand-over buffers and the parked records are rewritten from here on, and the
            // scanners read what this step's movers have just stored (workgroup scope: the waves share the compute unit's L1)
            __syncthreads();
        }
    }
}
