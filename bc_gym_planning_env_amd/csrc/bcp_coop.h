// bcp_coop.h -- wave-cooperative exact footprint test and the distance-field pre-classification.
//
// 1. classify(): a per-env O(1) test that settles most poses without rasterising anything.  It reads a Euclidean
//    distance transform of the lethal cells (uint8 floor of the distance in pixels, built once per costmap) at a
//    few sample points on the robot's long axis:
//      * every footprint pixel lies within R_out of some OUTER sample  -> all samples farther than R_out from any
//        lethal cell  => no collision;
//      * a disc of radius R_in around an INNER sample lies inside the filled footprint -> a lethal cell closer than
//        that => collision.
//    The radii carry the worst-case pixel slack of the reference pipeline (vertex rounding 0.7071 px, Bresenham
//    0.5 px, 16.16 truncation, sample-centre rounding 0.7071 px), so both verdicts are provably what the exact test
//    would return; everything else is AMBIGUOUS and goes to 2.
// 2. coop_collides(): the exact test for ONE pose executed by all 64 lanes of a wave: lane = mask row, loop over
//    polygon edges with the edge parameters broadcast from the lane that owns the edge.  Row coverage is built as
//    bit masks:  OUTLINE runs are OR-ed in,  SPANS use the parity form of the even-odd scanline
//        x is inside a span   <=>   #{active edges with x_e < x} is odd   (or x == x_e, which is an OUTLINE pixel)
//    i.e. XOR of suffix masks starting at floor(x_e) + 1 -- no sorting, any contour.
#pragma once

#include "bcp_raster.h"

namespace bcp {

constexpr int kMaxSamples = 8;

// distance-field description (kernel argument)
struct CullDesc {
    const uint8_t* edt;   // [(rows + 2 pad) * (cols + 2 pad)] floor(min(clamp, distance to nearest lethal cell))
    int64_t env_stride;   // bytes per env (0: one field shared by all envs)
    int32_t on;           // 0: no distance field -> every in-map pose is AMBIGUOUS
    int32_t pad, width, height;  // padding on each side, padded row width / row count
    int32_t clamp;        // the field saturates at this distance
    int32_t reach;        // any footprint pixel is within `reach` px of the robot pixel (off-map test)
    int32_t n_out, n_in;
    int32_t t_out;        // free  <=>  edt >= t_out at every outer sample
    int32_t t_in[kMaxSamples];   // hit <=  edt <= t_in[j] at inner sample j
    double out_x[kMaxSamples], in_x[kMaxSamples];  // sample abscissae on the robot axis, in pixels
    double axis_y;        // ordinate of the sample axis in the robot frame, in pixels
    // The outer test only asks "is a lethal cell closer than t_out": the field as ONE BIT per cell, in tiles of
    // 32 x 32 cells (32 row words = one 128-byte line each; the samples of a pose lie on a line of <= 2 reach px, so
    // they meet three to five lines instead of one each).  word = near[((y >> 5) * near_tx + (x >> 5)) * 32 + (y & 31)]
    const uint32_t* near;
    int64_t near_stride;  // words per env (0: shared)
    int32_t near_tx, near_words;   // tiles per tile row; words of one entry
    // What step_local_kernel's outer test reads: `near` itself (shift 0), or a copy at 1/2 or 1/4 of the resolution -- a bit
    // of it is the OR of the 2 x 2 / 4 x 4 cells it stands for (near_coarsen_kernel), so "not near" still holds for every
    // one of them.  A 128-byte line then covers 64 x 64 / 128 x 128 cells: the samples of a pose meet fewer lines (each a
    // full line of memory traffic for maps that do not stay in cache), at the price of a few more undecided poses.
    const uint32_t* step_near;
    int64_t step_near_stride;
    int32_t step_near_tx, step_near_shift;
};

enum { kFree = 0, kHit = 1, kAmbiguous = 2 };

// distance-field value at padded cell (x, y); `outside` when the cell is not stored (small padding of private maps)
__device__ __forceinline__ int edt_at(const CullDesc& C, const uint8_t* field, int x, int y, int outside)
{
    if ((unsigned)x >= (unsigned)C.width || (unsigned)y >= (unsigned)C.height) return outside;
    return (int)field[y * C.width + x];
}

// outer test: kFree, or kAmbiguous when a lethal cell may touch the footprint
__device__ __forceinline__ int classify_outer(const CullDesc& C, int64_t env, int rows, int cols, int px, int py, double c,
                                              double s)
{
    // the whole kernel image misses the map -> nothing to collide with (env.py:483-484 drops off-map cells)
    if (px + C.reach < 0 || px - C.reach >= cols || py + C.reach < 0 || py - C.reach >= rows) return kFree;
    if (!C.on) return kAmbiguous;
    const double ay_c = C.axis_y * c, ay_s = C.axis_y * s;
    const uint8_t* field = C.edt + env * C.env_stride;
    // fully unrolled over the (at most kMaxSamples) samples: addresses first, then all loads, then the compares --
    // so the byte loads are in flight together instead of one L2 round trip per sample
    int idx[kMaxSamples];
#pragma unroll
    for (int i = 0; i < kMaxSamples; ++i) {
        const int du = (int)rint(C.out_x[i] * c - ay_s), dv = (int)rint(C.out_x[i] * s + ay_c);
        const int x = px + C.pad + du, y = py + C.pad + dv;
        const bool stored = i < C.n_out && (unsigned)x < (unsigned)C.width && (unsigned)y < (unsigned)C.height;
        idx[i] = stored ? y * C.width + x : -1;
    }
    // a sample outside the stored (padded) rectangle is more than `pad` px away from every cell of the map
    const int not_stored = min(C.pad + 1, 255);
    int val[kMaxSamples];
#pragma unroll
    for (int i = 0; i < kMaxSamples; ++i) val[i] = idx[i] >= 0 ? (int)field[idx[i]] : (i < C.n_out ? not_stored : 255);
    int near_out = 0;
#pragma unroll
    for (int i = 0; i < kMaxSamples; ++i) near_out |= val[i] < C.t_out;
    return near_out ? kAmbiguous : kFree;
}

// The outer test in two halves, so that a caller can put other loads between issuing the lookups and using them.
struct OuterLookups {
    int val[kMaxSamples];
    bool off_map;
};

__device__ __forceinline__ OuterLookups outer_lookups_issue(const CullDesc& C, int64_t env, int rows, int cols, int px,
                                                            int py, double c, double s)
{
    OuterLookups L;
    L.off_map = px + C.reach < 0 || px - C.reach >= cols || py + C.reach < 0 || py - C.reach >= rows;
    const double ay_c = C.axis_y * c, ay_s = C.axis_y * s;
    const GlobalPtr<const uint8_t> field = as_global(C.edt) + env * C.env_stride;
    // a sample outside the stored (padded) rectangle is more than `pad` px away from every cell of the map
    const int not_stored = min(C.pad + 1, 255);
#pragma unroll
    for (int i = 0; i < kMaxSamples; ++i) {
        const int du = (int)rint(C.out_x[i] * c - ay_s), dv = (int)rint(C.out_x[i] * s + ay_c);
        const int x = px + C.pad + du, y = py + C.pad + dv;
        const bool stored = !L.off_map && i < C.n_out && (unsigned)x < (unsigned)C.width && (unsigned)y < (unsigned)C.height;
        L.val[i] = stored ? (int)field[y * C.width + x] : (i < C.n_out ? not_stored : 255);
    }
    return L;
}

// What the outer test needs of a CullDesc, as a value: a kernel fetches it (scalar loads) before its barriers, so that
// the loads are not queued behind them on the critical path.
struct OuterParams {
    int32_t reach, pad, width, height, n_out, near_tx, near_shift, t_out;
    double out_x[kMaxSamples], axis_y;
};

__device__ __forceinline__ OuterParams outer_params(const CullDesc& C)
{
    OuterParams o;
    o.reach = C.reach;
    o.pad = C.pad;
    o.width = C.width;
    o.height = C.height;
    o.n_out = C.n_out;
    o.near_tx = C.near_tx;
    o.near_shift = 0;
    o.t_out = C.t_out;
#pragma unroll
    for (int i = 0; i < kMaxSamples; ++i) o.out_x[i] = C.out_x[i];
    o.axis_y = C.axis_y;
    return o;
}

// the same lookups from the 1-bit tiles (CullDesc::near) of this pose's map entry, in global memory or staged in LDS
template <typename WordPtr>
__device__ __forceinline__ OuterLookups outer_lookups_near(const OuterParams& C, WordPtr tiles, int rows, int cols, int px, int py,
                                                           double c, double s)
{
    OuterLookups L;
    L.off_map = px + C.reach < 0 || px - C.reach >= cols || py + C.reach < 0 || py - C.reach >= rows;
    const double ay_c = C.axis_y * c, ay_s = C.axis_y * s;
    const int not_stored = min(C.pad + 1, 255);
    uint32_t word[kMaxSamples];
    int bit[kMaxSamples];
#pragma unroll
    for (int i = 0; i < kMaxSamples; ++i) {
        const int du = (int)rint(C.out_x[i] * c - ay_s), dv = (int)rint(C.out_x[i] * s + ay_c);
        const int x = px + C.pad + du, y = py + C.pad + dv;
        const bool stored = !L.off_map && i < C.n_out && (unsigned)x < (unsigned)C.width && (unsigned)y < (unsigned)C.height;
        bit[i] = stored ? (x & 31) : -1;
        word[i] = stored ? tiles[((y >> 5) * C.near_tx + (x >> 5)) * 32 + (y & 31)] : 0u;
    }
#pragma unroll
    for (int i = 0; i < kMaxSamples; ++i)   // as the byte field reads: 0 = closer than t_out, 255 = not
        L.val[i] = bit[i] >= 0 ? (((word[i] >> bit[i]) & 1u) ? 0 : 255) : (i < C.n_out ? not_stored : 255);
    return L;
}

// The outer test on the 1-bit tiles in one piece, as step_local_kernel's movers run it: straight-line code (a sample that
// is not stored reads word 0 and is masked out afterwards: no branch per sample, all eight loads in flight together), and
// the sample offsets in float32.  The offsets only select WHICH pixel stands for a sample; kSlackOuter budgets 0.7072 px
// for that choice against the 0.70711 px of an exactly rounded centre, and a float32 offset (|offset| < 128 px, relative
// error 1.2e-7 in cos / sin and in the fused multiply-add) is within 3e-5 px of the float64 one: a tie can fall the
// other way, to a pixel 0.50003 px from the centre per axis, 0.70715 px in all -- inside the budget, so "free" still
// means what it means in classify_outer (the verdict tests against the oracle cover both forms).
template <typename WordPtr>
__device__ __forceinline__ int classify_near(const OuterParams& C, WordPtr tiles, int rows, int cols, int px, int py, double c,
                                             double s)
{
    if (px + C.reach < 0 || px - C.reach >= cols || py + C.reach < 0 || py - C.reach >= rows) return kFree;
    const float cf = (float)c, sf = (float)s, ay = (float)C.axis_y;
    const float ay_c = ay * cf, ay_s = ay * sf;
    const int x0 = px + C.pad, y0 = py + C.pad;
    uint32_t word[kMaxSamples];
    int bit[kMaxSamples];
#pragma unroll
    for (int i = 0; i < kMaxSamples; ++i) {
        const float ox = (float)C.out_x[i];
        const int x = x0 + (int)rintf(fmaf(ox, cf, -ay_s)), y = y0 + (int)rintf(fmaf(ox, sf, ay_c));
        const bool stored = (i < C.n_out) & ((unsigned)x < (unsigned)C.width) & ((unsigned)y < (unsigned)C.height);
        const int xs = x >> C.near_shift, ys = y >> C.near_shift;   // (the tiles may be at a fraction of the resolution)
        const int at = ((ys >> 5) * C.near_tx + (xs >> 5)) * 32 + (ys & 31);
        bit[i] = stored ? (xs & 31) : 32;
        word[i] = tiles[stored ? at : 0];
    }
    // a sample outside the stored (padded) rectangle is more than `pad` px away from every cell of the map
    const bool unstored_is_near = min(C.pad + 1, 255) < C.t_out;
    uint32_t near = 0;
#pragma unroll
    for (int i = 0; i < kMaxSamples; ++i)
        near |= bit[i] < 32 ? ((word[i] >> bit[i]) & 1u) : (uint32_t)((i < C.n_out) & unstored_is_near);
    return near ? kAmbiguous : kFree;
}

__device__ __forceinline__ int outer_lookups_verdict(const OuterParams& C, const OuterLookups& L)
{
    if (L.off_map) return kFree;
    int near_out = 0;
#pragma unroll
    for (int i = 0; i < kMaxSamples; ++i) near_out |= L.val[i] < C.t_out;
    return near_out ? kAmbiguous : kFree;
}

__device__ __forceinline__ int outer_lookups_verdict(const CullDesc& C, const OuterLookups& L)
{
    if (L.off_map) return kFree;
    int near_out = 0;
#pragma unroll
    for (int i = 0; i < kMaxSamples; ++i) near_out |= L.val[i] < C.t_out;
    return near_out ? kAmbiguous : kFree;
}

// inner test for a pose the outer test could not clear: true => certainly colliding
__device__ __forceinline__ bool classify_inner_hit(const CullDesc& C, int64_t env, int px, int py, double c, double s)
{
    if (!C.on) return false;
    const double ay_c = C.axis_y * c, ay_s = C.axis_y * s;
    const uint8_t* field = C.edt + env * C.env_stride;
    // unrolled like the outer test: all lookups in flight together
    int idx[kMaxSamples];
#pragma unroll
    for (int j = 0; j < kMaxSamples; ++j) {
        const int du = (int)rint(C.in_x[j] * c - ay_s), dv = (int)rint(C.in_x[j] * s + ay_c);
        const int x = px + C.pad + du, y = py + C.pad + dv;
        const bool stored = j < C.n_in && (unsigned)x < (unsigned)C.width && (unsigned)y < (unsigned)C.height;
        idx[j] = stored ? y * C.width + x : -1;
    }
    int val[kMaxSamples];
#pragma unroll
    for (int j = 0; j < kMaxSamples; ++j) val[j] = idx[j] >= 0 ? (int)field[idx[j]] : 255;  // not stored: no verdict
    int hit_in = 0;
#pragma unroll
    for (int j = 0; j < kMaxSamples; ++j) hit_in |= (j < C.n_in) & (val[j] <= C.t_in[j]);
    return hit_in != 0;
}

__device__ __forceinline__ int classify(const CullDesc& C, int64_t env, int rows, int cols, int px, int py, double c,
                                        double s)
{
    const int cls = classify_outer(C, env, rows, cols, px, py, c, s);
    if (cls != kAmbiguous) return cls;
    return classify_inner_hit(C, env, px, py, c, s) ? kHit : kAmbiguous;
}

// ---- wave helpers -------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)__lane_id(); }

__device__ __forceinline__ int bcast_i(int v, int src) { return __builtin_amdgcn_readlane(v, src); }

__device__ __forceinline__ double bcast_d(double v, int src)
{
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, src);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), src);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

// orders the LDS accesses of the lanes of ONE wavefront (a lane reads what another lane of the same wave wrote)
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// bits [pos, 32*NW) of an NW-word row mask, word w
__device__ __forceinline__ uint32_t suffix_word(int pos, int w)
{
    const int rel = pos - 32 * w;
    return rel <= 0 ? 0xFFFFFFFFu : (rel >= 32 ? 0u : 0xFFFFFFFFu << rel);
}

// Per-lane description of edge l = (V[l-1] -> V[l]); lanes >= K hold an inert edge.
struct EdgeRegs {
    int y0, y1;        // span activity y0 <= y < y1 (y0 == y1: horizontal, never active)
    int x0fp, dxfp;    // 16.16 x at y0 and slope (CollectPolyEdges)
    int sx, sy;        // Bresenham start point (end with the smaller x)
    int dx, dy;        // |dx|, |dy| after the left-to-right normalisation
    int ystep;         // +1 / -1
    uint32_t inv;      // floor(2^32 / D) + 1 with D = 2*dy: floor(n / D) == umulhi(n, inv) for n * D < 2^32
};

// Row coverage sink of the cooperative rasteriser: called once per 64-row chunk with each lane's row masks.
//   bool rows(int y /*this lane's centred row*/, bool valid, const uint32_t cover[NW], int ubase /*centred u of bit 0*/)
// returns a wave-uniform "stop".
//
// Work split between the waves of one workgroup (all on the same pose): wave = (chunk slot, edge slot).
//   chunks: a wave takes the 64-row chunks first_chunk, first_chunk + chunk_stride, ...
//   edges : with ESPLIT == 2, two waves share a chunk, one taking the even and one the odd edges; outline runs just
//           OR together, but the span parity needs every edge, so the partners exchange their partial XOR masks
//           through LDS (`xch`, double-buffered, one workgroup barrier per round -- every wave of the workgroup
//           runs the same number of rounds).
template <int NW, int ESPLIT, typename RowSink>
__device__ __forceinline__ bool coop_raster(const DevParams& P, double qx, double qy, double c, double s, RowSink& sink,
                                            int first_chunk = 0, int chunk_stride = 1, int first_edge = 0,
                                            LdsU32 xch = nullptr, int wave = 0)
{
    const int K = P.n_verts;
    const int lane = lane_id();
    // ---- lane k < K owns vertex k and edge k
    int u = 0, v = 0;
    if (lane < K) {  // (qx, qy) = footprint vertex `lane` divided by the resolution, supplied by the caller
        u = (int)rint(fma(qy, -s, qx * c));   // path_tools.py:142-150
        v = (int)rint(fma(qy, c, qx * s));
    }
    const int prev = lane == 0 ? K - 1 : lane - 1;
    const int up = __shfl(u, prev), vp = __shfl(v, prev);
    const bool owner = lane < K;
    // extents of the integer polygon: a scalar loop over the K owner lanes (v_readlane + s_min / s_max)
    int vmin = 0x7fffffff, vmax = -0x7fffffff, umin = 0x7fffffff, umax = -0x7fffffff;
    for (int k = 0; k < K; ++k) {
        const int uk = bcast_i(u, k), vk = bcast_i(v, k);
        vmin = min(vmin, vk);
        vmax = max(vmax, vk);
        umin = min(umin, uk);
        umax = max(umax, uk);
    }
    sink.extent(umin, umax);
    EdgeRegs E;
    {
        // span edge (CollectPolyEdges): top = end with the smaller y
        const int ddy = v - vp;
        E.y0 = min(v, vp);
        E.y1 = owner ? max(v, vp) : E.y0;
        E.x0fp = (vp < v ? up : u) << 16;
        E.dxfp = ddy != 0 ? ((u - up) * 65536) / ddy : 0;
        // Bresenham (LineIterator, leftToRight): start from the end with the smaller x
        int sx = up, sy = vp, dx = u - up, dy = v - vp;
        if (dx < 0) {
            dx = -dx;
            dy = -dy;
            sx = u;
            sy = v;
        }
        E.ystep = 1;
        if (dy < 0) {
            dy = -dy;
            E.ystep = -1;
        }
        E.sx = sx;
        E.sy = sy;
        E.dx = dx;
        E.dy = owner ? dy : -1;  // dy < 0: inert
        // floor(2^32 / D) + 1 through an fp64 quotient: exact, because 2^32 / D is either an integer (D a power of
        // two) or at least 1/D >= 2^-9 away from one, far more than the quotient's rounding error (2^-21)
        const uint32_t D = 2u * (uint32_t)(dy > 0 ? dy : 1);
        E.inv = (uint32_t)(4294967296.0 / (double)D) + 1u;
    }
    const int ubase = umin;  // bit 0 of the row masks <-> centred column umin

    const int n_chunks = (vmax - vmin) / 64 + 1;
    const int n_rounds = (n_chunks + chunk_stride - 1) / chunk_stride;  // the same for every wave of the workgroup
    bool stop = false;
    for (int round = 0; round < n_rounds; ++round) {
        const int ybase = vmin + 64 * (first_chunk + round * chunk_stride);
        const int y = ybase + lane;
        const bool valid = ybase <= vmax && y <= vmax;
        uint32_t cov_or[NW], cov_xor[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) cov_or[w] = cov_xor[w] = 0;
        // no lethal cell under the image on any row of this chunk: nothing to rasterise
        const bool matters = !stop && ybase <= vmax && sink.chunk_matters(y, valid);
        if (matters) {
            for (int e = first_edge; e < K; e += ESPLIT) {
                const int ey0 = bcast_i(E.y0, e), ey1 = bcast_i(E.y1, e);
                // rows an edge can touch: [y0, y1] for the outline, [y0, y1) for the spans (horizontal: y0 == y1)
                if (ey1 < ybase || ey0 > ybase + 63) continue;
                const int sy = bcast_i(E.sy, e), sx = bcast_i(E.sx, e);
                const int dx = bcast_i(E.dx, e), dy = bcast_i(E.dy, e), ystep = bcast_i(E.ystep, e);
                // SPANS: crossing of the active edge, parity mask starts one past floor(x_e)
                if (y >= ey0 && y < ey1) {
                    const int xe = bcast_i(E.x0fp, e) + (y - ey0) * bcast_i(E.dxfp, e);
                    const int pos = (xe >> 16) + 1 - ubase;
#pragma unroll
                    for (int w = 0; w < NW; ++w) cov_xor[w] ^= suffix_word(pos, w);
                }
                // OUTLINE: the run of this edge on row y (i = |y - sy| steps along the minor / major axis)
                const int i = (y - sy) * ystep;
                if (i >= 0 && i <= dy) {
                    int lo, hi;
                    if (dy > dx) {         // y-major: x = sx + floor((2*dx*i + dy - 1) / (2*dy))
                        lo = hi = sx + (int)__umulhi((uint32_t)(2 * dx * i + dy - 1), bcast_i((int)E.inv, e));
                    } else if (dy == 0) {  // horizontal edge / single point
                        lo = sx;
                        hi = sx + dx;
                    } else {  // x-major: steps floor((2*dx*(i-1)+dx)/(2*dy)) + 1 .. min(dx, floor((2*dx*i+dx)/(2*dy)))
                        const uint32_t inv = (uint32_t)bcast_i((int)E.inv, e);
                        const int qhi = (int)__umulhi((uint32_t)(2 * dx * i + dx), inv);
                        const int qlo = i == 0 ? -1 : (int)__umulhi((uint32_t)(2 * dx * (i - 1) + dx), inv);
                        lo = sx + qlo + 1;
                        hi = sx + (qhi > dx ? dx : qhi);
                    }
                    const int p0 = lo - ubase, p1 = hi + 1 - ubase;
#pragma unroll
                    for (int w = 0; w < NW; ++w) cov_or[w] |= suffix_word(p0, w) & ~suffix_word(p1, w);
                }
            }
        }
        if (ESPLIT == 2) {
            // partner = the wave with the other edge slot of the same chunk slot (wave ^ 1)
            const LdsU32 mine = xch + ((round & 1) * 4 + wave) * (NW * 64) + lane;
            const LdsU32 theirs = xch + ((round & 1) * 4 + (wave ^ 1)) * (NW * 64) + lane;
#pragma unroll
            for (int w = 0; w < NW; ++w) mine[w * 64] = cov_xor[w];
            __syncthreads();
#pragma unroll
            for (int w = 0; w < NW; ++w) cov_xor[w] ^= theirs[w * 64];
        }
#pragma unroll
        for (int w = 0; w < NW; ++w) cov_or[w] |= cov_xor[w];
        if (matters && sink.rows(y, valid, cov_or, ubase)) {
            if (ESPLIT == 1) return true;
            stop = true;  // keep taking part in the remaining barriers
        }
    }
    return stop;
}

// Row sink testing coverage against the lethal bitmap (pose_collides, env.py:464-489)
// TILED: `words` is the map's copy in tiles of 32 x 32 cells (pack_bitmap_kernel: word (r, w) of the row-major mask at
// ((r >> 5) * wpr + w) * 32 + (r & 31)) -- the 64 rows a wave reads of one word column are two or three runs of 128 bytes,
// where the row-major rows of a private map each sit in a sector of their own.
template <int NW, typename WordPtr, bool TILED = false>
struct CoopCollisionSink {
    WordPtr words;
    int n_rows, n_cols, wpr, px, py;
    int c_lo, c_hi;        // map columns of the image: px + umin .. px + umax
    uint32_t leth[NW];     // this lane's row: lethal bits of map columns c_lo .. c_lo + 32*NW - 1 (bit 0 <-> c_lo)
    __device__ __forceinline__ void extent(int umin, int umax)
    {
        c_lo = px + umin;
        c_hi = px + umax;
    }
    // The lane's row of one chunk, as stored: the words that hold map columns c_lo .. c_hi (the rest zero).  In two halves, so
    // that a caller can put work between the loads and their use (coop_collides_sparse: the edge parameters).
    __device__ __forceinline__ void load_row(int y, bool valid, uint32_t raw[NW + 1]) const
    {
        const int r = py + y;
        const int w0 = c_lo >> 5;  // arithmetic shift: floor
        const int sh = c_lo & 31;
        const bool row_ok = valid && (unsigned)r < (unsigned)n_rows;
        const int width = c_hi - c_lo + 1;  // columns beyond the image never matter
        const int last_w = (sh + width - 1) >> 5;   // (... nor do the words that only hold such columns)
#pragma unroll
        for (int w = 0; w <= NW; ++w) {
            const int wi = w0 + w;
            const int at = TILED ? (((r >> 5) * wpr + wi) << 5) + (r & 31) : r * wpr + wi;
            raw[w] = (row_ok && w <= last_w && (unsigned)wi < (unsigned)wpr) ? words[at] : 0u;
        }
    }
    // Loads the lane's row once per chunk; returns whether ANY row of the chunk holds a lethal cell under the image.
    __device__ __forceinline__ bool chunk_matters(int y, bool valid)
    {
        uint32_t raw[NW + 1];
        load_row(y, valid, raw);
        return rows_matter(raw);
    }
    __device__ __forceinline__ bool rows_matter(const uint32_t raw[NW + 1])
    {
        const int sh = c_lo & 31;
        const int width = c_hi - c_lo + 1;
        uint32_t any = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            uint32_t v = sh ? (raw[w] >> sh) | (raw[w + 1] << (32 - sh)) : raw[w];
            const int rem = width - 32 * w;
            if (rem <= 0) v = 0;
            else if (rem < 32) v &= 0xFFFFFFFFu >> (32 - rem);
            leth[w] = v;
            any |= v;
        }
        return __any(any != 0);
    }
    __device__ __forceinline__ bool rows(int y, bool valid, const uint32_t cover[NW], int ubase) const
    {
        uint32_t acc = 0;  // bit 0 of cover <-> centred column ubase == umin <-> map column c_lo: same alignment
#pragma unroll
        for (int w = 0; w < NW; ++w) acc |= leth[w] & cover[w];
        return __any(acc != 0);
    }
};

// exact pose_collides for ONE pose, all 64 lanes cooperating; returns a wave-uniform verdict.
// WIDE: the kernel image may be wider than 96 px (8-word row masks instead of 3).  It is a template parameter so that
// a kernel carries only the variant it needs (the rasteriser is the bulk of the code).
template <bool WIDE, typename WordPtr>
__device__ __forceinline__ bool coop_collides(const DevParams& P, double qx, double qy, double c, double s, int px,
                                              int py, WordPtr words, int rows, int cols, int wpr)
{
    constexpr int NW = WIDE ? 8 : 3;
    CoopCollisionSink<NW, WordPtr> sink{words, rows, cols, wpr, px, py, 0, 0};
    return coop_raster<NW, 1>(P, qx, qy, c, s, sink);
}

// runtime-selected width (operator kernels, where code size does not matter)
template <typename WordPtr>
__device__ __forceinline__ bool coop_collides(const DevParams& P, double qx, double qy, double c, double s, int px,
                                              int py, WordPtr words, int rows, int cols, int wpr, bool wide)
{
    return wide ? coop_collides<true>(P, qx, qy, c, s, px, py, words, rows, cols, wpr)
                : coop_collides<false>(P, qx, qy, c, s, px, py, words, rows, cols, wpr);
}

// the same for a workgroup of 4 waves on one pose: wave = 2 * chunk slot + edge slot.  `xch`: 2 * 4 * NW * 64 words
// of LDS.  Returns this wave's partial verdict (OR them over the workgroup).
template <bool WIDE, typename WordPtr>
__device__ __forceinline__ bool coop_collides_quad(const DevParams& P, double qx, double qy, double c, double s, int px,
                                                   int py, WordPtr words, int rows, int cols, int wpr, int wave, LdsU32 xch)
{
    constexpr int NW = WIDE ? 8 : 3;
    CoopCollisionSink<NW, WordPtr> sink{words, rows, cols, wpr, px, py, 0, 0};
    return coop_raster<NW, 2>(P, qx, qy, c, s, sink, wave >> 1, 2, wave & 1, xch, wave);
}

// ---- the exact test, cell by cell ------------------------------------------------------------------------------------
// coop_raster builds the coverage of EVERY row of the footprint image (lane = row, a loop over the edges with ~90
// vector operations per edge and 64-row chunk) and then looks whether a lethal cell lies under it.  On the maps this
// path meets -- walls one cell thick -- the image holds ~1500 cells and the map a handful of lethal ones beneath its
// bounding box (metric workload: median 3, 90th percentile 14), so this form turns the question round:
//   1. the lethal cells under the bounding box are listed (lane = row: the row's lethal bits, room in the list
//      reserved with one LDS atomic, the set bits written out);
//   2. a lane takes a (cell, edge) PAIR -- a group of G = 16 (or 32) lanes per cell, lane e of the group holding edge
//      e's parameters, which it computed itself -- and asks of that one pixel what cv2.fillPoly's raster holds there:
//          outline: the pixel lies on the run of edge e on its row          (same run arithmetic as coop_raster)
//          span   : edge e is active on its row and crosses it to its left  (x > floor(x_e): the parity form, see above)
//      two ballots later every group knows its cell: covered <=> any outline bit, or an odd number of crossings.
// No loop over the edges, no v_readlane: ~60 vector operations per 4 cells.  Same verdict as coop_raster, pixel for
// pixel; `qverts` = the footprint vertices / resolution in LDS (2 doubles each), `list` = kSparseLdsWords words of LDS
// owned by the calling wave.  More than kSparseCap lethal cells under the image (filled obstacles) make the function
// return kSparseTooMany: the caller then rasterises.
constexpr int kSparseCap = 256;
constexpr int kSparseFilterFrom = 12;   // lists longer than this are filtered by the footprint's oriented box first
                                        // (round 3: 24 -> 12, the step ends with its longest tests: 12.47 -> 12.19 us)
constexpr int kSparseLdsWords = kSparseCap + 1;   // the list + its fill counter
enum { kSparseFree = 0, kSparseHit = 1, kSparseTooMany = 2 };

// minimum / maximum over the 16 lanes of a DPP row (every lane gets the result): four rotate-and-combine steps
template <int CTRL>
__device__ __forceinline__ int dpp_ror(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false);   // row_ror:n = 0x120 + n
}

__device__ __forceinline__ int row_min(int v)
{
    v = min(v, dpp_ror<0x128>(v));
    v = min(v, dpp_ror<0x124>(v));
    v = min(v, dpp_ror<0x122>(v));
    return min(v, dpp_ror<0x121>(v));
}

__device__ __forceinline__ int row_max(int v)
{
    v = max(v, dpp_ror<0x128>(v));
    v = max(v, dpp_ror<0x124>(v));
    v = max(v, dpp_ror<0x122>(v));
    return max(v, dpp_ror<0x121>(v));
}

// minimum and maximum of TWO 16-bit values at once over the 16 lanes of a DPP row (v_pk_min_i16 / v_pk_max_i16): the extents of the
// footprint's image, u in the low half, v in the high half (|u|, |v| < 2^15: the image is a few hundred pixels across at most)
typedef short CoopS16x2 __attribute__((ext_vector_type(2)));

template <int CTRL>
__device__ __forceinline__ CoopS16x2 dpp_ror_pk(CoopS16x2 x)
{
    return __builtin_bit_cast(CoopS16x2, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, false));
}

__device__ __forceinline__ CoopS16x2 row_min_pk(CoopS16x2 x)
{
    x = __builtin_elementwise_min(x, dpp_ror_pk<0x128>(x));
    x = __builtin_elementwise_min(x, dpp_ror_pk<0x124>(x));
    x = __builtin_elementwise_min(x, dpp_ror_pk<0x122>(x));
    return __builtin_elementwise_min(x, dpp_ror_pk<0x121>(x));
}

__device__ __forceinline__ CoopS16x2 row_max_pk(CoopS16x2 x)
{
    x = __builtin_elementwise_max(x, dpp_ror_pk<0x128>(x));
    x = __builtin_elementwise_max(x, dpp_ror_pk<0x124>(x));
    x = __builtin_elementwise_max(x, dpp_ror_pk<0x122>(x));
    return __builtin_elementwise_max(x, dpp_ror_pk<0x121>(x));
}

// 1 / d for an integer 0 < |d| < 2^15, good to the last bits of a double: the float32 reciprocal (1 ulp) and two Newton steps.
// What the two exact integer divisions of an edge's set-up are made of below: a quotient n / d with |n / d| < 2^32 then comes
// out within 2^-19 of the true value, a non-integral true value is at least 1 / |d| > 2^-15 away from the next integer, and an
// integral one is restored by the small push before the truncation -- the same integers as the / of the rasteriser
// (tests/test_host_logic.py checks the identities over every divisor and dividend the set-up can meet).
__device__ __forceinline__ double coop_exact_rcp(int d)
{
    const double dd = (double)d;
    double r = (double)__builtin_amdgcn_rcpf((float)d);
    r = fma(fma(-dd, r, 1.0), r, r);
    return fma(fma(-dd, r, 1.0), r, r);
}

template <bool WIDE, bool TILED = false, typename WordPtr>
__device__ __forceinline__ int coop_collides_sparse(const DevParams& P, LdsF64 qverts, double c, double s, int px, int py,
                                                    WordPtr words, int rows, int cols, int wpr, LdsU32 list,
                                                    [[maybe_unused]] unsigned long long* phase = nullptr)
{
    constexpr int NW = WIDE ? 8 : 3;
    const int K = P.n_verts;
    const int lane = lane_id();
    const int G = K <= 16 ? 16 : 32;            // lanes per cell; lane e of a group owns edge e = (V[e-1] -> V[e])
    const int gshift = K <= 16 ? 4 : 5;         // (G = 1 << gshift: shifts, not the divisions lane / G and 64 / G compile to)
    const int e = lane & (G - 1), group = lane >> gshift;
    const bool owner = e < K;
    // ---- this lane's edge, from its two vertices (inert lanes take vertex 0 twice: no effect on the extents)
    const int ve = owner ? e : 0, vp_i = owner ? (e == 0 ? K - 1 : e - 1) : 0;
    const double qx = qverts[2 * ve], qy = qverts[2 * ve + 1], pqx = qverts[2 * vp_i], pqy = qverts[2 * vp_i + 1];
    const int u = (int)rint(fma(qy, -s, qx * c)), v = (int)rint(fma(qy, c, qx * s));       // path_tools.py:142-150
    const int up = (int)rint(fma(pqy, -s, pqx * c)), vp = (int)rint(fma(pqy, c, pqx * s));
    // (u and v as one packed pair: two reductions instead of four)
    CoopS16x2 uv;
    uv.x = (short)u;
    uv.y = (short)v;
    CoopS16x2 lo2 = row_min_pk(uv), hi2 = row_max_pk(uv);
    if (G == 32) {   // the group spans two DPP rows
        lo2 = __builtin_elementwise_min(lo2, __builtin_bit_cast(CoopS16x2, __shfl_xor(__builtin_bit_cast(int, lo2), 16)));
        hi2 = __builtin_elementwise_max(hi2, __builtin_bit_cast(CoopS16x2, __shfl_xor(__builtin_bit_cast(int, hi2), 16)));
    }
    lo2 = __builtin_bit_cast(CoopS16x2, bcast_i(__builtin_bit_cast(int, lo2), 0));   // (the same in every group; scalar from here on)
    hi2 = __builtin_bit_cast(CoopS16x2, bcast_i(__builtin_bit_cast(int, hi2), 0));
    const int umin = (int)lo2.x, vmin = (int)lo2.y, umax = (int)hi2.x, vmax = (int)hi2.y;
    // the lethal words of the image's first 64 rows are asked for right away: with the map in global memory (private maps)
    // their round trip -- 1.6 k cycles of a 3.9 k-cycle test on C4 -- then runs under the edge parameters below
    CoopCollisionSink<NW, WordPtr, TILED> sink{words, rows, cols, wpr, px, py, 0, 0};
    sink.extent(umin, umax);
    // (a map in LDS is read where it is used, as before: asked for early its words only sit in registers -- C3 0.8 % slower)
    uint32_t first_rows[NW + 1];
    if constexpr (TILED) sink.load_row(vmin + lane, vmin + lane <= vmax, first_rows);
    // span edge (CollectPolyEdges): active for y0 <= y < y1, x in 16.16 from the end with the smaller y
    const int ddy = v - vp;
    const int y0 = min(v, vp), y1 = owner ? max(v, vp) : y0;
    const int x0fp = (vp < v ? up : u) << 16;
    // ((u - up) * 65536) / ddy, truncated towards zero like the integer division it stands for (coop_exact_rcp)
    int dxfp = 0;
    if (ddy != 0) {
        const double q0 = (double)((u - up) * 65536) * coop_exact_rcp(ddy);
        dxfp = (int)(q0 + copysign(0x1p-12, q0));
    }
    // Bresenham (LineIterator, leftToRight): from the end with the smaller x
    int sx = up, sy = vp, dx = u - up, dy = v - vp;
    if (dx < 0) {
        dx = -dx;
        dy = -dy;
        sx = u;
        sy = v;
    }
    const bool down = dy < 0;   // ystep = -1
    if (down) dy = -dy;
    // floor(2^32 / (2 dy)) + 1 (see coop_raster), without the float64 division
    const uint32_t inv = (uint32_t)(4294967296.0 * coop_exact_rcp(2 * (dy > 0 ? dy : 1)) + 0x1p-17) + 1u;
    if (!owner) dy = -1;        // inert: no run on any row
    // ---- the lethal cells under the image's columns, all row chunks into one list (lane = row of the chunk; the order
    //      of the cells does not matter, so a lane just reserves room for its row's cells with one LDS atomic)
#ifdef BCP_DIAG
    if (phase) phase[0] = __builtin_amdgcn_s_memtime();   // the edges are set up
#endif
    if (lane == 0) list[kSparseCap] = 0;
    wave_lds_sync();
    const int n_chunks = (vmax - vmin) / 64 + 1;
    for (int chunk = 0; chunk < n_chunks; ++chunk) {
        const int y_row = vmin + 64 * chunk + lane;
        // (the lane's row of the chunk; wave-uniform result)
        if (!((TILED && chunk == 0) ? sink.rows_matter(first_rows) : sink.chunk_matters(y_row, y_row <= vmax))) continue;
        int count = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) count += (int)__popc(sink.leth[w]);
        if (count) {
            int at = (int)atomicAdd((unsigned int*)&list[kSparseCap], (unsigned int)count);
            if (at + count <= kSparseCap) {
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    uint32_t bits = sink.leth[w];
                    while (bits) {
                        const int b = (int)__builtin_ctz(bits);
                        bits &= bits - 1;
                        list[at++] = ((uint32_t)(64 * chunk + lane) << 16) | (uint32_t)(32 * w + b);   // (row - vmin, column - umin)
                    }
                }
            }
        }
    }
    wave_lds_sync();
    int total = bcast_i((int)list[kSparseCap], 0);
#ifdef BCP_DIAG
    if (phase) phase[1] = __builtin_amdgcn_s_memtime();   // the lethal cells are listed
    if (phase) phase[2] = (unsigned long long)total;
#endif
    if (total > kSparseCap) return kSparseTooMany;
    // A long list (the median is 3 cells) sets the pace of its whole workgroup -- and the slowest workgroup that of the
    // step: drop the cells that lie outside the footprint's own bounding box (robot frame, 2 px of slack for vertex
    // rounding and Bresenham) first, 64 cells per trip, compacting the list in place.  Short lists skip this: the pass
    // costs more than testing a dozen cells.
    if (total > kSparseFilterFrom) {
        const float cf = (float)c, sf = (float)s;
        int kept = 0;
        for (int base = 0; base < total; base += 64) {
            const bool valid = base + lane < total;
            const uint32_t cell = valid ? list[base + lane] : 0u;
            const float x = (float)(umin + (int)(cell & 0xFFFFu)), y = (float)(vmin + (int)(cell >> 16));
            const float xr = x * cf + y * sf, yr = y * cf - x * sf;
            const bool keep = valid && xr >= P.qbox[0] - 2.0f && xr <= P.qbox[1] + 2.0f && yr >= P.qbox[2] - 2.0f &&
                              yr <= P.qbox[3] + 2.0f;
            const uint64_t keeps = __ballot(keep);
            wave_lds_sync();   // (every lane has read its cell before anybody overwrites the slots below it)
            if (keep) list[kept + (int)__popcll(keeps & ((1ull << lane) - 1ull))] = cell;
            kept += (int)__popcll(keeps);
        }
        wave_lds_sync();
        total = kept;
    }
    // ---- a group of lanes per cell, a lane per edge
    const int per_pass = 64 >> gshift;
    const uint64_t group_mask = (G == 16 ? 0xFFFFull : 0xFFFFFFFFull) << (group << gshift);
    for (int base = 0; base < total; base += per_pass) {
        const bool valid = base + group < total;
        const uint32_t cell = valid ? list[base + group] : 0u;
        const int y = vmin + (int)(cell >> 16), x = umin + (int)(cell & 0xFFFFu);
        const bool crossing = y >= y0 && y < y1 && x > ((x0fp + (y - y0) * dxfp) >> 16);
        bool on_outline = false;
        const int i = down ? sy - y : y - sy;
        if (i >= 0 && i <= dy) {     // the run of this edge on the pixel's row
            int lo, hi;
            if (dy > dx) {           // y-major: x = sx + floor((2*dx*i + dy - 1) / (2*dy))
                lo = hi = sx + (int)__umulhi((uint32_t)(2 * dx * i + dy - 1), inv);
            } else if (dy == 0) {    // horizontal edge / single point
                lo = sx;
                hi = sx + dx;
            } else {                 // x-major: steps floor((2*dx*(i-1)+dx)/(2*dy)) + 1 .. min(dx, floor((2*dx*i+dx)/(2*dy)))
                const int qhi = (int)__umulhi((uint32_t)(2 * dx * i + dx), inv);
                const int qlo = i == 0 ? -1 : (int)__umulhi((uint32_t)(2 * dx * (i - 1) + dx), inv);
                lo = sx + qlo + 1;
                hi = sx + (qhi > dx ? dx : qhi);
            }
            on_outline = x >= lo && x <= hi;
        }
        const uint64_t outlines = __ballot(valid && on_outline), crossings = __ballot(valid && crossing);
        const bool covered = (outlines & group_mask) != 0 || (__popcll(crossings & group_mask) & 1);
        if (__any(covered)) return kSparseHit;
    }
    return kSparseFree;
}

}  // namespace bcp
