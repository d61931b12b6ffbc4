// bcp_ego.h -- device code of the egocentric observation: extract_egocentric_costmap
// (utilities/costmap_utils.py:25-75 = cv2.getRotationMatrix2D + cv2.warpAffine with INTER_NEAREST) for every env at
// once, plus the kernels that group images by map entry for private / pooled maps.  Included by bcplan.hip, which holds
// the host entry points (bcp_egocentric_costmaps, bcp_goal_n_state).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include "bcp_coop.h"   // bcast_i / bcast_d

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "bcp_ego.h is written for gfx950 (MI355X): lds_byte_shifted_16 relies on d16 loads writing the whole register (SRAM-ECC targets), and the sampling kernels on 160 KB of LDS per workgroup"
#endif

namespace bcp {

extern __shared__ __attribute__((aligned(16))) uint32_t ego_lds[];

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
    int32_t win_lds_bytes;       // ego_costmap_window_kernel: LDS bytes for the visible part of the map (+ ring)
    uint8_t* out;                // [n][drows][dcols]
};

__device__ __forceinline__ int sat_int(double v)   // cv::saturate_cast<int>(double): nearest-even, saturating
{
    const double r = rint(v);
    return r >= 2147483647.0 ? 2147483647 : (r <= -2147483648.0 ? (-2147483647 - 1) : (int)r);
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

// A byte of LDS delivered as (byte << 16) (ds_read_u8_d16_hi; no compiler builtin).  Relies on what SRAM-ECC targets
// (gfx90a / gfx942 / gfx950) do with d16 loads: the whole register is written, the other half as zero -- the
// bit-exact image tests are the check.  The compiler does not see the read as a
// memory operation: the consumer calls lds_reads_done() (s_waitcnt lgkmcnt(0)) before it touches the result.
__device__ __forceinline__ uint32_t lds_byte_shifted_16(uint32_t addr)
{
    uint32_t v = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("ds_read_u8_d16_hi %0, %1" : "=v"(v) : "v"(addr));   // (gfx950: the low half is written as zero)
#else
    (void)addr;
#endif
    return v;
}

__device__ __forceinline__ void lds_reads_done()
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
}

typedef uint64_t __attribute__((aligned(1))) u64_unaligned;
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;

// ---- building blocks ------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) int* LdsI32;
typedef __attribute__((address_space(3))) uint8_t* LdsU8;
constexpr int kRowOff = -2147483647 - 1;   // row-table X0 of a row that lies off the map (real X0 are >= INT_MIN + 512)
constexpr int kRowOutShift = 1024;              // an off-map row is parked this many cells left of the ring (> any dcols)

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
// The map is fetched as aligned dwords, sixteen independent loads in flight per thread -- a cold map (pool entries come
// from HBM) then costs two or three memory round trips, the first of them hidden behind the border fill -- and is
// scattered into the ringed layout byte by byte.
__device__ __forceinline__ void ego_stage_map(const EgoArgs& a, const uint8_t* __restrict__ src, int vr, int vc, LdsU8 lmap,
                                              int pitch, int map_bytes, int threads = 256)
{
    constexpr int kInFlight = 16;
    const int total = a.rows * a.cols;
    const int off = (int)((uintptr_t)src & 3);   // the map entry need not start on a dword boundary
    const uint32_t* __restrict__ w32 = reinterpret_cast<const uint32_t*>(src - off);
    const int n_words = (off + total + 3) >> 2;
    uint32_t v[kInFlight];
#pragma unroll
    for (int u = 0; u < kInFlight; ++u) {   // first batch: issued before the fill
        const int w = threadIdx.x + u * threads;
        v[u] = w < n_words ? w32[w] : 0u;
    }
    __attribute__((address_space(3))) uint32_t* l32 = (__attribute__((address_space(3))) uint32_t*)lmap;
    for (int k = threadIdx.x; k < map_bytes / 4; k += threads) l32[k] = (uint32_t)a.border * 0x01010101u;
    __syncthreads();
    for (int w0 = threadIdx.x; w0 < n_words; w0 += kInFlight * threads) {
        if (w0 != (int)threadIdx.x) {
#pragma unroll
            for (int u = 0; u < kInFlight; ++u) {
                const int w = w0 + u * threads;
                v[u] = w < n_words ? w32[w] : 0u;
            }
        }
#pragma unroll
        for (int u = 0; u < kInFlight; ++u) {
            const int w = w0 + u * threads;
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

// How source coordinates map onto the LDS copy of (a part of) the costmap: the copy holds map columns c0 .. c0 + w - 1
// and rows r0 .. r0 + h - 1 inside a one-cell ring of the border value, `pitch` = w + 2 bytes per row.
//   X' = X + x_add, clamped to [0, x_hi]   (x_add = 1 - c0)
//   Y' = Y + y_add, clamped to [0, y_hi]   (y_add = 1 - r0)          byte address = base + Y' * pitch + X'
struct EgoStage {
    int x_add, y_add, x_hi, y_hi, pitch, base;
};

__device__ __forceinline__ EgoStage ego_stage_of(int lds_base, int c0, int r0, int w, int h)
{
    EgoStage G;
    G.x_add = 1 - c0;
    G.y_add = 1 - r0;
    G.x_hi = w + 1;
    G.y_hi = h + 1;
    G.pitch = w + 2;
    G.base = lds_base;
    return G;
}

// per-row terms of one image (rounding term included) for rows t0, t0 + tstep, ..., as a table [drows][2]; kRowOff
// marks a row that lies off the map in the plain (22.10) table of the global-memory path.  Staged sampling works in
// 16.16: its table holds (term + ring offset) << 6, so that the integer part of a source coordinate -- (row term +
// column term) >> 10 in cv::warpAffine's 22.10 -- is the upper half of the sum.  A row that lies off the map gets terms
// that put every pixel of it on the ring's left column (it is only sampled when it shares a bundle of rows with a live
// one).  A row that is not off the map stays within (window width) cells of it, so the shifted terms fit as long as the
// map is narrower than 2^15 - 2 dcols cells (LDS-resident maps are).
constexpr int kEgoBoundInts = 16;   // behind the tables: {first live row, first inside row, inside end, live end} per wave

// ints of one table: row terms, row bounds
__device__ __forceinline__ int ego_table_ints(const EgoArgs& a) { return 2 * a.drows + kEgoBoundInts; }

template <bool STAGED>
__device__ __forceinline__ void ego_row_terms(const EgoArgs& a, const EgoImage& I, const EgoStage& G, LdsI32 row_tab,
                                              int t0, int tstep)
{
    const int last_cx = sat_int(I.m0 * (a.dcols - 1) * 1024), last_cy = sat_int(I.m3 * (a.dcols - 1) * 1024);
    // Rows that are not off the map form an interval of the image, and so do the rows entirely inside it (the source
    // coordinates of both row ends are monotone in y): [live_lo, live_hi) and [in_lo, in_hi), found with ballots.
    const int lane = t0 & 63;
    int live_lo = a.drows, live_hi = 0, in_lo = a.drows, in_hi = 0;
    for (int y0 = t0 - lane; y0 < a.drows; y0 += tstep) {   // (wave-uniform: this pass covers rows y0 .. y0 + 63)
        const int y = y0 + lane;
        const bool real = y < a.drows;
        const int rx = sat_int((I.m1 * y + I.m2) * 1024) + 512, ry = sat_int((I.m4 * y + I.m5) * 1024) + 512;
        const int xa = rx >> 10, xb = (rx + last_cx) >> 10, ya = ry >> 10, yb = (ry + last_cy) >> 10;
        const bool off = (xa < 0 && xb < 0) || (xa >= I.vc && xb >= I.vc) || (ya < 0 && yb < 0) || (ya >= I.vr && yb >= I.vr);
        if (STAGED) {
            // (monotone along a row too: both ends inside => every pixel inside)
            const bool inside = (unsigned)xa < (unsigned)I.vc && (unsigned)xb < (unsigned)I.vc &&
                                (unsigned)ya < (unsigned)I.vr && (unsigned)yb < (unsigned)I.vr;
            const uint64_t live = __ballot(real && !off), in = __ballot(real && inside);
            if (live) {
                live_lo = min(live_lo, y0 + (int)__builtin_ctzll(live));
                live_hi = max(live_hi, y0 + 64 - (int)__builtin_clzll(live));
            }
            if (in) {
                in_lo = min(in_lo, y0 + (int)__builtin_ctzll(in));
                in_hi = max(in_hi, y0 + 64 - (int)__builtin_clzll(in));
            }
            if (real) {
                row_tab[2 * y] = off ? (int)((uint32_t)(-kRowOutShift * 1024) << 6) : (int)((uint32_t)(rx + G.x_add * 1024) << 6);
                row_tab[2 * y + 1] = off ? 0 : (int)((uint32_t)(ry + G.y_add * 1024) << 6);
            }
        } else if (real) {
            row_tab[2 * y] = off ? kRowOff : rx;
            row_tab[2 * y + 1] = ry;
        }
    }
    if (STAGED && lane == 0) {
        const LdsI32 bounds = row_tab + 2 * a.drows + 4 * ((t0 >> 6) & 3);
        bounds[0] = live_lo;
        bounds[1] = in_lo;
        bounds[2] = in_hi;
        bounds[3] = live_hi;
    }
}

typedef short EgoI16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short EgoU16x2 __attribute__((ext_vector_type(2)));
typedef int EgoI32x2 __attribute__((ext_vector_type(2)));

// LDS reads whose latency the pixel loop hides itself: issued here, consumed only after lds_wait_*() -- the compiler does
// not know they are in flight (its own waits would count them and drain the queue early), so every LDS access between
// the first issue and the wait goes through these.
__device__ __forceinline__ uint32_t lds_issue_u8(uint32_t addr)
{
    uint32_t v = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("ds_read_u8 %0, %1" : "=v"(v) : "v"(addr));
#else
    (void)addr;
#endif
    return v;
}

__device__ __forceinline__ EgoI32x2 lds_issue_b64(uint32_t addr)
{
    EgoI32x2 v = {0, 0};
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr));
#else
    (void)addr;
#endif
    return v;
}

template <int PX>
__device__ __forceinline__ void lds_wait_pixels(uint32_t (&val)[PX], EgoI32x2& row)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (PX == 8)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(val[0]), "+v"(val[1]), "+v"(val[2]), "+v"(val[3]), "+v"(val[4]), "+v"(val[5]), "+v"(val[6]),
                       "+v"(val[7]), "+v"(row)
                     :
                     : "memory");
    else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(val[0]), "+v"(val[1]), "+v"(val[2]), "+v"(val[3]), "+v"(row) : : "memory");
#endif
}

// the pixels of rows r_first + rl, + rstep, ... for this lane's column groups (rl = the lane's row within the wave's
// bundle of kRows rows).  Pixel groups are PX columns wide; the last group of a row is shifted left so that it ends
// at the last column (it recomputes a few pixels of its neighbour instead of needing a narrower store); a lane whose
// row would lie below the image repeats the last row (same bytes to the same addresses) -- no execution masks.
// Staged sampling (the map copy in LDS), per pixel: two adds (row term + column term, 16.16), one v_perm_b32 that puts
// the two integer parts side by side as {Y, X} 16-bit halves, one v_dot2_u32_u16 with {pitch, 1} that turns them into
// the LDS byte address, the byte read -- and, only for bundles with a row that can leave the map, v_pk_max_i16 /
// v_pk_min_i16 in between, which clamp both coordinates onto the border ring of the LDS copy at once.  Which bundles
// those are follows from the row bounds of ego_row_terms: scalar trip counts, no per-trip test.  Bundles off the map
// are filled with the border value without sampling.  The loop is software-pipelined by hand: the byte reads of
// bundle k are in flight while the addresses of bundle k + 1 are computed; the row terms are fetched two bundles ahead.
template <int PX>
__device__ __forceinline__ void ego_pixels_lds(const EgoArgs& a, const EgoImage& I, const EgoStage& G, LdsI32 row_tab,
                                               uint8_t* __restrict__ image, int cg, int rl, int r_first, int rstep, int parts)
{
    constexpr int kRows = 64 / (128 / PX);
    const uint64_t border8 = (uint64_t)(uint32_t)a.border * 0x0101010101010101ull;
    const EgoI16x2 ring_lo = {0, 0}, ring_hi = {(short)G.x_hi, (short)G.y_hi};
    const EgoU16x2 weights = {1, (unsigned short)G.pitch};
    const uint32_t rows_at = (uint32_t)(uintptr_t)row_tab;
    // row bounds -> bundle bounds [0, it_a) off | [it_a, it_b) clamped | [it_b, it_c) inside | [it_c, it_d) clamped | off
    int live_lo = a.drows, in_lo = a.drows, in_hi = 0, live_hi = 0;
    {
        const LdsI32 bounds = row_tab + 2 * a.drows;
        for (int p = 0; p < parts; ++p) {
            live_lo = min(live_lo, __builtin_amdgcn_readfirstlane(bounds[4 * p + 0]));
            in_lo = min(in_lo, __builtin_amdgcn_readfirstlane(bounds[4 * p + 1]));
            in_hi = max(in_hi, __builtin_amdgcn_readfirstlane(bounds[4 * p + 2]));
            live_hi = max(live_hi, __builtin_amdgcn_readfirstlane(bounds[4 * p + 3]));
        }
    }
    const auto trips_below = [rstep](int rows) { return rows <= 0 ? 0 : (rows + rstep - 1) / rstep; };   // #k: k * rstep < rows
    const int n_it = trips_below(a.drows - r_first);
    const int it_a = min(trips_below(live_lo - (kRows - 1) - r_first), n_it);
    const int it_d = max(it_a, min(trips_below(live_hi - r_first), n_it));
    const int it_b = max(it_a, min(trips_below(in_lo - r_first), it_d));
    const int it_c = max(it_b, min(in_hi - kRows - r_first < 0 ? 0 : (in_hi - kRows - r_first) / rstep + 1, it_d));
    const int last_row = a.drows - 1;
    for (int xg = PX * cg; xg < a.dcols; xg += 128) {
        const int x0 = min(xg, a.dcols - PX);
        const auto store = [&](int k, uint64_t packed) {
            uint8_t* const p = image + (uint32_t)(__mul24(min(r_first + k * rstep + rl, last_row), a.dcols) + x0);
            if (PX == 8)
                *reinterpret_cast<u64_unaligned*>(p) = packed;
            else
                *reinterpret_cast<u32_unaligned*>(p) = (uint32_t)packed;
        };
        for (int k = 0; k < it_a; ++k) store(k, border8);
        for (int k = it_d; k < n_it; ++k) store(k, border8);
        if (it_a == it_d) continue;
        EgoI32x2 cc[PX];
#pragma unroll
        for (int j = 0; j < PX; ++j) {   // this lane's column terms (cv::hal::warpAffine's adelta / bdelta), 16.16
            // (|M0 x 1024| < 2^31 by a wide margin -- the transform is a rotation -- so saturate_cast<int> is plain rounding)
            cc[j].x = (int)((uint32_t)(int)rint(I.m0 * (x0 + j) * 1024) << 6);
            cc[j].y = (int)((uint32_t)(int)rint(I.m3 * (x0 + j) * 1024) << 6);
        }
        const auto row_at = [&](int k) { return rows_at + 8u * (uint32_t)min(r_first + k * rstep + rl, last_row); };
        const auto addresses = [&](const EgoI32x2 row, bool clamp, uint32_t (&addr)[PX]) {
            uint32_t yx[PX];
#pragma unroll
            for (int j = 0; j < PX; ++j)
                yx[j] = __builtin_amdgcn_perm((uint32_t)(row.y + cc[j].y), (uint32_t)(row.x + cc[j].x), 0x07060302u);
            if (clamp) {
#pragma unroll
                for (int j = 0; j < PX; ++j)
                    yx[j] = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(
                        __builtin_elementwise_max(__builtin_bit_cast(EgoI16x2, yx[j]), ring_lo), ring_hi));
            }
            // (saturate_cast<short> never bites: |X|, |Y| < 2^15 on live rows, and off-map rows are parked)
#pragma unroll
            for (int j = 0; j < PX; ++j)
                addr[j] = __builtin_amdgcn_udot2(__builtin_bit_cast(EgoU16x2, yx[j]), weights, (uint32_t)G.base, false);
        };
        // prologue: row terms of bundles it_a and it_a + 1, addresses of bundle it_a
        EgoI32x2 row = lds_issue_b64(row_at(it_a)), row_next = lds_issue_b64(row_at(min(it_a + 1, it_d - 1)));
        uint32_t addr[PX], val[PX];
#pragma unroll
        for (int j = 0; j < PX; ++j) val[j] = 0;
        lds_wait_pixels<PX>(val, row);
        lds_wait_pixels<PX>(val, row_next);
        addresses(row, it_a < it_b || it_a >= it_c, addr);
        for (int k = it_a; k < it_d; ++k) {
            EgoI32x2 row_after = lds_issue_b64(row_at(min(k + 2, it_d - 1)));
            // Packing four pixels takes two operations instead of four: the third byte arrives already shifted
            // (ds_read_u8_d16_hi puts it into bits 16..23; on this ECC target the low half comes back zero), one
            // v_perm_b32 places bytes one and three, one v_or3_b32 joins the three registers.
#pragma unroll
            for (int j = 0; j < PX; ++j) val[j] = (j & 3) == 2 ? lds_byte_shifted_16(addr[j]) : lds_issue_u8(addr[j]);
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : "+v"(row_next));   // (the next bundle's arithmetic stays behind the issue of these reads)
#endif
            uint32_t addr_next[PX];
            addresses(row_next, k + 1 < it_b || k + 1 >= it_c, addr_next);
            lds_wait_pixels<PX>(val, row_after);
            uint32_t half[2] = {0, 0};
#pragma unroll
            for (int q = 0; q < PX / 4; ++q)   // (b3 << 24) | (b1 << 8), then | b0 | (b2 << 16)
                half[q] = __builtin_amdgcn_perm(val[4 * q + 3], val[4 * q + 1], 0x040c000cu) | val[4 * q] | val[4 * q + 2];
            store(k, ((uint64_t)half[1] << 32) | half[0]);
            row_next = row_after;
#pragma unroll
            for (int j = 0; j < PX; ++j) addr[j] = addr_next[j];
        }
    }
}

// the same from global memory (maps that do not fit LDS and whose window does not either): plain 22.10 terms, a bounds
// test per pixel, only the in-map lanes issue a load
template <int PX>
__device__ __forceinline__ void ego_pixels_global(const EgoArgs& a, const EgoImage& I, const uint8_t* __restrict__ src,
                                                  LdsI32 row_tab, uint8_t* __restrict__ image, int cg, int r0, int rstep)
{
    const uint32_t border = (uint32_t)a.border;
    const uint64_t border8 = (uint64_t)border * 0x0101010101010101ull;
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
                    const int X = (rx + ccx[j]) >> 10, Y = (ry + ccy[j]) >> 10;
                    uint32_t val = border;
                    if ((unsigned)X < (unsigned)I.vc && (unsigned)Y < (unsigned)I.vr)
                        val = (uint32_t)src[(uint32_t)__mul24(Y, a.cols) + (uint32_t)X];
                    half[j >> 2] |= val << (8 * (j & 3));
                }
                packed = ((uint64_t)half[1] << 32) | half[0];
            }
            uint8_t* const p = image + (uint32_t)(__mul24(y, a.dcols) + x0);
            if (PX == 8)
                *reinterpret_cast<u64_unaligned*>(p) = packed;
            else
                *reinterpret_cast<u32_unaligned*>(p) = (uint32_t)packed;
        }
    }
}

template <bool STAGED, int PX>
__device__ __forceinline__ void ego_pixels(const EgoArgs& a, const EgoImage& I, const EgoStage& G,
                                           const uint8_t* __restrict__ src, LdsI32 row_tab, uint8_t* __restrict__ image,
                                           int cg, int rl, int r_first, int rstep, int parts)
{
    if (STAGED) ego_pixels_lds<PX>(a, I, G, row_tab, image, cg, rl, r_first, rstep, parts);
    else ego_pixels_global<PX>(a, I, src, row_tab, image, cg, r_first + rl, rstep);
}


constexpr int kEgoWaves = 8;   // wavefronts (= images in flight) per workgroup of ego_costmap_kernel

// One WAVEFRONT per image, persistent workgroups of kEgoWaves waves that stage a shared costmap in LDS once and then
// walk over images.
//   * transforms: lane l of a wave prepares the (inverted) warp matrix of the wave's l-th image, so the float64
//     sin / cos / inversion work is done once per image by one lane; the wave then takes the images one by one and
//     broadcasts that lane's matrix (v_readlane -> scalar registers).
//   * pixels: lane = (row mod 4, group of 8 consecutive columns).  The column terms of a lane's 8 pixels stay in
//     registers, the per-row terms come from a small per-wave LDS table, and the 8 pixels leave as one 64-bit store
//     (image rows are dcols bytes apart, so these stores are generally unaligned).
//   * rows whose source segment lies entirely off the map are filled with the border value without sampling: the
//     source coordinates are monotone in x, so it is enough to look at the row's two ends; such rows, the rows
//     entirely inside the map and the ones in between form intervals of the image (ego_row_terms finds their bounds).
//   * shared map: the LDS copy carries a one-cell ring of the border value and the source coordinates are clamped
//     onto it where a row can leave the map, so a pixel is add, add, v_perm (both integer parts into one register),
//     [v_pk_max, v_pk_min,] v_dot2 (address), LDS byte read, pack -- no bounds compare and no select (ego_pixels).
//     The ring offset rides in the per-row terms.
// LDS: [shared map + ring, padded to 8 bytes] [kEgoWaves x ((drows x {X0, Y0}) + row bounds)]
// STAGED = false: maps that do not fit LDS are sampled straight from global memory.
template <bool STAGED, int PX>
__global__ void __launch_bounds__(64 * kEgoWaves) ego_costmap_kernel(const EgoArgs a)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const int pitch = a.cols + 2;
    const int map_bytes = STAGED ? (((a.rows + 2) * pitch + 7) & ~7) : 0;
    const LdsU8 lmap = (LdsU8)ego_lds;
    const LdsI32 row_tab = (LdsI32)(lmap + map_bytes) + wave * ego_table_ints(a);
    if (STAGED)
        ego_stage_map(a, a.data, a.valid_rows ? a.valid_rows[0] : a.rows, a.valid_cols ? a.valid_cols[0] : a.cols, lmap,
                      pitch, map_bytes, blockDim.x);
    const int lds_base = (int)(uint32_t)(uintptr_t)lmap;   // (source coordinates become raw LDS byte addresses)
    constexpr int kGroups = 128 / PX, kRows = 64 / kGroups;   // lanes of a wave: kRows image rows x kGroups pixel groups
    const int cg = lane % kGroups, rl = lane / kGroups;
    const int64_t P = (int64_t)a.drows * a.dcols;
    const int64_t first = (int64_t)blockIdx.x * waves + wave, stride = (int64_t)gridDim.x * waves;
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
            EgoStage G = ego_stage_of(lds_base, 0, 0, I.vc, I.vr);
            G.pitch = pitch;
            ego_row_terms<STAGED>(a, I, G, row_tab, lane, 64);
            wave_lds_sync();
            ego_pixels<STAGED, PX>(a, I, G, a.data + I.g * a.map_stride, row_tab, a.out + img * P, cg, rl, 0, kRows, 1);
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
    const int map_bytes = ((a.rows + 2) * pitch + 7) & ~7;   // (the tables behind it are read 8 bytes at a time: a misaligned ds_read_b64 takes 33 LDS cycles instead of 2)
    const LdsU8 lmap = (LdsU8)ego_lds;
    const LdsI32 tables = (LdsI32)(lmap + map_bytes);
    const LdsI32 wave_tab = tables + wave * ego_table_ints(a);
    const int lds_base = (int)(uint32_t)(uintptr_t)lmap;
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
                EgoStage G = ego_stage_of(lds_base, 0, 0, I.vc, I.vr);
                G.pitch = pitch;
                ego_row_terms<true>(a, I, G, wave_tab, lane, 64);
                wave_lds_sync();
                ego_pixels<true, PX>(a, I, G, nullptr, wave_tab, a.out + img * P, cg, rl, 0, kRows, 1);
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
                EgoStage G = ego_stage_of(lds_base, 0, 0, I.vc, I.vr);
                G.pitch = pitch;
                ego_row_terms<true>(a, I, G, tables, threadIdx.x, 256);
                __syncthreads();
                ego_pixels<true, PX>(a, I, G, nullptr, tables, a.out + img * P, cg, rl, wave * kRows, 4 * kRows, 4);
            }
        }
        pos = run_end;
    }
}

// Maps too large for LDS (a 333 x 183 AisleTurn costmap with its ring already is): one workgroup per image stages
// only the part of the map the window can see -- the bounding box of the window's four corners in source coordinates
// (exact: the source coordinates are monotone along rows and columns), at most about diag(window)^2 bytes -- inside
// the usual border ring, and the 4 waves then share the image like the left-over path above.
// LDS: [window part of the map + ring (a.win_lds_bytes)] [one row table]
template <int PX>
__global__ void __launch_bounds__(256) ego_costmap_window_kernel(const EgoArgs a)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const LdsU8 lmap = (LdsU8)ego_lds;
    const LdsI32 tables = (LdsI32)(lmap + a.win_lds_bytes);
    const int lds_base = (int)(uint32_t)(uintptr_t)lmap;
    constexpr int kGroups = 128 / PX, kRows = 64 / kGroups;
    const int cg = lane % kGroups, rl = lane / kGroups;
    const int64_t P = (int64_t)a.drows * a.dcols;
    const int64_t first = blockIdx.x, stride = gridDim.x;
    for (int64_t base = first; base < a.n_images; base += 64 * stride) {
        EgoXform T;
        memset(&T, 0, sizeof(T));
        const int64_t mine = base + lane * stride;   // lane l: transform of the workgroup's l-th image of this batch
        if (mine < a.n_images) T = ego_transform(a, mine);
        const int64_t left = (a.n_images - base + stride - 1) / stride;
        const int count = (int)(left < 64 ? left : 64);
        for (int k = 0; k < count; ++k) {
            const int64_t img = base + k * stride;
            const EgoImage I = ego_broadcast(T, k);
            // source bounding box of the window: its corners (first / last column terms, first / last row terms)
            const int cx1 = sat_int(I.m0 * (a.dcols - 1) * 1024), cy1 = sat_int(I.m3 * (a.dcols - 1) * 1024);
            const int rx0 = sat_int(I.m2 * 1024) + 512, ry0 = sat_int(I.m5 * 1024) + 512;
            const int rx1 = sat_int((I.m1 * (a.drows - 1) + I.m2) * 1024) + 512;
            const int ry1 = sat_int((I.m4 * (a.drows - 1) + I.m5) * 1024) + 512;
            const int xs[4] = {rx0 >> 10, (rx0 + cx1) >> 10, rx1 >> 10, (rx1 + cx1) >> 10};
            const int ys[4] = {ry0 >> 10, (ry0 + cy1) >> 10, ry1 >> 10, (ry1 + cy1) >> 10};
            int c0 = max(min(min(xs[0], xs[1]), min(xs[2], xs[3])), 0);
            int c1 = min(max(max(xs[0], xs[1]), max(xs[2], xs[3])), I.vc - 1);
            int r0 = max(min(min(ys[0], ys[1]), min(ys[2], ys[3])), 0);
            int r1 = min(max(max(ys[0], ys[1]), max(ys[2], ys[3])), I.vr - 1);
            int w = c1 - c0 + 1, h = r1 - r0 + 1;
            if (w <= 0 || h <= 0) w = h = 0;   // the window misses the map: every row is flagged off the map below
            const int pitch = w + 2;
            __syncthreads();   // the previous image is finished with the LDS copy and the table
            if ((h + 2) * pitch <= a.win_lds_bytes) {
                // ring first (top / bottom rows, left / right columns) ...
                for (int c = threadIdx.x; c < pitch; c += 256) {
                    lmap[c] = (uint8_t)a.border;
                    lmap[(h + 1) * pitch + c] = (uint8_t)a.border;
                }
                for (int r = threadIdx.x; r < h; r += 256) {
                    lmap[(r + 1) * pitch] = (uint8_t)a.border;
                    lmap[(r + 1) * pitch + w + 1] = (uint8_t)a.border;
                }
                // ... then the visible rows: a wave takes rows wave, wave + 4, ...; a row is fetched as aligned dwords
                // (one per lane: w + 3 <= 256 bytes), eight rows in flight per wave, and scattered byte-wise into LDS
                const uint8_t* const origin = a.data + I.g * a.map_stride + (int64_t)r0 * a.cols + c0;
                constexpr int kInFlight = 8;   // rows in flight per wave
                for (int rb = wave; rb < h; rb += 4 * kInFlight) {
                    uint32_t word[kInFlight];
                    int skew[kInFlight];
#pragma unroll
                    for (int u = 0; u < kInFlight; ++u) {
                        const int r = rb + 4 * u;
                        const uint8_t* row = origin + (int64_t)r * a.cols;
                        skew[u] = (int)((uintptr_t)row & 3);
                        const uint32_t* aligned = reinterpret_cast<const uint32_t*>(row - skew[u]);
                        word[u] = (r < h && 4 * lane < w + skew[u]) ? aligned[lane] : 0u;
                    }
#pragma unroll
                    for (int u = 0; u < kInFlight; ++u) {
                        const int r = rb + 4 * u;
                        if (r < h) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int c = 4 * lane + j - skew[u];
                                if ((unsigned)c < (unsigned)w) lmap[(r + 1) * pitch + 1 + c] = (uint8_t)(word[u] >> (8 * j));
                            }
                        }
                    }
                }
                const EgoStage G = ego_stage_of(lds_base, c0, r0, w, h);
                ego_row_terms<true>(a, I, G, tables, threadIdx.x, 256);
                __syncthreads();
                ego_pixels<true, PX>(a, I, G, nullptr, tables, a.out + img * P, cg, rl, wave * kRows, 4 * kRows, 4);
            } else {
                // (cannot happen for windows the host accepted; kept as a safe fallback: sample from global memory)
                EgoStage G = ego_stage_of(0, 0, 0, I.vc, I.vr);
                ego_row_terms<false>(a, I, G, tables, threadIdx.x, 256);
                __syncthreads();
                ego_pixels<false, PX>(a, I, G, a.data + I.g * a.map_stride, tables, a.out + img * P, cg, rl, wave * kRows,
                                      4 * kRows, 4);
            }
        }
    }
}

// ---- sparse costmaps: fill + patch ---------------------------------------------------------------------------------------
// The costmaps this path meets are nearly empty (RandomMiniEnv: two walls one cell thick, ~120 cells of 33 489 non-zero; the
// AisleTurn maps of the reference's PPO runner: 543 of 60 939 and 1 040 of 179 200) and the border value of
// extract_egocentric_costmap is 0 (costmap_utils.py:27): an egocentric image is then ZEROS plus the few pixels whose source
// cell is not.  INTER_NEAREST is an exact inverse map, so instead of asking every destination pixel for its source cell
// (15 561 fixed-point evaluations and LDS reads per image, issue-bound at 46 % of the HBM write rate)
//   1. the image is filled with zeros in 16-byte stores -- a plain memset, all the HBM traffic there is;
//   2. every non-zero source cell (X, Y) is mapped FORWARD to the image with the inverse of the dst -> src matrix; cells whose
//      image lies outside the window drop out here (a window sees a tenth of a 350 x 512 map), the others are compacted into
//      a per-wave LDS list so that step 3 runs on full wavefronts whatever the order of the cell list;
//   3. the 3 x 3 pixels around the forward position are tested with cv::warpAffine's own 22.10 formula: a pixel whose source
//      cell is (X, Y) gets the cell's value.  A rotation has scale 1: the pixels whose source coordinate rounds to (X, Y) lie
//      within 0.5 (|cos| + |sin|) + 2^-9 <= 0.71 px of the forward image of the cell centre in each axis, hence within the
//      3 x 3 block around its rounding -- the test is exact, the neighbourhood only has to contain the candidates.
// Lists of the non-zero cells are built per map entry (ego_cells_kernel: a counting pass sizes them, round 4); an entry with
// more than `cap` of them (a pool entry re-sampled after the lists were sized) is drawn pixel by pixel (ego_image_slow), and
// the host routes whole calls whose maps are dense -- or whose border value is not 0 -- to the sampling kernels above.
constexpr int kEgoCellCapMin = 512;   // least stride of a list (a RandomMiniEnv world: <= 2 x 183 cells; pool entries change)

// list[entry][k] = value << 24 | row << 12 | column of the k-th non-zero cell inside the entry's valid region (any order);
// counts[entry] = how many there are (may exceed `cap`: the list then holds the first `cap` found); *max_count = running
// maximum over the entries built so far.  One workgroup per entry.  cells == nullptr: the counting pass only.
__global__ void __launch_bounds__(256) ego_cells_kernel(const uint8_t* __restrict__ data, EntrySelect sel, int rows, int cols,
                                                        const int32_t* __restrict__ valid_rows,
                                                        const int32_t* __restrict__ valid_cols, int cap,
                                                        uint32_t* __restrict__ cells, int32_t* __restrict__ counts,
                                                        int32_t* __restrict__ max_count)
{
    __shared__ int n_found;
    const int64_t total = sel.size();
    for (int64_t it = blockIdx.x; it < total; it += gridDim.x) {
        const int64_t m = sel.entry(it);
        const int vr = valid_rows ? valid_rows[m] : rows, vc = valid_cols ? valid_cols[m] : cols;
        if (threadIdx.x == 0) n_found = 0;
        __syncthreads();
        const uint8_t* src = data + m * (int64_t)rows * cols;
        const int cells_total = rows * cols;
        for (int base = 4 * (int)threadIdx.x; base < cells_total; base += 4 * 256) {
            uint32_t four = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (base + j < cells_total) four |= (uint32_t)src[base + j] << (8 * j);
            if (four == 0) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t v = (four >> (8 * j)) & 255u;
                if (v == 0) continue;
                const int idx = base + j, r = idx / cols, c = idx - r * cols;
                if (r >= vr || c >= vc) continue;
                const int at = atomicAdd(&n_found, 1);
                if (cells && at < cap) cells[m * cap + at] = (v << 24) | ((uint32_t)r << 12) | (uint32_t)c;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            counts[m] = n_found;
            atomicMax(max_count, n_found);
        }
        __syncthreads();
    }
}

// one image, pixel by pixel from global memory (entries whose list overflowed: rare, correctness only)
__device__ __forceinline__ void ego_image_slow(const EgoArgs& a, const EgoImage& I, const uint8_t* __restrict__ src,
                                               uint8_t* __restrict__ image, int lane)
{
    const int P = a.drows * a.dcols;
    for (int p = lane; p < P; p += 64) {
        const int y = p / a.dcols, x = p - y * a.dcols;
        const int X = (sat_int((I.m1 * y + I.m2) * 1024) + 512 + sat_int(I.m0 * x * 1024)) >> 10;
        const int Y = (sat_int((I.m4 * y + I.m5) * 1024) + 512 + sat_int(I.m3 * x * 1024)) >> 10;
        uint8_t v = (uint8_t)a.border;
        if ((unsigned)X < (unsigned)I.vc && (unsigned)Y < (unsigned)I.vr) v = src[(int64_t)Y * a.cols + X];
        image[p] = v;
    }
}

typedef uint32_t EgoU32x4 __attribute__((ext_vector_type(4)));

constexpr int kEgoHeld = 768;   // cells a wave can hold back in LDS between the culling pass and the patches

// LDS of ego_sparse_kernel, per wave: cv::hal::warpAffine's column terms {adelta, bdelta}(x) = {sat(M0 x 1024), sat(M3 x 1024)}
// for every column of the window, its row terms {sat((M1 y + M2) 1024) + 512, sat((M4 y + M5) 1024) + 512} for every row,
// and the list of cells held back.
static size_t ego_sparse_lds_bytes(int drows, int dcols, int waves)
{
    return (size_t)waves * ((size_t)(drows + dcols) * 8 + (size_t)kEgoHeld * 4);
}

typedef int EgoI32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) EgoI32x2* LdsI32x2;

// the 3 x 3 candidates of one source cell: warpAffine's expression, from the tables, decides which of them copy the cell.
// The forward position only has to put the candidates inside the block: float32 is within 2e-3 px of the float64 value
// (every term is below 6 000 px when the sum is anywhere near the window), and a pixel that samples the cell lies within
// 0.71 + 2^-9 px of the true position, so within 1.22 < 1.5 px of its rounding.
__device__ __forceinline__ void ego_patch_cell(const EgoArgs& a, LdsI32x2 col_tab, LdsI32x2 row_tab, uint8_t* __restrict__ image,
                                               uint32_t cell, float f0, float f1, float f2, float f3, float f4, float f5)
{
    const int sx = (int)(cell & 0xFFFu), sy = (int)((cell >> 12) & 0xFFFu);
    const uint8_t v = (uint8_t)(cell >> 24);
    const float fx = f0 * (float)sx + f1 * (float)sy + f2, fy = f3 * (float)sx + f4 * (float)sy + f5;
    const int xc = (int)rintf(fx), yc = (int)rintf(fy);
    EgoI32x2 ct[3];
    bool xin[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int x = xc - 1 + j;
        xin[j] = (unsigned)x < (unsigned)a.dcols;
        ct[j] = col_tab[xin[j] ? x : 0];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int y = yc - 1 + r;
        if ((unsigned)y >= (unsigned)a.drows) continue;
        const EgoI32x2 rt = row_tab[y];
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (xin[j] && ((rt.x + ct[j].x) >> 10) == sx && ((rt.y + ct[j].y) >> 10) == sy)
                image[(int64_t)y * a.dcols + (xc - 1 + j)] = v;
    }
}

// One wavefront per image (border value 0, cell lists built).  Order of a wave's work, chosen for what it waits on:
//   A. the cell list is read and culled FIRST (four loads in flight per lane; the cells whose forward image meets the window
//      are compacted into the wave's LDS list) -- loads and stores retire through one in-order counter on this chip, so a
//      list load issued behind the zero fill would wait for every store of the fill --, and the wave writes warpAffine's
//      column and row terms of THIS image into its LDS tables (the float64 arithmetic of the exact test, (drows + dcols) x 2
//      evaluations per image instead of 12 per cell: round 4, the patches were 4 k cycles per 64 cells without the tables);
//   B. the zero fill (all the HBM traffic there is);
//   C. s_waitcnt vmcnt(0) -- a patch must not be overtaken by the zeros: same wave, same addresses, no other ordering --,
//      then the patches, a held cell per lane: integer adds and compares on table entries.
// More than kEgoHeld cells inside one window (a dense corner of an otherwise sparse map): further passes of A and C.
__global__ void __launch_bounds__(64 * kEgoWaves) ego_sparse_kernel(const EgoArgs a, const uint32_t* __restrict__ cells,
                                                                     const int32_t* __restrict__ counts, int cap)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const int64_t P = (int64_t)a.drows * a.dcols;
    const int64_t first = (int64_t)blockIdx.x * waves + wave, stride = (int64_t)gridDim.x * waves;
    const int per_wave_words = (a.drows + a.dcols) * 2 + kEgoHeld;
    const LdsI32x2 col_tab = (LdsI32x2)((__attribute__((address_space(3))) uint32_t*)ego_lds + wave * per_wave_words);
    const LdsI32x2 row_tab = col_tab + a.dcols;
    __attribute__((address_space(3))) uint32_t* const mine_cells = (__attribute__((address_space(3))) uint32_t*)(row_tab + a.drows);
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
            uint8_t* const image = a.out + img * P;
            const int n_cells = counts[I.g];
            if (n_cells > cap) {
                ego_image_slow(a, I, a.data + I.g * a.map_stride, image, lane);
                continue;
            }
            // forward map of a source cell: the inverse of the dst -> src matrix
            float f0, f1, f2, f3, f4, f5;
            {
                const double det = I.m0 * I.m4 - I.m1 * I.m3;
                const double id = det != 0.0 ? 1.0 / det : 0.0;
                const double d0 = I.m4 * id, d1 = -I.m1 * id, d3 = -I.m3 * id, d4 = I.m0 * id;
                f0 = (float)d0;
                f1 = (float)d1;
                f2 = (float)(-(d0 * I.m2 + d1 * I.m5));
                f3 = (float)d3;
                f4 = (float)d4;
                f5 = (float)(-(d3 * I.m2 + d4 * I.m5));
            }
            const float x_hi = (float)a.dcols + 1.0f, y_hi = (float)a.drows + 1.0f;
            const uint32_t* const list = cells + I.g * (int64_t)cap;
            auto meets_window = [&](uint32_t cell) -> bool {
                const float sx = (float)(cell & 0xFFFu), sy = (float)((cell >> 12) & 0xFFFu);
                const float fx = f0 * sx + f1 * sy + f2, fy = f3 * sx + f4 * sy + f5;
                return cell != 0u && fx > -2.0f && fx < x_hi && fy > -2.0f && fy < y_hi;   // (a listed cell has a non-zero value byte)
            };
            // ---- A / B / C, in passes: cull list cells into the wave's LDS list until the list is read or the LDS list is
            // full; (first pass only: this image's tables, the zero fill, the wait for it); patch what is held.  One pass
            // unless a window holds more than kEgoHeld cells (a dense corner of an otherwise sparse map): a later pass's list
            // loads wait for the patch stores before them -- once per pass, not once per 64 cells (round 4's first form
            // streamed the list 64 cells at a time behind the fill: 3.4 ms per call on the 350 x 512 AisleTurn map, where a
            // third of the windows hold more than 256 cells).
            int next = 0;            // first list cell not culled yet (uniform)
            bool filled = false;
            do {
                int held = 0;        // cells in the wave's LDS list (uniform)
                bool full = false;
                for (int c0 = next; c0 < n_cells && !full; c0 += 256) {
                    uint32_t cell[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) cell[u] = c0 + 64 * u + lane < n_cells ? list[c0 + 64 * u + lane] : 0u;
                    next = c0 + 256;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const bool in = meets_window(cell[u]);
                        const uint64_t hits = __ballot(in);
                        const int more = __builtin_popcountll(hits);
                        if (held + more > kEgoHeld) {   // (this chunk of 64 starts the next pass)
                            full = true;
                            next = c0 + 64 * u;
                            break;
                        }
                        if (in) mine_cells[held + __builtin_amdgcn_mbcnt_hi((uint32_t)(hits >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hits, 0u))] = cell[u];
                        held += more;
                    }
                }
                if (!filled) {
                    // warpAffine's terms of this image, if anything is to be patched
                    if (held > 0) {
                        for (int t = lane; t < a.dcols + a.drows; t += 64) {
                            EgoI32x2 e;
                            if (t < a.dcols) {
                                e.x = sat_int(I.m0 * t * 1024);
                                e.y = sat_int(I.m3 * t * 1024);
                            } else {
                                const int y = t - a.dcols;
                                e.x = sat_int((I.m1 * y + I.m2) * 1024) + 512;
                                e.y = sat_int((I.m4 * y + I.m5) * 1024) + 512;
                            }
                            col_tab[t] = e;   // (row_tab follows col_tab)
                        }
                    }
                    // zeros: bytes up to the first 16-byte boundary, aligned 16-byte stores, the tail
                    const int head = (int)((16u - (uint32_t)(uintptr_t)image) & 15u);
                    const int64_t body = (P - head) >> 4;           // whole 16-byte pieces
                    const int tail = (int)(P - head - (body << 4));
                    if (lane < head) image[lane] = 0;
                    EgoU32x4* const q = reinterpret_cast<EgoU32x4*>(image + head);
                    const EgoU32x4 zero = {0u, 0u, 0u, 0u};
                    for (int64_t c = lane; c < body; c += 64) q[c] = zero;   // (non-temporal stores: 7 % slower)
                    if (lane < tail) image[head + (body << 4) + lane] = 0;
                    filled = true;
                    if (held == 0) break;   // (nothing in the window; a list that fills LDS holds something)
                    // the patches go behind the zeros (measured alternatives to this wait, round 3: all the fills of a batch
                    // of images first 0.32 against 0.20 ms; a two-stage pipeline with s_waitcnt vmcnt(16) 0.27 ms)
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
                }
                __builtin_amdgcn_wave_barrier();
                for (int c0 = 0; c0 < held; c0 += 64)
                    if (c0 + lane < held) ego_patch_cell(a, col_tab, row_tab, image, mine_cells[c0 + lane], f0, f1, f2, f3, f4, f5);
                __builtin_amdgcn_wave_barrier();
            } while (next < n_cells);
        }
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

}  // namespace bcp
