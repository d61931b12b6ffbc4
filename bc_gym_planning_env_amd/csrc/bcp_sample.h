// bcp_sample.h -- RandomMiniEnv worlds sampled on the device ("true on-device sampling"):
// _sample_mini_env_params (envs/mini_env.py:328-359) -- rejection sampling of an obstacle wedge and a start / end pose
// (:269-325), the two walls (envs/base/maps.py:28-44 -> cv2.line), and the acceptance test (pose_collides of both path
// ends, not too close to each other) -- with one WAVEFRONT per independent numpy RandomState stream:
//   * lane 0 runs the generator (MT19937 exactly as numpy's legacy RandomState: same seeding, same 53-bit doubles, the
//     same draw order as the reference) and the scalar geometry,
//   * all 64 lanes clear the candidate's lethal bitmap in LDS, test both path ends against it with the cooperative
//     rasteriser of the step kernels, and expand an accepted world's bitmap into its uint8 costmap.
// Included by bcplan.hip (entry points bcp_mini_world_seed, bcp_sample_mini_worlds).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "bcp_device.h"
#include "bcp_coop.h"

namespace bcp {

constexpr int kMtWords = 624;          // MT19937 state; word kMtWords of a chain's record is the position
constexpr int kMtRecord = kMtWords + 1;

struct MiniWorldParams {   // RandomMiniEnvParams (envs/mini_env.py:30-47) + the EnvParams fields the sampler reads
    double inner_h, inner_w, mid_margin, out_margin;
    double min_obstacle_angle, max_obstacle_angle;
    double lim_euc_dist, lim_ang_dist, angular_pose_noise_scale;
    double resolution, goal_spat_dist, goal_ang_dist;
};

typedef __attribute__((address_space(3))) uint32_t* MtLds;

// ---- numpy.random.RandomState (legacy MT19937) -----------------------------------------------------------------
// mt19937_seed(state, seed): init_genrand
__global__ void mt_seed_kernel(const int64_t* __restrict__ seeds, int64_t n_chains, uint32_t* __restrict__ state)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chains) return;
    uint32_t* key = state + c * kMtRecord;
    uint32_t seed = (uint32_t)((uint64_t)seeds[c] & 0xffffffffull);
    for (int pos = 0; pos < kMtWords; ++pos) {
        key[pos] = seed;
        seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)pos + 1u;
    }
    key[kMtWords] = kMtWords;   // position: the first draw regenerates the block
}

// The stream as the wavefront sees it: a WINDOW of two consecutive 624-word blocks in LDS -- block A, the one numpy's
// record describes, and block B = its successor (the genrand update of A, done out of place and in parallel) -- and the
// position `pos` in [0, 1248) of the next unread word.  After mt_reserve() pos < 624, i.e. at least 624 words ahead are
// addressable at random: lane 0 draws the scalar quantities one after the other, and rejection loops evaluate 64 tries
// at once, every lane reading the words ITS try would have consumed, before the position is advanced past the first
// try that succeeded.  Same numbers, same order as RandomState; only the waiting is gone.
struct MtStream {
    MtLds buf;   // [2][624] blocks, then the position word
    int cur;     // which of the two blocks is A (wave-uniform)
    __device__ __forceinline__ MtLds block_a() const { return buf + cur * kMtWords; }
    __device__ __forceinline__ MtLds block_b() const { return buf + (cur ^ 1) * kMtWords; }
    __device__ __forceinline__ uint32_t pos() const { return buf[2 * kMtWords]; }
    __device__ __forceinline__ void set_pos(uint32_t p) const { buf[2 * kMtWords] = p; }
    // rk_random's output number `at` of the window (tempered)
    __device__ __forceinline__ uint32_t word(uint32_t at) const
    {
        uint32_t y = at < (uint32_t)kMtWords ? block_a()[at] : block_b()[at - kMtWords];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
    // rk_double: 53-bit double in [0, 1) from the two outputs at `at`; RandomState.random_sample() / .rand()
    __device__ __forceinline__ double real(uint32_t at) const
    {
        const uint32_t a = word(at) >> 5, b = word(at + 1) >> 6;
        return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
    }
    // lane 0 only: the next double / RandomState.uniform(low, high) = low + (high - low) * random_sample()
    __device__ __forceinline__ double next_real() const
    {
        const uint32_t at = pos();
        set_pos(at + 2);
        return real(at);
    }
    __device__ __forceinline__ double next_uniform(double low, double high) const { return low + (high - low) * next_real(); }
};

// genrand block update, out of place: dst = successor block of src.  new[i] depends on old[i], old[i + 1] and on
// old[i + 397] (i < 227) or new[i - 227] (i >= 227), so three sweeps of 227 words are each fully parallel.
__device__ __forceinline__ void mt_twist(MtLds src, MtLds dst, int lane)
{
    constexpr uint32_t kUpper = 0x80000000u, kLower = 0x7fffffffu, kMatrix = 0x9908b0dfu;
    constexpr int kShift = kMtWords - 397;   // 227
#pragma unroll 1
    for (int base = 0; base < kMtWords - 1; base += kShift) {
        const int end = min(base + kShift, kMtWords - 1);
        for (int i = base + lane; i < end; i += 64) {
            const uint32_t y = (src[i] & kUpper) | (src[i + 1] & kLower);
            const uint32_t far = i < kShift ? src[i + 397] : dst[i - kShift];
            dst[i] = far ^ (y >> 1) ^ ((y & 1u) ? kMatrix : 0u);
        }
        wave_lds_sync();
    }
    if (lane == 0) {
        const uint32_t y = (src[kMtWords - 1] & kUpper) | (dst[0] & kLower);
        dst[kMtWords - 1] = dst[396] ^ (y >> 1) ^ ((y & 1u) ? kMatrix : 0u);
    }
    wave_lds_sync();
}

// all lanes: slide the window until pos < 624 (numpy regenerates at pos == 624, a fresh seed's state)
__device__ __forceinline__ void mt_reserve(MtStream& m, int lane)
{
    wave_lds_sync();
    uint32_t at = m.pos();
    while (at >= (uint32_t)kMtWords) {
        m.cur ^= 1;                                // B becomes A ...
        mt_twist(m.block_a(), m.block_b(), lane);  // ... and gets a successor of its own
        at -= kMtWords;
    }
    if (lane == 0) m.set_pos(at);
    wave_lds_sync();
}

// ---- geometry -------------------------------------------------------------------------------------------------------
struct MiniWorld {   // MiniEnvParams (envs/mini_env.py:79-92)
    double start[3], end[3], a[2], o[2], b[2], h, w;
};

struct Wedge {   // not_inside_obstacle (envs/mini_env.py:199-208)
    double ax, ay, first, last;
    __device__ __forceinline__ bool clear_of(double x, double y) const
    {
        const double phi = normalize_angle(atan2(y - ay, x - ax));   // cart2pol, coordinate_transformations.py:124-135
        if (first <= phi && phi <= last) return false;
        return !(first <= phi + kTwoPi && phi + kTwoPi <= last);
    }
};

enum { kDrawOk = 0, kDrawEmpty = 1 /* SpaceSeemsEmptyError: redraw the obstacle */, kDrawFail = 2 /* ValueError */ };

// One rejection loop of the reference ("for _ in range(1000): draw; if ok: break"), 64 tries at a time: `try_at(at, ...)`
// evaluates the try whose first random word is window word `at` (every try consumes kWordsPerTry words, accepted or not).
// Returns the lane that holds the first accepted try (its locals are the result), or -1 after 1000 failures; the
// stream is left exactly where the sequential loop would have left it.
template <int kWordsPerTry, typename Try>
__device__ __forceinline__ int first_accepted(MtStream& m, int lane, Try try_at)
{
    static_assert(64 * kWordsPerTry <= kMtWords, "a batch of tries must fit the look-ahead window");
    for (int t0 = 0; t0 < 1000; t0 += 64) {
        mt_reserve(m, lane);
        const uint32_t at = m.pos();
        const int batch = min(64, 1000 - t0);
        const bool ok = try_at(at + (uint32_t)(lane * kWordsPerTry)) && lane < batch;
        const uint64_t hits = __ballot(ok);
        const int winner = hits ? (int)__builtin_ctzll(hits) : -1;
        wave_lds_sync();
        if (lane == 0) m.set_pos(at + (uint32_t)((winner >= 0 ? winner + 1 : batch) * kWordsPerTry));
        if (winner >= 0) return winner;
    }
    return -1;
}

// _sample_mini_env_params_no_final_check (envs/mini_env.py:269-325); all lanes call it, all lanes get W
__device__ __forceinline__ int draw_candidate(MtStream& m, const MiniWorldParams& p, MiniWorld& W, int lane)
{
    mt_reserve(m, lane);
    double o0 = 0, o1 = 0, first = 0, width = 0, pick = 0;
    if (lane == 0) {
        o0 = m.next_uniform(-p.inner_w / 2, p.inner_w / 2);
        o1 = m.next_uniform(-p.inner_h / 2, p.inner_h / 2);
        first = m.next_uniform(0, kTwoPi);
        width = m.next_uniform(p.min_obstacle_angle, p.max_obstacle_angle);
        pick = m.next_real();
    }
    W.o[0] = bcast_d(o0, 0);
    W.o[1] = bcast_d(o1, 0);
    first = bcast_d(first, 0);
    width = bcast_d(width, 0);
    pick = bcast_d(pick, 0);
    const double reach = 3 * (p.inner_h + p.inner_w + p.mid_margin + p.out_margin);
    W.h = p.inner_h + 2 * p.mid_margin + 2 * p.out_margin;
    W.w = p.inner_w + 2 * p.mid_margin + 2 * p.out_margin;
    const Wedge wedge{W.o[0], W.o[1], first, first + width};
    double sx, sy, ex, ey, heading;
    if (pick < 0.7) {
        // _sample_pose_circ (:146-180): antipodal points of a circle; a try = the angle and a heading the reference
        // draws and then overwrites
        const double radius = fmin((p.inner_w + p.inner_h) / 4. + p.mid_margin, p.lim_euc_dist);
        double x = 0, y = 0;
        const int winner = first_accepted<4>(m, lane, [&](uint32_t at) {
            const double phi = 0 + (kTwoPi - 0) * m.real(at);
            x = radius * cos(phi);
            y = radius * sin(phi);
            return wedge.clear_of(x, y) && wedge.clear_of(-x, -y);
        });
        if (winner < 0) return kDrawEmpty;
        sx = bcast_d(x, winner);
        sy = bcast_d(y, winner);
        ex = -sx;
        ey = -sy;
        heading = atan2(-sy - sy, -sx - sx);
    } else {
        // _pick_pts_square_method (:183-236); a try = x, y, heading
        const double half_w = p.inner_w / 2 + p.mid_margin, half_h = p.inner_h / 2 + p.mid_margin;
        double x = 0, y = 0, th = 0;
        int winner = first_accepted<6>(m, lane, [&](uint32_t at) {
            x = -half_w + (half_w - -half_w) * m.real(at);
            y = -half_h + (half_h - -half_h) * m.real(at + 2);
            th = normalize_angle(0 + (kTwoPi - 0) * m.real(at + 4));
            return wedge.clear_of(x, y);
        });
        if (winner < 0) return kDrawFail;
        sx = bcast_d(x, winner);
        sy = bcast_d(y, winner);
        const double sth = bcast_d(th, winner);
        winner = first_accepted<6>(m, lane, [&](uint32_t at) {
            x = -half_w + (half_w - -half_w) * m.real(at);
            y = -half_h + (half_h - -half_h) * m.real(at + 2);
            th = normalize_angle(0 + (kTwoPi - 0) * m.real(at + 4));
            const double dx = sx - x, dy = sy - y;
            return wedge.clear_of(x, y) && py_mod(sth - th, kTwoPi) < p.lim_ang_dist &&
                   sqrt(fma(dy, dy, dx * dx)) < p.lim_euc_dist;
        });
        if (winner < 0) return kDrawFail;
        ex = bcast_d(x, winner);
        ey = bcast_d(y, winner);
        heading = atan2(ey - sy, ex - sx);
    }
    mt_reserve(m, lane);
    const double half = p.angular_pose_noise_scale / 2.0;
    double n0 = 0, n1 = 0;
    if (lane == 0) {
        n0 = m.next_uniform(-half, half);
        n1 = m.next_uniform(-half, half);
    }
    n0 = bcast_d(n0, 0);
    n1 = bcast_d(n1, 0);
    const double th0 = normalize_angle(heading);   // OrientedPoint normalises on construction ...
    W.start[0] = sx;
    W.start[1] = sy;
    W.start[2] = normalize_angle(th0 + n0);          // ... and again after the noise
    W.end[0] = ex;
    W.end[1] = ey;
    W.end[2] = normalize_angle(th0 + n1);
    W.a[0] = reach * cos(first) + W.o[0];
    W.a[1] = reach * sin(first) + W.o[1];
    W.b[0] = reach * cos(first + width) + W.o[0];
    W.b[1] = reach * sin(first + width) + W.o[1];
    return kDrawOk;
}

// cv::clipLine on 64-bit points (drawing.cpp), as the oracle's clip_line
__device__ __forceinline__ bool clip_segment(int64_t cols, int64_t rows, int64_t& x1, int64_t& y1, int64_t& x2, int64_t& y2)
{
    const int64_t right = cols - 1, bottom = rows - 1;
    int c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
    int c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
        int64_t a;
        if (c1 & 12) {
            a = c1 < 8 ? 0 : bottom;
            x1 += (int64_t)((double)(a - y1) * (double)(x2 - x1) / (double)(y2 - y1));
            y1 = a;
            c1 = (x1 < 0) + (x1 > right) * 2;
        }
        if (c2 & 12) {
            a = c2 < 8 ? 0 : bottom;
            x2 += (int64_t)((double)(a - y2) * (double)(x2 - x1) / (double)(y2 - y1));
            y2 = a;
            c2 = (x2 < 0) + (x2 > right) * 2;
        }
        if ((c1 & c2) == 0 && (c1 | c2) != 0) {
            if (c1) {
                a = c1 == 1 ? 0 : right;
                y1 += (int64_t)((double)(a - x1) * (double)(y2 - y1) / (double)(x2 - x1));
                x1 = a;
                c1 = 0;
            }
            if (c2) {
                a = c2 == 1 ? 0 : right;
                y2 += (int64_t)((double)(a - x2) * (double)(y2 - y1) / (double)(x2 - x1));
                x2 = a;
                c2 = 0;
            }
        }
    }
    return (c1 | c2) == 0;
}

// cv2.line(thickness=1) into a 1-bit map in LDS: clip, then the 8-connected left-to-right Bresenham iterator.  All
// lanes call it with the same end points; pixel i of the line is at minor offset floor((2 * minor * i + major - 1) /
// (2 * major)) (the closed form of the iterator's error accumulation, as in coop_raster's outline runs), so the lanes
// set the pixels 64 at a time.
__device__ __forceinline__ void draw_wall_bits(MtLds bits, int rows, int cols, int wpr, int64_t x1, int64_t y1, int64_t x2,
                                               int64_t y2, int lane)
{
    if ((uint64_t)x1 >= (uint64_t)cols || (uint64_t)x2 >= (uint64_t)cols || (uint64_t)y1 >= (uint64_t)rows ||
        (uint64_t)y2 >= (uint64_t)rows) {
        if (!clip_segment(cols, rows, x1, y1, x2, y2)) return;
    }
    int dx = (int)(x2 - x1), dy = (int)(y2 - y1);
    int x0 = (int)x1, y0 = (int)y1;
    if (dx < 0) {
        dx = -dx;
        dy = -dy;
        x0 = (int)x2;
        y0 = (int)y2;
    }
    int step_y = 1;
    if (dy < 0) {
        dy = -dy;
        step_y = -1;
    }
    const bool vert = dy > dx;
    const int major = vert ? dy : dx, minor = vert ? dx : dy;
    for (int i = lane; i <= major; i += 64) {
        const int across = major > 0 ? (2 * minor * i + major - 1) / (2 * major) : 0;
        const int x = vert ? x0 + across : x0 + i;
        const int y = vert ? y0 + step_y * i : y0 + step_y * across;
        __hip_atomic_fetch_or(bits + (y * wpr + (x >> 5)), 1u << (x & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
}

typedef uint64_t __attribute__((aligned(1))) SampleU64Unaligned;

// ---- the sampler: one wavefront per chain -----------------------------------------------------------------------
// LDS per wave: [MT19937 window: 2 x 624 words + position] [lethal bitmap: rows * wpr words]
// worlds: [n_chains * episodes][14] = start(3), end(3), obstacle_a(2), obstacle_o(2), obstacle_b(2), h, w
// status[chain]: 0 ok, 1 = "the sampling space looks empty" (the reference raises ValueError)
// counts / first_world (optional): see "ring mode" below
// WIDE as in coop_collides: one copy of the rasteriser per kernel
constexpr int kSampleLdsWords = 2 * kMtWords + 2;   // the bitmap follows the window (on an even word)
// ... and behind the bitmap (rounded up to an even word): the footprint vertices / resolution as doubles [2 * BCP_MAX_VERTS] and the
// cell list of the sparse exact test [kSparseLdsWords] (coop_collides_sparse: what the step kernels settle their parked poses with)
__host__ __device__ __forceinline__ size_t sample_lds_words(int rows, int wpr)
{
    return (size_t)((kSampleLdsWords + rows * wpr + 1) & ~1) + 4 * BCP_MAX_VERTS + kSparseLdsWords;
}

template <bool WIDE>
__global__ void __launch_bounds__(64) mini_world_sample_kernel(DevParams P, MiniWorldParams mp, uint32_t* __restrict__ mt_state,
                                                               int64_t n_chains, int episodes, int rows, int cols,
                                                               const int32_t* __restrict__ counts,
                                                               const int64_t* __restrict__ first_world,
                                                               double* __restrict__ worlds, uint8_t* __restrict__ maps,
                                                               int32_t* __restrict__ status)
{
    extern __shared__ uint32_t sample_lds[];
    const int lane = threadIdx.x;
    const int64_t chain = blockIdx.x;
    if (chain >= n_chains) return;
    const int wpr = (cols + 31) / 32;
    if (counts && counts[chain] <= 0) return;   // ring mode: nothing to top up behind this chain's env
    MtStream mt{(MtLds)sample_lds, 0};
    const MtLds bits = mt.buf + kSampleLdsWords;
    __attribute__((address_space(3))) double* const qv_w =
        (__attribute__((address_space(3))) double*)(mt.buf + ((kSampleLdsWords + rows * wpr + 1) & ~1));
    const LdsF64 qv = qv_w;
    const LdsU32 cell_list = (LdsU32)(qv_w + 2 * BCP_MAX_VERTS);
    if (lane < P.n_verts) {
        qv_w[2 * lane] = P.qverts[lane][0];
        qv_w[2 * lane + 1] = P.qverts[lane][1];
    }
    uint32_t* record = mt_state + chain * kMtRecord;   // numpy's: 624 key words + position (624 = "regenerate first")
    for (int k = lane; k < kMtWords; k += 64) mt.buf[k] = record[k];
    if (lane == 0) mt.set_pos(min(record[kMtWords], (uint32_t)kMtWords));
    wave_lds_sync();
    mt_twist(mt.block_a(), mt.block_b(), lane);
    const double vqx = lane < P.n_verts ? P.qverts[lane][0] : 0.0, vqy = lane < P.n_verts ? P.qverts[lane][1] : 0.0;
    const double inv_res = 1.0 / mp.resolution;
    int failed = 0;
    // ring mode (counts != nullptr): this chain produces its next counts[chain] worlds; world number j of the stream
    // (first_world[chain], first_world[chain] + 1, ...) goes to slot j % episodes of the chain's `episodes` pool entries
    const int n_new = counts ? min(counts[chain], episodes) : episodes;
    const int64_t j0 = counts ? first_world[chain] : 0;
    for (int e = 0; e < n_new && !failed; ++e) {
        bool accepted = false;
        MiniWorld W;
        for (int tries = 0; tries < 1000 && !accepted && !failed; ++tries) {
            const int rc = draw_candidate(mt, mp, W, lane);
            if (rc == kDrawFail) failed = 1;
            if (rc != kDrawOk) continue;
            // prepare_map_and_path (:362-388): empty map with origin (-h/2, -w/2), two walls from the apex
            const double ox = -W.h / 2., oy = -W.w / 2.;
            for (int k = lane; k < rows * wpr; k += 64) bits[k] = 0u;
            wave_lds_sync();
            const int64_t px_o = (int64_t)rint((W.o[0] - ox) * inv_res), py_o = (int64_t)rint((W.o[1] - oy) * inv_res);
            draw_wall_bits(bits, rows, cols, wpr, px_o, py_o, (int64_t)rint((W.a[0] - ox) * inv_res),
                           (int64_t)rint((W.a[1] - oy) * inv_res), lane);
            draw_wall_bits(bits, rows, cols, wpr, px_o, py_o, (int64_t)rint((W.b[0] - ox) * inv_res),
                           (int64_t)rint((W.b[1] - oy) * inv_res), lane);
            wave_lds_sync();
            // pose_collides of the two ends of the coarse path (:338-343), cooperatively
            bool collides = false;
#pragma unroll 1
            for (int end = 0; end < 2; ++end) {
                const double x = end ? W.end[0] : W.start[0], y = end ? W.end[1] : W.start[1];
                const double th = end ? W.end[2] : W.start[2];
                const int px = (int)rint((x - ox) * inv_res), py = (int)rint((y - oy) * inv_res);
                // (the walls are one cell thick: the lethal cells under the footprint's image, tested one by one --
                //  coop_collides_sparse, the same verdict pixel for pixel -- instead of the image rasterised row by row; a
                //  map too dense for the list falls back to the rasteriser)
                const double c = cos(th), s = sin(th);
                const int verdict = coop_collides_sparse<WIDE>(P, qv, c, s, px, py, (LdsWords)bits, rows, cols, wpr, cell_list);
                bool hit = verdict == kSparseHit;
                if (verdict == kSparseTooMany) hit = coop_collides<WIDE>(P, vqx, vqy, c, s, px, py, (LdsWords)bits, rows, cols, wpr);
                collides = collides || hit;
            }
            // beginning and goal must not be immediately too close (:345-351)
            const double dx = W.start[0] - W.end[0], dy = W.start[1] - W.end[1];
            const double dth = fabs(normalize_angle(W.start[2] - W.end[2]));
            const bool too_close = hypot(dx, dy) < mp.goal_spat_dist && dth < mp.goal_ang_dist;
            accepted = !collides && !too_close;
        }
        if (!accepted) {
            failed = 1;
            break;
        }
        const int64_t g = chain * episodes + (j0 + e) % episodes;
        if (lane == 0) {
            double* o = worlds + g * 14;
            o[0] = W.start[0];
            o[1] = W.start[1];
            o[2] = W.start[2];
            o[3] = W.end[0];
            o[4] = W.end[1];
            o[5] = W.end[2];
            o[6] = W.a[0];
            o[7] = W.a[1];
            o[8] = W.o[0];
            o[9] = W.o[1];
            o[10] = W.b[0];
            o[11] = W.b[1];
            o[12] = W.h;
            o[13] = W.w;
        }
        // the accepted world's costmap: 254 where the bitmap is set, eight cells per lane and store (the last group of a
        // row is shifted left to end at the row's end)
        uint8_t* map = maps + g * (int64_t)rows * cols;
        if (cols >= 8) {
            const int groups = (cols + 7) / 8;
            for (int it = lane; it < rows * groups; it += 64) {
                const int r = it / groups, c0 = min((it - r * groups) * 8, cols - 8);
                const int w = c0 >> 5;
                const uint64_t two = ((uint64_t)(w + 1 < wpr ? bits[r * wpr + w + 1] : 0u) << 32) | bits[r * wpr + w];
                const uint64_t eight = (two >> (c0 & 31)) & 0xffull;
                // bit b -> byte b: one bit per byte, then "byte != 0" as 0 / 1, then x 254
                const uint64_t spread = (eight * 0x0101010101010101ull) & 0x8040201008040201ull;
                const uint64_t ones = ((spread + 0x7f7f7f7f7f7f7f7full) >> 7) & 0x0101010101010101ull;
                *reinterpret_cast<SampleU64Unaligned*>(map + r * (int64_t)cols + c0) = ones * (uint64_t)BCP_LETHAL;
            }
        } else {
            for (int idx = lane; idx < rows * cols; idx += 64) {
                const int r = idx / cols, c = idx - r * cols;
                map[idx] = ((bits[r * wpr + (c >> 5)] >> (c & 31)) & 1u) ? (uint8_t)BCP_LETHAL : (uint8_t)0;
            }
        }
    }
    mt_reserve(mt, lane);
    for (int k = lane; k < kMtWords; k += 64) record[k] = mt.block_a()[k];
    if (lane == 0) {
        record[kMtWords] = mt.pos();
        status[chain] = failed;
    }
}

// ---- from sampled worlds to what a PlanEnv starts with -----------------------------------------------------------
// make_initial_state (envs/base/env.py:179-214) for every world: refine_path of the coarse (start, end) path
// (utilities/path_tools.py:178-240: points every path_delta, np.linspace arithmetic, inserted points carry the start
// heading) and the reward provider's initial state (reward.py:261-288, or :355-371 for pure pursuit).
// One thread per world.  paths: [G][max_len][3], lens: [G], init: [G][2] = (min_spat_dist_so_far, target_idx).
// status[g]: 0, 1 = path longer than max_len, 2 = "Goal pose too close to initial pose" (ValueError in the reference).
__device__ __forceinline__ void mini_world_path(const double* __restrict__ worlds, int64_t g, double path_delta, double sp,
                                                double ap, int pure_pursuit, int max_len, double* __restrict__ paths,
                                                int32_t* __restrict__ lens, double* __restrict__ init,
                                                int32_t* __restrict__ status)
{
    const double* w = worlds + g * 14;
    const double x0 = w[0], y0 = w[1], th0 = w[2], x1 = w[3], y1 = w[4], th1 = w[5];
    double* p = paths + g * (int64_t)max_len * 3;
    const double dx = x1 - x0, dy = y1 - y0;
    const double d = sqrt(dx * dx + dy * dy);   // np.linalg.norm(..., axis=1): sqrt(add.reduce(x * x))
    int m;
    int rc = 0;
    if (d > path_delta) {
        const int npoints = (int)(d / path_delta) + 2;
        m = npoints;   // npoints - 1 interpolated rows (the last linspace sample is dropped), then the end pose
        if (m > max_len) {
            rc = 1;
            m = max_len;
        }
        const double sx = dx / (double)(npoints - 1), sy = dy / (double)(npoints - 1);   // np.linspace: step = delta / div
        for (int i = 0; i < m - 1; ++i) {
            p[3 * i + 0] = (double)i * sx + x0;    // arange(num) * step + start
            p[3 * i + 1] = (double)i * sy + y0;
            p[3 * i + 2] = th0;
        }
    } else {
        m = 2;
        p[0] = x0;
        p[1] = y0;
        p[2] = th0;
    }
    p[3 * (m - 1) + 0] = x1;
    p[3 * (m - 1) + 1] = y1;
    p[3 * (m - 1) + 2] = th1;
    lens[g] = m;
    double min_dist;
    int target;
    if (pure_pursuit) {   // reward.py:355-371
        target = 1;
        min_dist = hypot(x1 - x0, y1 - y0);
    } else {              // find_last_reached(path[0], path) (path_tools.py:408-448), reward.py:261-288
        int last = -1;
        for (int j = 0; j < m; ++j) {
            const double xj = p[3 * j], yj = p[3 * j + 1], tj = p[3 * j + 2];
            const bool near = hypot(xj - x0, yj - y0) < sp;
            if (!near) continue;   // (three independent predicates: the other two cost a cos, a sin and an fmod per way point)
            const bool aligned = fabs(normalize_angle(th0 - tj)) < ap;
            const bool ahead = cos(tj) * (x0 - xj) + sin(tj) * (y0 - yj) >= -sp / 9;
            if (aligned && ahead) last = j;
        }
        if (last == m - 1) rc = 2;
        target = min(last + 1, m - 1);
        min_dist = hypot(p[3 * target] - x0, p[3 * target + 1] - y0);
    }
    init[2 * g] = min_dist;
    init[2 * g + 1] = (double)target;
    status[g] = rc;
}

__global__ void mini_world_paths_kernel(const double* __restrict__ worlds, EntrySelect sel, double path_delta, double sp,
                                        double ap, int pure_pursuit, int max_len, double* __restrict__ paths,
                                        int32_t* __restrict__ lens, double* __restrict__ init, int32_t* __restrict__ status)
{
    const int64_t total = sel.size();
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (int64_t)gridDim.x * blockDim.x)
        mini_world_path(worlds, sel.entry(it), path_delta, sp, ap, pure_pursuit, max_len, paths, lens, init, status);
}

// ---- pools that never run out: the `episodes` entries of a chain as a ring over its stream of worlds ----------------
// RandomMiniEnv.reset draws a fresh world every time (envs/mini_env.py:441-459).  With one stream per env (env c on
// chain c, entries c * E .. c * E + E - 1) the pool holds worlds generated[c] - E .. generated[c] - 1 of the stream,
// world j in entry c * E + j % E, and next_geom walks them in order except that the newest world's entry points to
// itself (an env that gets there before the next refresh repeats that world instead of wrapping onto an old one).
// This kernel looks where env c is, frees the entries behind it -- world numbers first_world[c] .. + counts[c] - 1 are
// to be sampled into them -- closes the ring behind the last of them (the new guard) and appends the freed entries to
// `dirty`.  The OLD guard stays shut until mini_world_ring_release_kernel, after everything has been rewritten: steps
// may run while the new worlds are being made (they neither read nor reach a dirty entry).
// info[0] = number of dirty entries, info[1] = envs found on the guard entry (they may have repeated a world).
__global__ void __launch_bounds__(256) mini_world_ring_plan_kernel(int64_t n_chains, int episodes,
                                                                   const int32_t* __restrict__ geom_of_env,
                                                                   int32_t* __restrict__ next_geom,
                                                                   int64_t* __restrict__ generated, int32_t* __restrict__ counts,
                                                                   int64_t* __restrict__ first_world, int32_t* __restrict__ dirty,
                                                                   int32_t* __restrict__ info)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int E = episodes;
    int count = 0;
    int64_t gen = 0;
    bool starved = false;
    if (c < n_chains) {
        gen = generated[c];
        const int64_t oldest = gen - E;
        const int64_t slot = (int64_t)geom_of_env[c] - c * E;
        const int64_t world = oldest + (((slot - oldest) % E) + E) % E;   // the world number env c is on
        count = (int)(world - oldest);
        starved = world == gen - 1;
        counts[c] = count;
        first_world[c] = gen;
        generated[c] = gen + count;
        if (count > 0) {
            const int32_t guard = (int32_t)(c * E + (gen + count - 1) % E);
            next_geom[guard] = guard;
        }
    }
    int incl = count;   // wave-wide inclusive prefix sum: one atomic per wavefront
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    const int total = __shfl(incl, 63);
    int base = 0;
    if (lane == 63 && total > 0) base = atomicAdd(&info[0], total);
    base = __shfl(base, 63);
    const uint64_t starving = __ballot(starved);
    if (lane == 0 && starving) atomicAdd(&info[1], __popcll(starving));
    for (int e = 0; e < count; ++e) dirty[base + incl - count + e] = (int32_t)(c * E + (gen + e) % E);
}

// opens the old guard of every chain that got new worlds: its env may now walk on into them
__global__ void mini_world_ring_release_kernel(int64_t n_chains, int episodes, const int32_t* __restrict__ counts,
                                               const int64_t* __restrict__ first_world, int32_t* __restrict__ next_geom)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chains || counts[c] <= 0) return;
    const int64_t gen = first_world[c];
    next_geom[c * episodes + (gen - 1) % episodes] = (int32_t)(c * episodes + gen % episodes);
}

// initial state of the selected pool entries (make_initial_state, envs/base/env.py:179-214): pose = first way point,
// everything else zero, reward state from `init`
__global__ void pool_initial_state_kernel(EntrySelect sel, const double* __restrict__ paths, int max_len,
                                          const double* __restrict__ init, DevState st)
{
    const int64_t total = sel.size();
    for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (int64_t)gridDim.x * blockDim.x) {
        const int64_t g = sel.entry(it);
        const double* p = paths + g * (int64_t)max_len * 3;
        st.x[g] = p[0];
        st.y[g] = p[1];
        st.angle[g] = p[2];
        st.v[g] = 0.0;
        st.w[g] = 0.0;
        if (st.steer) st.steer[g] = 0.0;
        if (st.wheel) st.wheel[g] = 0.0;
        st.min_dist[g] = init[2 * g];
        st.target_idx[g] = (int32_t)init[2 * g + 1];
        st.cur_iter[g] = 0;
        st.collided[g] = 0;
    }
}

}  // namespace bcp
