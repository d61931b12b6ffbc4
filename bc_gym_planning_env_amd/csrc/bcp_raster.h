// bcp_raster.h -- footprint rasterisation + costmap collision test, without materialising the kernel image.
//
// Reference semantics (utilities/path_tools.py:122-162 + envs/base/env.py:464-489):
//   mask   = cv2.fillPoly(zeros(2*half+1), [round(R(theta) * footprint / res) + half], 255)
//   hit    = any(costmap[py + ky - half_y, px + kx - half_x] == 254  for (ky,kx) in nonzero(mask), inside the map)
// cv2.fillPoly on an integer contour sets  OUTLINE | SPANS:
//   OUTLINE: every polygon edge drawn with the 8-connected Bresenham LineIterator (left-to-right start point,
//            err = D - 2d, minor step when err < 0)  ->  per (edge,row) ONE contiguous pixel run;
//   SPANS:   even-odd scanline over non-horizontal edges active on y0 <= y < y1 with 16.16 fixed-point x
//            (dx = (DX << 16) / DY, C truncation), pairs of x-sorted crossings filled on [ceil(xa), floor(xb)].
// Both parts are unions of pixel runs, so   hit = OR over runs of (lethal_bits(row) & run) != 0   against a 1-bit
// lethal mask of the costmap: same verdict as the reference, bit for bit, with no image.
//
// Coordinates: everything is translation invariant (Bresenham works on differences, 16.16 ceil/floor commute with
// integer shifts), so runs are enumerated in CENTRED pixel coordinates (u, v) = round(R(theta) * footprint / res):
// kernel pixel (kx, ky) = (u + half_x, v + half_y), map cell = (px + u, py + v).  The half sizes of
// path_tools.py:147-149 are only needed to size the image (bcp_pixel_footprint), never for the collision test.
#pragma once

#include "bcp_device.h"

namespace bcp {

struct MapXform {
    double ox, oy, inv_res;  // origin, 1.0 / resolution  (coordinate_transformations.py:204)
};

// Per-thread scratch in LDS: 2 words per footprint vertex, laid out [word][vertex][thread] so that the lanes of a
// wave hit consecutive banks.
// LDS pointers keep their address space in the type: every access is a ds_* instruction, never a flat one.
typedef __attribute__((address_space(3))) uint32_t* LdsU32;
typedef const __attribute__((address_space(3))) uint32_t* LdsWords;
typedef const __attribute__((address_space(3))) double* LdsF64;

struct VertLds {
    LdsU32 base;     // this thread's column
    int stride;      // threads per block
    __device__ __forceinline__ void put_vertex(int k, int u, int v) const
    {
        base[(2 * k) * stride] = ((uint32_t)u & 0xFFFFu) | ((uint32_t)v << 16);
    }
    __device__ __forceinline__ void get_vertex(int k, int& u, int& v) const
    {
        const uint32_t a = base[(2 * k) * stride];
        u = (int)(int16_t)(a & 0xFFFFu);
        v = (int)a >> 16;
    }
    // 16.16 slope of edge k = (V[k-1] -> V[k]); meaningless for horizontal edges
    __device__ __forceinline__ void put_slope(int k, int dxfp) const { base[(2 * k + 1) * stride] = (uint32_t)dxfp; }
    __device__ __forceinline__ int get_slope(int k) const { return (int)base[(2 * k + 1) * stride]; }
};

// rotated, resolution-scaled vertex k  (path_tools.py:142-145; np.dot's 2-term ddot is fma(a1,b1,a0*b0))
__device__ __forceinline__ void footprint_vertex(const DevParams& P, int k, double c, double s, double& px, double& py)
{
    const double qx = P.qverts[k][0], qy = P.qverts[k][1];
    px = fma(qy, -s, qx * c);
    py = fma(qy, c, qx * s);
}

// OUTLINE runs of edge (x0,y0)->(x1,y1): cv::LineIterator(pt0, pt1, connectivity 8, leftToRight = true).
// Returns true when the sink asked to stop.
template <typename Sink>
__device__ __forceinline__ bool outline_edge(int x0, int y0, int x1, int y1, Sink& sink)
{
    int sx = x0, sy = y0, dx = x1 - x0, dy = y1 - y0;
    if (dx < 0) {  // start from the end point with the smaller x
        dx = -dx;
        dy = -dy;
        sx = x1;
        sy = y1;
    }
    int ystep = 1;
    if (dy < 0) {
        dy = -dy;
        ystep = -1;
    }
    bool stop = false;
    if (dy > dx) {
        // y-major: one pixel per row, x = sx + floor((2*dx*i + dy - 1) / (2*dy))  (minor step when err < 0)
        int rem = dy - 1, q = 0;
        const int two_d = 2 * dx, two_D = 2 * dy;
        int yy = sy;
        for (int i = 0; i <= dy; ++i) {
            if (!stop) stop = sink.pixel(yy, sx + q);
            rem += two_d;
            if (rem >= two_D) {
                rem -= two_D;
                ++q;
            }
            yy += ystep;
        }
    } else if (dy == 0) {
        stop = sink.span(sy, sx, sx + dx);  // horizontal edge or single point
    } else {
        // x-major: row j holds steps i in [lo, hi], hi(j) = min(dx, floor((2*dx*j + dx) / (2*dy))), lo(j) = hi(j-1)+1
        const int two_d = 2 * dy;
        const int qs = (2 * dx) / two_d, rs = (2 * dx) - qs * two_d;
        int q = dx / two_d, rem = dx - q * two_d;
        int lo = 0, yy = sy;
        for (int j = 0; j <= dy; ++j) {
            const int hi = q > dx ? dx : q;
            if (!stop) stop = sink.span(yy, sx + lo, sx + hi);
            lo = hi + 1;
            q += qs;
            rem += rs;
            if (rem >= two_d) {
                rem -= two_d;
                ++q;
            }
            yy += ystep;
        }
    }
    return stop;
}

// Enumerates the pixel runs of the filled footprint in centred coordinates:
//   sink.pixel(v, u) / sink.span(v, ua, ub) -> bool   (return true to stop early)
// Runs may overlap; their union is exactly the pixel set cv2.fillPoly writes.  Returns true if stopped.
template <typename Sink>
__device__ __forceinline__ bool raster_runs(const DevParams& P, double c, double s, const VertLds& L, Sink& sink)
{
    const int K = P.n_verts;
    // ---- integer vertices (path_tools.py:150), OUTLINE, slopes, monotone-chain analysis
    double fx, fy;
    footprint_vertex(P, K - 1, c, s, fx, fy);
    int x0 = (int)rint(fx), y0 = (int)rint(fy);  // pt0 = v[count-1]  (CollectPolyEdges)
    bool stop = false;
    int n_edges = 0;        // non-horizontal edges
    int changes = 0;        // sign changes of dy along the contour (horizontal edges skipped)
    int first_sign = 0, prev_sign = 0;
    int top = -1;           // vertex where a falling run (dy < 0) turns into a rising one (dy > 0): a local min of y
    int first_up = -1;      // start vertex of the first rising edge
    int ymax = -0x7fffffff;
    for (int k = 0; k < K; ++k) {
        footprint_vertex(P, k, c, s, fx, fy);
        const int x1 = (int)rint(fx), y1 = (int)rint(fy);
        L.put_vertex(k, x1, y1);
        if (!stop) stop = outline_edge(x0, y0, x1, y1, sink);
        const int dy = y1 - y0;
        if (dy != 0) {
            L.put_slope(k, ((x1 - x0) * 65536) / dy);  // edge.dx, C division truncates toward zero
            ++n_edges;
            const int sign = dy > 0 ? 1 : -1;
            if (prev_sign == 0) {
                first_sign = sign;
                if (sign > 0) first_up = (k + K - 1) % K;
            } else if (sign != prev_sign) {
                ++changes;
                if (sign > 0) top = (k + K - 1) % K;
            }
            if (sign > 0 && first_up < 0) first_up = (k + K - 1) % K;
            prev_sign = sign;
        }
        ymax = max(ymax, y1);
        x0 = x1;
        y0 = y1;
    }
    if (stop) return true;
    if (n_edges < 2) return false;  // FillEdgeCollection: total < 2
    if (prev_sign != first_sign) {  // wrap-around between the last and the first non-horizontal edge
        ++changes;
        if (first_sign > 0) top = first_up;
    }

    if (changes == 2) {
        // ---- SPANS, fast path: the contour is two y-monotone chains, so every row y0 <= y < y1 has exactly two
        // active edges: one on the chain walked forward from `top`, one on the chain walked backward.
        int ka = top, kb = (top + 1) % K;  // ka: forward cursor sits on edge ka -> next is ka+1; kb: backward
        int ua, va, ub, vb;
        L.get_vertex(top, ua, va);
        ub = ua;
        vb = va;
        int xa = 0, xb = 0, dxa = 0, dxb = 0, enda = va, endb = vb;
        for (int y = va; y < ymax; ++y) {
            while (y >= enda) {  // advance the forward chain to the next non-horizontal edge
                ka = ka + 1 == K ? 0 : ka + 1;
                int u1, v1;
                L.get_vertex(ka, u1, v1);
                if (v1 != va) {
                    xa = ua << 16;
                    dxa = L.get_slope(ka);
                    enda = v1;
                }
                ua = u1;
                va = v1;
            }
            while (y >= endb) {  // advance the backward chain: edge kb joins V[kb-1] and V[kb]
                kb = kb == 0 ? K - 1 : kb - 1;
                const int kprev = kb == 0 ? K - 1 : kb - 1;
                int u1, v1;
                L.get_vertex(kprev, u1, v1);
                if (v1 != vb) {
                    xb = ub << 16;
                    dxb = L.get_slope(kb);
                    endb = v1;
                }
                ub = u1;
                vb = v1;
            }
            const int lo = min(xa, xb), hi = max(xa, xb);
            const int x1s = (lo + 65535) >> 16, x2s = hi >> 16;
            if (x1s <= x2s && sink.span(y, x1s, x2s)) return true;
            xa += dxa;
            xb += dxb;
        }
        return false;
    }

    // ---- SPANS, general contour: per row pair up the x-sorted crossings of all active edges.
    int ymin = 0x7fffffff;
    for (int k = 0; k < K; ++k) {
        int u, v;
        L.get_vertex(k, u, v);
        ymin = min(ymin, v);
    }
    for (int yy = ymin; yy < ymax; ++yy) {
        int prev_x = -0x7fffffff - 1, prev_e = -1;
        int remaining = -1;  // active crossings not yet paired (known after the first sweep)
        for (;;) {
            int ax = 0x7fffffff, ae = K, bx = 0x7fffffff, be = K, cnt = 0;
            int pu, pv;
            L.get_vertex(K - 1, pu, pv);
            for (int e = 0; e < K; ++e) {
                int cu, cv;
                L.get_vertex(e, cu, cv);
                const int ey0 = min(pv, cv), ey1 = max(pv, cv);
                if (yy >= ey0 && yy < ey1) {
                    ++cnt;
                    const int ex = (pv < cv ? pu : cu) << 16;
                    const int xe = ex + (yy - ey0) * L.get_slope(e);
                    const bool after = xe > prev_x || (xe == prev_x && e > prev_e);
                    if (after) {
                        if (xe < ax || (xe == ax && e < ae)) {
                            bx = ax; be = ae; ax = xe; ae = e;
                        } else if (xe < bx || (xe == bx && e < be)) {
                            bx = xe; be = e;
                        }
                    }
                }
                pu = cu;
                pv = cv;
            }
            if (remaining < 0) remaining = cnt;
            if (ae == K || be == K) break;
            const int x1s = (ax + 65535) >> 16, x2s = bx >> 16;
            if (x1s <= x2s && sink.span(yy, x1s, x2s)) return true;
            remaining -= 2;
            if (remaining < 2) break;
            prev_x = bx;
            prev_e = be;
        }
    }
    return false;
}

// Sink testing runs against the 1-bit lethal mask: bit (c & 31) of words[r * wpr + (c >> 5)] <=> costmap[r, c] == 254.
// Cells outside the map are ignored (env.py:483-484).
template <typename WordPtr>
struct CollisionSink {
    WordPtr words;
    int rows, cols, wpr;
    int px, py;  // world_to_pixel of the robot origin (coordinate_transformations.py:185-205)
    __device__ __forceinline__ bool pixel(int v, int u) const
    {
        const int r = py + v, c = px + u;
        if ((unsigned)r >= (unsigned)rows || (unsigned)c >= (unsigned)cols) return false;
        return (words[r * wpr + (c >> 5)] >> (c & 31)) & 1u;
    }
    __device__ __forceinline__ bool span(int v, int ua, int ub) const
    {
        const int r = py + v;
        if ((unsigned)r >= (unsigned)rows) return false;
        const int c0 = max(px + ua, 0), c1 = min(px + ub, cols - 1);
        if (c0 > c1) return false;
        const int w0 = c0 >> 5, w1 = c1 >> 5;
        const int base = r * wpr;
        const uint32_t m0 = 0xFFFFFFFFu << (c0 & 31), m1 = 0xFFFFFFFFu >> (31 - (c1 & 31));
        if (w0 == w1) return (words[base + w0] & m0 & m1) != 0;
        uint32_t acc = (words[base + w0] & m0) | (words[base + w1] & m1);
        for (int w = w0 + 1; w < w1; ++w) acc |= words[base + w];
        return acc != 0;
    }
};

// pose_collides (envs/base/env.py:464-489)
template <typename WordPtr>
__device__ __forceinline__ bool pose_collides(const DevParams& P, double x, double y, double th, const MapXform& X,
                                              WordPtr words, int rows, int cols, int wpr, const VertLds& L)
{
    CollisionSink<WordPtr> sink;
    sink.words = words;
    sink.rows = rows;
    sink.cols = cols;
    sink.wpr = wpr;
    sink.px = (int)rint((x - X.ox) * X.inv_res);
    sink.py = (int)rint((y - X.oy) * X.inv_res);
    return raster_runs(P, cos(th), sin(th), L, sink);
}

// half sizes of the kernel image (path_tools.py:147-149)
__device__ __forceinline__ void footprint_half_sizes(const DevParams& P, double c, double s, int& hx, int& hy)
{
    double mx = -INFINITY, nx = INFINITY, my = -INFINITY, ny = INFINITY;
    for (int k = 0; k < P.n_verts; ++k) {
        double px, py;
        footprint_vertex(P, k, c, s, px, py);
        mx = px > mx ? px : mx;
        nx = px < nx ? px : nx;
        my = py > my ? py : my;
        ny = py < ny ? py : ny;
    }
    hx = (int)ceil(mx > -nx ? mx : -nx);
    hy = (int)ceil(my > -ny ? my : -ny);
}

}  // namespace bcp
