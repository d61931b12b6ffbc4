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
// Both parts are unions of intervals, so   hit = OR over intervals of (lethal_bits(row) & interval) != 0
// against a 1-bit lethal mask of the costmap -- same verdict as the reference, bit for bit, with no image.
#pragma once

#include "bcp_device.h"

namespace bcp {

// 1-bit lethal mask view: bit (c & 31) of words[r * wpr + (c >> 5)] <=> costmap[r, c] == 254 (inside the valid map)
struct BitmapView {
    const uint32_t* words;  // LDS (shared map) or global (per-env map)
    int rows, cols, wpr;
};

struct MapXform {
    double ox, oy, inv_res;  // origin, 1.0 / resolution  (coordinate_transformations.py:204)
};

// does any lethal bit lie on map row r, columns [c0, c1] (inclusive)?  Cells outside the map are ignored
// (env.py:483-484).
template <typename WordPtr>
__device__ __forceinline__ bool interval_hits(WordPtr words, int rows, int cols, int wpr, int r, int c0, int c1)
{
    if ((unsigned)r >= (unsigned)rows) return false;
    c0 = max(c0, 0);
    c1 = min(c1, cols - 1);
    if (c0 > c1) return false;
    int w0 = c0 >> 5, w1 = c1 >> 5;
    const int base = r * wpr;
    uint32_t acc = 0;
    for (int w = w0; w <= w1; ++w) {
        uint32_t m = 0xFFFFFFFFu;
        if (w == w0) m &= 0xFFFFFFFFu << (c0 & 31);
        if (w == w1) m &= 0xFFFFFFFFu >> (31 - (c1 & 31));
        acc |= words[base + w] & m;
    }
    return acc != 0;
}

// Per-thread edge table slot in LDS: 3 words per edge, [word][edge][thread] so consecutive lanes hit
// consecutive banks.
struct EdgeLds {
    uint32_t* base;  // this thread's column: base[(e * 3 + f) * stride]
    int stride;      // threads per block
    __device__ __forceinline__ void put(int e, int y0, int y1, int x0fp, int dxfp) const
    {
        base[(e * 3 + 0) * stride] = (uint32_t)y0 | ((uint32_t)y1 << 16);
        base[(e * 3 + 1) * stride] = (uint32_t)x0fp;
        base[(e * 3 + 2) * stride] = (uint32_t)dxfp;
    }
    __device__ __forceinline__ void get(int e, int& y0, int& y1, int& x0fp, int& dxfp) const
    {
        uint32_t a = base[(e * 3 + 0) * stride];
        y0 = (int)(a & 0xFFFFu);
        y1 = (int)(a >> 16);
        x0fp = (int)base[(e * 3 + 1) * stride];
        dxfp = (int)base[(e * 3 + 2) * stride];
    }
};

// rotated, resolution-scaled vertex k  (path_tools.py:142-145; np.dot's 2-term ddot is fma(a1,b1,a0*b0))
__device__ __forceinline__ void footprint_vertex(const DevParams& P, int k, double c, double s, double& px, double& py)
{
    double qx = P.qverts[k][0], qy = P.qverts[k][1];
    px = fma(qy, -s, qx * c);
    py = fma(qy, c, qx * s);
}

// Enumerates the footprint kernel image of get_pixel_footprint(th, footprint, res) as pixel runs:
//   sink.begin(hx, hy)            half sizes: the image is (2*hy+1) x (2*hx+1), robot origin at (hx, hy)
//   sink.emit(y, xa, xb) -> bool  run of set pixels on image row y, columns xa..xb; return true to stop early
// Runs may overlap; their union is exactly the set cv2.fillPoly writes.  `E` is this thread's LDS edge table.
// Returns true when the sink stopped the enumeration.
template <typename Sink>
__device__ bool raster_footprint(const DevParams& P, double th, const EdgeLds& E, Sink& sink)
{
    const int K = P.n_verts;
    const double c = cos(th), s = sin(th);
    // pass 1: half sizes (path_tools.py:147-149)
    double mx = -INFINITY, nx = INFINITY, my = -INFINITY, ny = INFINITY;
    for (int k = 0; k < K; ++k) {
        double px, py;
        footprint_vertex(P, k, c, s, px, py);
        mx = px > mx ? px : mx;
        nx = px < nx ? px : nx;
        my = py > my ? py : my;
        ny = py < ny ? py : ny;
    }
    const int hx = (int)ceil(mx > -nx ? mx : -nx);
    const int hy = (int)ceil(my > -ny ? my : -ny);
    sink.begin(hx, hy);

    // pass 2: integer vertices (path_tools.py:150), OUTLINE runs, edge table
    bool stop = false;
    int nE = 0, ymin = 0x7fffffff, ymax = -0x7fffffff;
    double fx, fy;
    footprint_vertex(P, K - 1, c, s, fx, fy);
    int x0 = (int)rint(fx) + hx, y0 = (int)rint(fy) + hy;  // pt0 = v[count-1]
    for (int k = 0; k < K; ++k) {
        footprint_vertex(P, k, c, s, fx, fy);
        const int x1 = (int)rint(fx) + hx, y1 = (int)rint(fy) + hy;
        // ---- Bresenham, LineIterator(pt0, pt1, 8, leftToRight) ----
        int sx = x0, sy = y0, dx = x1 - x0, dy = y1 - y0;
        if (dx < 0) {
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
        if (dy > dx) {
            // y-major: one pixel per row, x = sx + floor((2*dx*i + dy - 1) / (2*dy))
            int r2 = dy - 1, q = 0;
            const int two_d = 2 * dx, two_D = 2 * dy;
            int yy = sy;
            for (int i = 0; i <= dy; ++i) {
                if (!stop) stop = sink.emit(yy, sx + q, sx + q);
                r2 += two_d;
                if (r2 >= two_D) {
                    r2 -= two_D;
                    ++q;
                }
                yy += ystep;
            }
        } else if (dy == 0) {
            // horizontal edge or single point
            if (!stop) stop = sink.emit(sy, sx, sx + dx);
        } else {
            // x-major: row j holds i in [lo, hi], hi(j) = min(dx, floor((2*dx*j + dx) / (2*dy))), lo(j) = hi(j-1)+1
            const int two_d = 2 * dy;
            int lo = 0, yy = sy, num = dx;
            for (int j = 0; j <= dy; ++j) {
                int hi = num / two_d;
                hi = hi > dx ? dx : hi;
                if (!stop) stop = sink.emit(yy, sx + lo, sx + hi);
                lo = hi + 1;
                num += 2 * dx;
                yy += ystep;
            }
        }
        // ---- edge table (CollectPolyEdges) ----
        if (y0 != y1) {
            int ey0, ey1, ex;
            if (y0 < y1) {
                ey0 = y0; ey1 = y1; ex = x0 << 16;
            } else {
                ey0 = y1; ey1 = y0; ex = x1 << 16;
            }
            const int edx = ((x1 - x0) * 65536) / (y1 - y0);  // C division truncates toward zero
            E.put(nE, ey0, ey1, ex, edx);
            ++nE;
            ymin = min(ymin, ey0);
            ymax = max(ymax, ey1);
        }
        x0 = x1;
        y0 = y1;
    }
    if (stop) return true;
    if (nE < 2) return false;

    // SPANS (FillEdgeCollection): per row pair up the x-sorted crossings of the active edges.
    for (int yy = ymin; yy < ymax; ++yy) {
        int prev_x = -0x7fffffff - 1, prev_e = -1;
        int remaining = -1;  // active crossings not yet paired (known after the first sweep)
        for (;;) {
            int ax = 0x7fffffff, ae = nE, bx = 0x7fffffff, be = nE, cnt = 0;
            for (int e = 0; e < nE; ++e) {
                int ey0, ey1, ex, edx;
                E.get(e, ey0, ey1, ex, edx);
                if (yy < ey0 || yy >= ey1) continue;
                ++cnt;
                const int xe = ex + (yy - ey0) * edx;
                const bool after = xe > prev_x || (xe == prev_x && e > prev_e);
                if (!after) continue;
                if (xe < ax || (xe == ax && e < ae)) {
                    bx = ax; be = ae; ax = xe; ae = e;
                } else if (xe < bx || (xe == bx && e < be)) {
                    bx = xe; be = e;
                }
            }
            if (remaining < 0) remaining = cnt;
            if (ae == nE || be == nE) break;
            const int x1s = (ax + 65535) >> 16, x2s = bx >> 16;
            if (x1s <= x2s && sink.emit(yy, x1s, x2s)) return true;
            remaining -= 2;
            if (remaining < 2) break;
            prev_x = bx;
            prev_e = be;
        }
    }
    return false;
}

// Sink that tests runs against the lethal bitmap: pose_collides (envs/base/env.py:464-489).
template <typename WordPtr>
struct CollisionSink {
    WordPtr words;
    int rows, cols, wpr;
    int pxl, pyl;      // world_to_pixel of the robot origin (coordinate_transformations.py:185-205)
    int col_off, row_off;
    __device__ __forceinline__ void begin(int hx, int hy)
    {
        col_off = pxl - hx;  // map col = kx + px - W//2   (env.py:480, W//2 == hx)
        row_off = pyl - hy;
    }
    __device__ __forceinline__ bool emit(int y, int xa, int xb) const
    {
        return interval_hits(words, rows, cols, wpr, y + row_off, xa + col_off, xb + col_off);
    }
};

template <typename WordPtr>
__device__ __forceinline__ bool pose_collides(const DevParams& P, double x, double y, double th, const MapXform& X,
                                              WordPtr words, int rows, int cols, int wpr, const EdgeLds& E)
{
    CollisionSink<WordPtr> sink;
    sink.words = words;
    sink.rows = rows;
    sink.cols = cols;
    sink.wpr = wpr;
    sink.pxl = (int)rint((x - X.ox) * X.inv_res);
    sink.pyl = (int)rint((y - X.oy) * X.inv_res);
    return raster_footprint(P, th, E, sink);
}

}  // namespace bcp
