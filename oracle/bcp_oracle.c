/*
 * bcp_oracle.c -- CPU ORACLE (test infrastructure, NOT product code; see bcp_oracle.h).
 *
 * Plain-C float64 restatement of bc-gym-planning-env's PlanEnv.step() path.  Every function cites the
 * reference file:line it follows (paths relative to /root/reference/bc_gym_planning_env/).
 * Compile with -ffp-contract=off: the only fused multiply-adds are the explicit fma() calls in
 * bco_footprint_vertices (see there).
 */
#include "bcp_oracle.h"

#include <limits.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------------------ */
/* numpy float `%` (npy_divmod): result takes the sign of the divisor */
static double py_mod(double a, double b)
{
    double r = fmod(a, b);
    if (r != 0.0) {
        if ((b < 0.0) != (r < 0.0)) r += b;
    } else {
        r = copysign(0.0, b);
    }
    return r;
}

/* utilities/coordinate_transformations.py:28-36 : (z + pi) % (2 pi) - pi */
double bco_normalize_angle(double z) { return py_mod(z + M_PI, 2.0 * M_PI) - M_PI; }

/* np.clip(a, lo, hi) == minimum(maximum(a, lo), hi) */
static double clipd(double a, double lo, double hi)
{
    double t = a < lo ? lo : a;
    return t > hi ? hi : t;
}

static double signd(double a) { return (a > 0.0) - (a < 0.0); }

/* utilities/coordinate_transformations.py:185-205: np.round((xy - origin) * (1./res)).astype(int) */
void bco_world_to_pixel(const double *xy, int64_t n, const double origin[2], double resolution, int64_t *out)
{
    double anti = 1.0 / resolution;
    for (int64_t i = 0; i < n; ++i) {
        out[2 * i + 0] = (int64_t)rint((xy[2 * i + 0] - origin[0]) * anti);
        out[2 * i + 1] = (int64_t)rint((xy[2 * i + 1] - origin[1]) * anti);
    }
}

/* utilities/path_tools.py:298-323, two-row case as called from tricycle_model.py:520-529 */
int bco_path_velocity(const double p0[3], const double p1[3], double dt, double *v, double *w)
{
    double dx = p1[0] - p0[0], dy = p1[1] - p0[1];
    double c0 = cos(p0[2]), s0 = sin(p0[2]);
    double sign = signd(c0 * dx + s0 * dy);               /* :311 */
    if (sign == 0.0) sign = signd(s0 * dy);               /* :312 */
    double ds = sqrt(dx * dx + dy * dy) * sign;           /* np.linalg.norm(axis=1) :314 */
    double da = p1[2] - p0[2];
    if (da < -M_PI) da += 2.0 * M_PI;                     /* :316 */
    if (da > M_PI) da -= 2.0 * M_PI;                      /* :317 */
    *v = ds / dt;
    *w = da / dt;
    return (fabs(da) < M_PI) ? BCO_ERR_NONE : BCO_ERR_ANGLE_JUMP; /* :319-322 */
}

/* robot_models/differential_drive.py:21-40 (np.sinc(t) = sin(pi t)/(pi t), t==0 -> 1e-20) */
void bco_kinematic_step(const double pose[3], double v, double w, double dt, double out[3])
{
    double angle = pose[2];
    double half_wdt = 0.5 * w * dt;
    double t = half_wdt / M_PI;
    double yy = M_PI * (t == 0.0 ? 1.0e-20 : t);
    double sinc = sin(yy) / yy;
    double v_factor = v * dt * sinc;
    out[0] = pose[0] + v_factor * cos(angle + half_wdt);
    out[1] = pose[1] + v_factor * sin(angle + half_wdt);
    out[2] = bco_normalize_angle(angle + w * dt);
}

/* differential_drive.py:43-52: draw only when variance > 0 */
static double gaussian_noise(double variance, double z, int slot, int *drawn)
{
    if (variance > 0.0) {
        *drawn |= 1 << slot;
        return 0.0 + sqrt(variance) * z; /* np.random.normal(0, std) == loc + scale * z */
    }
    return 0.0;
}

/* differential_drive.py:55-74 */
void bco_kinematic_step_noise(const double pose[3], double v, double w, double dt, const double a[6],
                              const double z[3], double out[3], int *drawn)
{
    int d = 0;
    v = v + gaussian_noise(a[0] * (v * v) + a[1] * (w * w), z[0], 0, &d);
    w = w + gaussian_noise(a[2] * (v * v) + a[3] * (w * w), z[1], 1, &d);
    double final_rot = gaussian_noise(a[4] * (v * v) + a[5] * (w * w), z[2], 2, &d);
    bco_kinematic_step(pose, v, w, dt, out);
    out[2] = bco_normalize_angle(out[2] + final_rot * dt);
    if (drawn) *drawn = d;
}

/* robot_models/tricycle_model.py:127-154 (clip_first = False branch) */
double bco_front_wheel_column_step(double cur, double desired, double max_angle, double max_speed,
                                   double p_gain, double dt)
{
    double max_delta = max_speed * dt;
    double delta = p_gain * (desired - cur);
    delta = clipd(delta, -max_delta, max_delta);
    double na = cur + delta;
    return clipd(na, -max_angle, max_angle);
}

/* robot_models/tricycle_model.py:157-188 */
void bco_velocity_dynamic_model_step(double cur_v, double cur_w, double wheel_angle, double desired_wheel_v,
                                     double L, double max_lin_acc, double max_ang_acc, double dt,
                                     double *new_v, double *new_w)
{
    double des_v = desired_wheel_v * cos(wheel_angle);
    double des_w = desired_wheel_v * sin(wheel_angle) / L;
    double acc_v = (des_v - cur_v) / dt;
    double acc_w = (des_w - cur_w) / dt;
    double lin = clipd(acc_v, -2 * max_lin_acc, max_lin_acc);
    double ang = clipd(acc_w, -max_ang_acc, max_ang_acc);
    double nv = cur_v + lin * dt;
    double nw = cur_w + ang * dt;
    if (0.0 > nv) nv = 0.0; /* python max(new_v, 0.0) :186 */
    *new_v = nv;
    *new_w = nw;
}

/* TricycleRobot.step tricycle_model.py:478-538 ; DiffDriveRobot.step differential_drive.py:236-265 */
int bco_robot_step(const bco_params *p, double st[7], const double cmd[2], const double z[3], int *drawn)
{
    static const double z0[3] = {0, 0, 0};
    if (!z) z = z0;
    int d = 0, err;
    double last[3] = {st[0], st[1], st[2]}, np_[3], mv, mw;
    if (p->model == BCO_MODEL_TRICYCLE) {
        double wa = st[6], new_wa, nv, nw;
        if (p->model_front_column_pid)
            new_wa = bco_front_wheel_column_step(wa, cmd[1], p->max_front_wheel_angle, p->max_front_wheel_speed,
                                                 p->front_column_p_gain, p->dt);
        else
            new_wa = clipd(cmd[1], -p->max_front_wheel_angle, p->max_front_wheel_angle);
        if (p->dynamic_model) {
            bco_velocity_dynamic_model_step(st[3], st[4], new_wa, cmd[0], p->front_wheel_from_axis,
                                            p->max_linear_acceleration, p->max_angular_acceleration, p->dt, &nv, &nw);
            if (p->noise_on)
                bco_kinematic_step_noise(last, nv, nw, p->dt, p->alpha, z, np_, &d);
            else
                bco_kinematic_step(last, nv, nw, p->dt, np_);
        } else { /* tricycle_kinematic_step :38-68 (never noisy) */
            nv = cmd[0] * cos(new_wa);
            nw = cmd[0] * sin(new_wa) / p->front_wheel_from_axis;
            bco_kinematic_step(last, nv, nw, p->dt, np_);
        }
        err = bco_path_velocity(last, np_, p->dt, &mv, &mw);
        st[5] = wa - cmd[1]; /* steering_motor_command :532 */
        st[6] = new_wa;
    } else {
        if (p->noise_on)
            bco_kinematic_step_noise(last, cmd[0], cmd[1], p->dt, p->alpha, z, np_, &d);
        else
            bco_kinematic_step(last, cmd[0], cmd[1], p->dt, np_);
        err = bco_path_velocity(last, np_, p->dt, &mv, &mw);
    }
    st[0] = np_[0];
    st[1] = np_[1];
    st[2] = np_[2];
    st[3] = mv;
    st[4] = mw;
    if (drawn) *drawn = d;
    return err;
}

/* ------------------------------------------------------------------------------------ */
/* utilities/path_tools.py:140-150.  np.dot(fp/res, m) is executed by numpy's bundled OpenBLAS ddot
 * whose 2-element tail is FMA-contracted: dot = fma(a1, b1, a0*b0) (verified against numpy 2.2.6 /
 * OpenBLAS 0.3.29 in the build container: 8000/8000 vertices match this form, 54% match plain
 * mul-add).  m = [[c, -s], [s, c]] applied to column vectors => rotation by +angle. */
void bco_footprint_vertices(double angle, const double *verts, int k, double res, int32_t *out_xy, int32_t half[2])
{
    double c = cos(angle), s = sin(angle), ns = -s;
    double mx = -INFINITY, my = -INFINITY, nx = INFINITY, ny = INFINITY;
    double px[BCO_MAX_VERTS], py[BCO_MAX_VERTS];
    for (int i = 0; i < k; ++i) {
        double qx = verts[2 * i] / res, qy = verts[2 * i + 1] / res;
        px[i] = fma(qy, ns, qx * c);
        py[i] = fma(qy, c, qx * s);
        if (px[i] > mx) mx = px[i];
        if (px[i] < nx) nx = px[i];
        if (py[i] > my) my = py[i];
        if (py[i] < ny) ny = py[i];
    }
    double cx = mx > -nx ? mx : -nx, cy = my > -ny ? my : -ny; /* np.maximum(amax, -amin) :147 */
    half[0] = (int32_t)ceil(cx);
    half[1] = (int32_t)ceil(cy);
    for (int i = 0; i < k; ++i) {
        out_xy[2 * i] = (int32_t)rint(px[i]) + half[0];
        out_xy[2 * i + 1] = (int32_t)rint(py[i]) + half[1];
    }
}

/* --- OpenCV drawing restatement (opencv imgproc drawing.cpp: LineIterator / Line / CollectPolyEdges /
 * FillEdgeCollection), integer coordinates, shift 0, connectivity 8, single-channel uint8. ------------ */
#define XY_SHIFT 16
#define XY_ONE (1 << XY_SHIFT)

/* cv::clipLine on int64 points */
static int clip_line(int64_t width, int64_t height, int64_t *x1, int64_t *y1, int64_t *x2, int64_t *y2)
{
    int c1, c2;
    int64_t right = width - 1, bottom = height - 1;
    if (width <= 0 || height <= 0) return 0;
    c1 = (*x1 < 0) + (*x1 > right) * 2 + (*y1 < 0) * 4 + (*y1 > bottom) * 8;
    c2 = (*x2 < 0) + (*x2 > right) * 2 + (*y2 < 0) * 4 + (*y2 > bottom) * 8;
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
        int64_t a;
        if (c1 & 12) {
            a = c1 < 8 ? 0 : bottom;
            *x1 += (int64_t)((double)(a - *y1) * (double)(*x2 - *x1) / (double)(*y2 - *y1));
            *y1 = a;
            c1 = (*x1 < 0) + (*x1 > right) * 2;
        }
        if (c2 & 12) {
            a = c2 < 8 ? 0 : bottom;
            *x2 += (int64_t)((double)(a - *y2) * (double)(*x2 - *x1) / (double)(*y2 - *y1));
            *y2 = a;
            c2 = (*x2 < 0) + (*x2 > right) * 2;
        }
        if ((c1 & c2) == 0 && (c1 | c2) != 0) {
            if (c1) {
                a = c1 == 1 ? 0 : right;
                *y1 += (int64_t)((double)(a - *x1) * (double)(*y2 - *y1) / (double)(*x2 - *x1));
                *x1 = a;
                c1 = 0;
            }
            if (c2) {
                a = c2 == 1 ? 0 : right;
                *y2 += (int64_t)((double)(a - *x2) * (double)(*y2 - *y1) / (double)(*x2 - *x1));
                *x2 = a;
                c2 = 0;
            }
        }
    }
    return (c1 | c2) == 0;
}

/* cv::LineIterator(img, pt1, pt2, 8, leftToRight=true) + Line(): 8-connected Bresenham */
static void line8(uint8_t *img, int rows, int cols, int64_t x1, int64_t y1, int64_t x2, int64_t y2, uint8_t value)
{
    if ((uint64_t)x1 >= (uint64_t)cols || (uint64_t)x2 >= (uint64_t)cols || (uint64_t)y1 >= (uint64_t)rows ||
        (uint64_t)y2 >= (uint64_t)rows) {
        if (!clip_line(cols, rows, &x1, &y1, &x2, &y2)) return;
    }
    int64_t dx = x2 - x1, dy = y2 - y1;
    int64_t step_x = 1, step_y = 1;
    if (dx < 0) { /* left to right: start from the point with the smaller x */
        dx = -dx;
        dy = -dy;
        x1 = x2;
        y1 = y2;
    }
    if (dy < 0) {
        dy = -dy;
        step_y = -1;
    }
    int vert = dy > dx;
    int64_t major = vert ? dy : dx, minor = vert ? dx : dy;
    int64_t err = major - (minor + minor);
    int64_t plus_delta = major + major, minus_delta = -(minor + minor);
    int64_t count = major + 1;
    int64_t x = x1, y = y1;
    for (int64_t i = 0; i < count; ++i) {
        img[y * cols + x] = value;
        int mask = err < 0;
        err += minus_delta + (mask ? plus_delta : 0);
        if (vert) {
            y += step_y;
            if (mask) x += step_x;
        } else {
            x += step_x;
            if (mask) y += step_y;
        }
    }
}

void bco_line(uint8_t *img, int rows, int cols, int64_t x0, int64_t y0, int64_t x1, int64_t y1, uint8_t value)
{
    line8(img, rows, cols, x0, y0, x1, y1, value);
}

typedef struct {
    int y0, y1;
    int64_t x, dx;
    int next; /* index into edge array, -1 = null */
} poly_edge;

static int cmp_edges(const void *a, const void *b)
{
    const poly_edge *e1 = (const poly_edge *)a, *e2 = (const poly_edge *)b;
    if (e1->y0 != e2->y0) return e1->y0 < e2->y0 ? -1 : 1;
    if (e1->x != e2->x) return e1->x < e2->x ? -1 : 1;
    if (e1->dx != e2->dx) return e1->dx < e2->dx ? -1 : 1;
    return 0;
}

static void hline(uint8_t *row, int x1, int x2, uint8_t value)
{
    for (int x = x1; x <= x2; ++x) row[x] = value;
}

/* cv::fillPoly for one contour = CollectPolyEdges (draws the outline with Line, builds the edge table)
 * followed by FillEdgeCollection (even-odd scanline over 16.16 fixed-point edge x). */
void bco_fill_poly(uint8_t *img, int rows, int cols, const int32_t *pts, int count, uint8_t value)
{
    poly_edge edges[BCO_MAX_VERTS * 2 + 2];
    int total = 0;
    if (count <= 0) return;
    /* CollectPolyEdges, shift = 0, offset = 0 */
    int64_t p0x = (int64_t)pts[2 * (count - 1)] << XY_SHIFT, p0y = pts[2 * (count - 1) + 1];
    for (int i = 0; i < count; ++i) {
        int64_t p1x = (int64_t)pts[2 * i] << XY_SHIFT, p1y = pts[2 * i + 1];
        int64_t t0x = (p0x + (XY_ONE >> 1)) >> XY_SHIFT, t1x = (p1x + (XY_ONE >> 1)) >> XY_SHIFT;
        line8(img, rows, cols, t0x, p0y, t1x, p1y, value);
        if (p0y != p1y) {
            poly_edge e;
            if (p0y < p1y) {
                e.y0 = (int)p0y;
                e.y1 = (int)p1y;
                e.x = p0x;
            } else {
                e.y0 = (int)p1y;
                e.y1 = (int)p0y;
                e.x = p1x;
            }
            e.dx = (p1x - p0x) / (p1y - p0y); /* C division truncates toward zero */
            e.next = -1;
            edges[total++] = e;
        }
        p0x = p1x;
        p0y = p1y;
    }
    /* FillEdgeCollection */
    if (total < 2) return;
    int y_max = INT_MIN, y_min = INT_MAX;
    int64_t x_max = -1, x_min = INT64_MAX;
    for (int i = 0; i < total; ++i) {
        poly_edge *e1 = &edges[i];
        int64_t x1 = e1->x + (int64_t)(e1->y1 - e1->y0) * e1->dx;
        if (e1->y0 < y_min) y_min = e1->y0;
        if (e1->y1 > y_max) y_max = e1->y1;
        if (e1->x < x_min) x_min = e1->x;
        if (e1->x > x_max) x_max = e1->x;
        if (x1 < x_min) x_min = x1;
        if (x1 > x_max) x_max = x1;
    }
    if (y_max < 0 || y_min >= rows || x_max < 0 || x_min >= ((int64_t)cols << XY_SHIFT)) return;
    qsort(edges, (size_t)total, sizeof(poly_edge), cmp_edges);
    /* sentinel */
    edges[total].y0 = INT_MAX;
    edges[total].next = -1;
    /* "tmp" head node is index total+1 */
    const int HEAD = total + 1;
    edges[HEAD].next = -1;
    int i = 0;
    poly_edge *e = &edges[0];
    if (y_max > rows) y_max = rows;
    for (int y = e->y0; y < y_max; ++y) {
        int last, prelast, keep_prelast;
        int sort_flag = 0, draw = 0, clipline = y < 0;
        prelast = HEAD;
        last = edges[HEAD].next;
        while (last >= 0 || e->y0 == y) {
            if (last >= 0 && edges[last].y1 == y) {
                edges[prelast].next = edges[last].next;
                last = edges[last].next;
                continue;
            }
            keep_prelast = prelast;
            if (last >= 0 && (e->y0 > y || edges[last].x < e->x)) {
                prelast = last;
                last = edges[last].next;
            } else if (i < total) {
                edges[prelast].next = i;
                e->next = last;
                prelast = i;
                e = &edges[++i];
            } else
                break;
            if (draw) {
                if (!clipline) {
                    uint8_t *timg = img + (size_t)y * cols;
                    int x1, x2;
                    if (edges[keep_prelast].x > edges[prelast].x) {
                        x1 = (int)((edges[prelast].x + XY_ONE - 1) >> XY_SHIFT);
                        x2 = (int)(edges[keep_prelast].x >> XY_SHIFT);
                    } else {
                        x1 = (int)((edges[keep_prelast].x + XY_ONE - 1) >> XY_SHIFT);
                        x2 = (int)(edges[prelast].x >> XY_SHIFT);
                    }
                    if (x1 < cols && x2 >= 0) {
                        if (x1 < 0) x1 = 0;
                        if (x2 >= cols) x2 = cols - 1;
                        hline(timg, x1, x2, value);
                    }
                }
                edges[keep_prelast].x += edges[keep_prelast].dx;
                edges[prelast].x += edges[prelast].dx;
            }
            draw ^= 1;
        }
        /* bubble sort the active list by x */
        keep_prelast = -1;
        do {
            prelast = HEAD;
            last = edges[HEAD].next;
            while (last != keep_prelast && last >= 0 && edges[last].next >= 0) {
                int te = edges[last].next;
                if (edges[last].x > edges[te].x) {
                    edges[prelast].next = te;
                    edges[last].next = edges[te].next;
                    edges[te].next = last;
                    prelast = te;
                    sort_flag = 1;
                } else {
                    prelast = last;
                    last = te;
                }
            }
            keep_prelast = prelast;
        } while (sort_flag && keep_prelast != edges[HEAD].next && keep_prelast != HEAD);
    }
}

/* utilities/path_tools.py:122-162 */
void bco_pixel_footprint(double angle, const double *verts, int k, double res, uint8_t *out, int cap, int *h, int *w)
{
    int32_t v[BCO_MAX_VERTS * 2], half[2];
    bco_footprint_vertices(angle, verts, k, res, v, half);
    *h = 2 * half[1] + 1;
    *w = 2 * half[0] + 1;
    if ((*h) * (*w) > cap) return;
    memset(out, 0, (size_t)(*h) * (*w));
    bco_fill_poly(out, *h, *w, v, k, 255);
}

/* envs/base/env.py:464-489 */
int bco_pose_collides(double x, double y, double angle, const double *verts, int k, const uint8_t *map, int rows,
                      int cols, const double origin[2], double res)
{
    enum { CAP = 512 * 512 };
    static __thread uint8_t *buf = NULL;
    if (!buf) buf = (uint8_t *)malloc(CAP);
    int h, w;
    bco_pixel_footprint(angle, verts, k, res, buf, CAP, &h, &w);
    if (h * w > CAP) return -1;
    double xy[2] = {x, y};
    int64_t pix[2];
    bco_world_to_pixel(xy, 1, origin, res, pix);
    for (int ky = 0; ky < h; ++ky) {
        int64_t r = pix[1] + ky - h / 2;
        if (r < 0 || r >= rows) continue;
        for (int kx = 0; kx < w; ++kx) {
            if (!buf[ky * w + kx]) continue;
            int64_t c = pix[0] + kx - w / 2;
            if (c < 0 || c >= cols) continue;
            if (map[r * cols + c] == BCO_LETHAL) return 1;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------ */
/* utilities/path_tools.py:397-448 */
int bco_find_last_reached(const double pose[3], const double *path, int m, double sp, double ap)
{
    double thr = -sp / 9; /* :423 */
    int last = -1;
    for (int j = 0; j < m; ++j) {
        const double *s = path + 3 * j;
        double dist = hypot(s[0] - pose[0], s[1] - pose[1]);                           /* :424 */
        double ang = fabs(bco_normalize_angle(pose[2] - s[2]));                        /* :425 */
        double par = cos(s[2]) * (pose[0] - s[0]) + sin(s[2]) * (pose[1] - s[1]);      /* :405 */
        if (dist < sp && ang < ap && par >= thr) last = j;
    }
    return last;
}

/* envs/base/reward.py:214-259 */
double bco_reward(const bco_params *p, const double pose[3], const double *path, int m, double *min_dist,
                  int32_t *target_idx)
{
    if (*target_idx > m - 1) return 0.0; /* done() :66-69 */
    int last = bco_find_last_reached(pose, path, m, p->spatial_precision, p->angular_precision);
    if (last >= 0 && last >= *target_idx) {
        *target_idx = last + 1;
        if (!(*target_idx > m - 1)) {
            const double *g = path + 3 * (*target_idx);
            *min_dist = hypot(g[0] - pose[0], g[1] - pose[1]);
        } else
            *min_dist = 0.0;
        return 1.0;
    }
    const double *g = path + 3 * (*target_idx);
    double d = hypot(g[0] - pose[0], g[1] - pose[1]);
    if (d < *min_dist) {
        double r = *min_dist - d;
        *min_dist = d;
        return r * p->spatial_progress_multiplier;
    }
    return 0.0;
}

/* envs/base/reward.py:261-288 */
int bco_initial_reward_state(const double *path, int m, double sp, double ap, double *min_dist, int32_t *target_idx)
{
    int last = bco_find_last_reached(path, path, m, sp, ap);
    if (last == m - 1) return -1;
    int t = last + 1; /* None + 1 cannot happen: path[0] always reaches itself */
    *target_idx = t;
    *min_dist = hypot(path[3 * t] - path[0], path[3 * t + 1] - path[1]);
    return 0;
}

/* envs/base/env.py:334-361 (+ :363-398, :442-461, :400-419), all delays 0 */
int bco_env_step(const bco_params *p, double st[7], double *min_dist, int32_t *target_idx, int32_t *cur_iter,
                 double *cur_time, uint8_t *collided_sticky, const double cmd[2], const double z[3],
                 const uint8_t *map, int rows, int cols, const double origin[2], double res, const double *path, int m,
                 double *reward, uint8_t *done, uint8_t *collided_now, int *drawn)
{
    double old[3] = {st[0], st[1], st[2]};
    int err = bco_robot_step(p, st, cmd, z, drawn);
    int col = bco_pose_collides(st[0], st[1], st[2], &p->verts[0][0], p->n_verts, map, rows, cols, origin, res);
    if (col) { /* robot.set_pose(*old): zeroes v, w; keeps wheel_angle / steering command */
        st[0] = old[0];
        st[1] = old[1];
        st[2] = old[2];
        st[3] = 0.0;
        st[4] = 0.0;
    }
    *cur_time = *cur_time + p->dt;
    *cur_iter = *cur_iter + 1;
    *collided_sticky = (uint8_t)(*collided_sticky || col);
    *reward = bco_reward(p, st, path, m, min_dist, target_idx);
    int goal = *target_idx > m - 1;
    int timed_out = *cur_iter >= p->iteration_timeout;
    *done = (uint8_t)(goal || timed_out || *collided_sticky);
    if (collided_now) *collided_now = (uint8_t)col;
    return err;
}

/* envs/base/reward.py:125-139, 330-353 */
double bco_reward_pure_pursuit(const double pose[3], const double *path, int m, int collided, double *min_dist,
                               int32_t *target_idx)
{
    /* update_goal(pose, radius=2.): first way point from target_idx on that is more than 2 m away.
     * np.linalg.norm of a 2-vector is sqrt(dot(x, x)); numpy's dot runs through OpenBLAS ddot, whose two-element
     * tail is fma-contracted (same observation as for the footprint rotation, bco_footprint_vertices). */
    int found = m - 1;
    for (int i = *target_idx; i < m; ++i) {
        double dx = path[3 * i] - pose[0], dy = path[3 * i + 1] - pose[1];
        if (sqrt(fma(dy, dy, dx * dx)) > 2.) {
            found = i;
            break;
        }
    }
    *target_idx = found;
    double reward = -0.05;
    const double *g = path + 3 * (m - 1); /* current_goal_pose() is the LAST way point (:115-120) */
    double dist = hypot(g[0] - pose[0], g[1] - pose[1]);
    reward += *min_dist - dist;
    *min_dist = dist;
    if (collided) reward -= 100;
    return reward;
}

/* envs/base/reward.py:355-371 */
void bco_initial_pure_pursuit_state(const double *path, int m, double *min_dist, int32_t *target_idx)
{
    const double *g = path + 3 * (m - 1);
    *target_idx = 1;
    *min_dist = hypot(g[0] - path[0], g[1] - path[1]);
}

/* envs/synth_turn_env.py:412-420 */
void bco_goal_direction_state(const double pose[3], const double last_waypoint[3], const double world_size[2],
                              const double ego_state[3], double out[5])
{
    double c = cos(pose[2]), s = sin(pose[2]);
    double tx = -pose[0] * c - pose[1] * s;
    double ty = pose[0] * s - pose[1] * c;
    double tt = bco_normalize_angle(-pose[2]);
    double ct = cos(tt), st = sin(tt);
    double ex = ct * last_waypoint[0] + (-st) * last_waypoint[1] + tx;
    double ey = st * last_waypoint[0] + ct * last_waypoint[1] + ty;
    double gx = ex / world_size[0], gy = ey / world_size[1];
    double norm = sqrt(fma(gy, gy, gx * gx)); /* np.linalg.norm: sqrt of a 2-term dot (fma-contracted ddot tail) */
    out[0] = gx / norm;
    out[1] = gy / norm;
    out[2] = ego_state[0];
    out[3] = ego_state[1];
    out[4] = ego_state[2];
}

/* _get_element_from_list_with_delay (env.py:27-49) for the k-th push since reset */
static void fifo_delay(double *q, int width, int delay, int k, const double *elem, double *out)
{
    if (delay <= 0) {
        for (int c = 0; c < width; ++c) out[c] = elem[c];
        return;
    }
    double first[8];
    int slot = (k - 1) % delay;
    if (k <= delay) { /* the list is not longer than `delay` yet: append, hand back element 1 */
        for (int c = 0; c < width; ++c) q[slot * width + c] = elem[c];
        for (int c = 0; c < width; ++c) out[c] = q[c];
    } else { /* pop(0): element k - delay, whose slot the new element takes */
        for (int c = 0; c < width; ++c) first[c] = q[slot * width + c];
        for (int c = 0; c < width; ++c) q[slot * width + c] = elem[c];
        for (int c = 0; c < width; ++c) out[c] = first[c];
    }
}

/* envs/base/env.py:334-361 + :363-398 with delays and either reward provider */
int bco_env_step_ex(const bco_params *p, double st[7], bco_delay_state *d, double *min_dist, int32_t *target_idx,
                    int32_t *cur_iter, double *cur_time, uint8_t *collided_sticky, const double cmd[2], const double z[3],
                    const uint8_t *map, int rows, int cols, const double origin[2], double res, const double *path, int m,
                    double *reward, uint8_t *done, uint8_t *collided_now, int *drawn)
{
    const int k = *cur_iter + 1;
    double action[2], pose[3], seen_pose[3], seen_state[7];
    fifo_delay(d ? d->control_q : NULL, 2, p->control_delay, k, cmd, action); /* :371 */
    double old[3] = {st[0], st[1], st[2]};
    int err = bco_robot_step(p, st, action, z, drawn);
    int col = bco_pose_collides(st[0], st[1], st[2], &p->verts[0][0], p->n_verts, map, rows, cols, origin, res);
    if (col) {
        st[0] = old[0];
        st[1] = old[1];
        st[2] = old[2];
        st[3] = 0.0;
        st[4] = 0.0;
    }
    pose[0] = st[0];
    pose[1] = st[1];
    pose[2] = st[2];
    fifo_delay(d ? d->pose_q : NULL, 3, p->pose_delay, k, pose, seen_pose);  /* :377-380 */
    *cur_time = *cur_time + p->dt;
    *cur_iter = *cur_iter + 1;
    fifo_delay(d ? d->state_q : NULL, 7, p->state_delay, k, st, seen_state); /* :385-389 */
    *collided_sticky = (uint8_t)(*collided_sticky || col);
    if (d && d->obs_pose)
        for (int c = 0; c < 3; ++c) d->obs_pose[c] = seen_pose[c];
    if (d && d->obs_state)
        for (int c = 0; c < 7; ++c) d->obs_state[c] = seen_state[c];
    int goal;
    if (p->reward_provider == BCO_REWARD_PURE_PURSUIT) {
        *reward = bco_reward_pure_pursuit(seen_pose, path, m, *collided_sticky, min_dist, target_idx);
        const double *g = path + 3 * (m - 1); /* done(): within 1 m of the last way point (reward.py:141-150) */
        goal = hypot(g[0] - seen_pose[0], g[1] - seen_pose[1]) < 1.0;
    } else {
        *reward = bco_reward(p, seen_pose, path, m, min_dist, target_idx);
        goal = *target_idx > m - 1;
    }
    int timed_out = *cur_iter >= p->iteration_timeout;
    *done = (uint8_t)(goal || timed_out || *collided_sticky);
    if (collided_now) *collided_now = (uint8_t)col;
    return err;
}

/* ---- egocentric observation ---------------------------------------------------------------------------- */
/* cv::getRotationMatrix2D (opencv imgproc imgwarp.cpp) */
void bco_rotation_matrix_2d(double cx, double cy, double angle_deg, double scale, double M[6])
{
    float fx = (float)cx, fy = (float)cy; /* Point2f center */
    double angle = angle_deg * (M_PI / 180);
    double alpha = cos(angle) * scale, beta = sin(angle) * scale;
    M[0] = alpha;
    M[1] = beta;
    M[2] = (1 - alpha) * fx - beta * fy;
    M[3] = -beta;
    M[4] = alpha;
    M[5] = beta * fx + (1 - alpha) * fy;
}

static int sat_int(double v) /* cv::saturate_cast<int>(double) == cvRound: nearest, ties to even */
{
    double r = rint(v);
    if (r >= 2147483647.0) return 2147483647;
    if (r <= -2147483648.0) return (-2147483647 - 1);
    return (int)r;
}

static int sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

/* cv::warpAffine + hal::warpAffine / WarpAffineInvoker + remapNearest, INTER_NEAREST, BORDER_CONSTANT */
void bco_warp_affine_nearest(const uint8_t *src, int rows, int cols, const double M_in[6], uint8_t *dst, int drows,
                             int dcols, uint8_t border)
{
    const int AB_BITS = 10, AB_SCALE = 1 << 10, round_delta = (1 << 10) / 2;
    double M[6];
    for (int k = 0; k < 6; ++k) M[k] = M_in[k];
    { /* dst -> src map */
        double D = M[0] * M[4] - M[1] * M[3];
        D = D != 0 ? 1. / D : 0;
        double A11 = M[4] * D, A22 = M[0] * D;
        M[0] = A11;
        M[1] *= -D;
        M[3] *= -D;
        M[4] = A22;
        double b1 = -M[0] * M[2] - M[1] * M[5];
        double b2 = -M[3] * M[2] - M[4] * M[5];
        M[2] = b1;
        M[5] = b2;
    }
    for (int y = 0; y < drows; ++y) {
        int X0 = sat_int((M[1] * y + M[2]) * AB_SCALE) + round_delta;
        int Y0 = sat_int((M[4] * y + M[5]) * AB_SCALE) + round_delta;
        for (int x = 0; x < dcols; ++x) {
            int adelta = sat_int(M[0] * x * AB_SCALE), bdelta = sat_int(M[3] * x * AB_SCALE);
            int sx = sat_short((X0 + adelta) >> AB_BITS), sy = sat_short((Y0 + bdelta) >> AB_BITS);
            uint8_t v = border;
            if ((unsigned)sx < (unsigned)cols && (unsigned)sy < (unsigned)rows) v = src[(size_t)sy * cols + sx];
            dst[(size_t)y * dcols + x] = v;
        }
    }
}

/* utilities/costmap_utils.py:25-75 */
void bco_extract_egocentric(const uint8_t *map, int rows, int cols, const double origin[2], double resolution,
                            const double pose[3], int has_window, const double window_origin[2],
                            const double window_size[2], uint8_t border, uint8_t *out, int32_t out_shape[2],
                            double *M_used)
{
    int64_t pix[2];
    const double zero[2] = {0.0, 0.0};
    bco_world_to_pixel(pose, 1, origin, resolution, pix); /* :41 */
    double M[6];
    bco_rotation_matrix_2d((double)pix[0], (double)pix[1], 180 * pose[2] / M_PI, 1.0, M); /* :43 */
    int dcols = cols, drows = rows;
    if (has_window) {
        int64_t sz[2], dsp[2];
        bco_world_to_pixel(window_size, 1, zero, resolution, sz); /* :48 */
        dcols = (int)sz[0];
        drows = (int)sz[1];
        double delta[2] = {window_origin[0] - (origin[0] - pose[0]), window_origin[1] - (origin[1] - pose[1])}; /* :53 */
        bco_world_to_pixel(delta, 1, zero, resolution, dsp);
        /* _compose_affine_transforms (:57-62): both operands become float32 3x3 matrices and are multiplied in
         * float32; with the shift matrix [[1,0,-dx],[0,1,-dy],[0,0,1]] on the left the only inexact operation is
         * the final addition of the translation column. */
        float t[6];
        for (int k = 0; k < 6; ++k) t[k] = (float)M[k];
        t[2] = t[2] + (-(float)dsp[0]);
        t[5] = t[5] + (-(float)dsp[1]);
        for (int k = 0; k < 6; ++k) M[k] = (double)t[k];
    }
    out_shape[0] = drows;
    out_shape[1] = dcols;
    if (M_used)
        for (int k = 0; k < 6; ++k) M_used[k] = M[k];
    if (out) bco_warp_affine_nearest(map, rows, cols, M, out, drows, dcols, border);
}

/* utilities/costmap_utils.py:78-104 (center_pixel_coords=None) */
void bco_rotate_costmap(const uint8_t *map, int rows, int cols, double angle, uint8_t border, uint8_t *out)
{
    double deg = -angle * (180.0 / M_PI); /* np.rad2deg(-angle) */
    if (deg != 0.) {
        double M[6];
        bco_rotation_matrix_2d((double)(cols / 2), (double)(rows / 2), deg, 1.0, M);
        bco_warp_affine_nearest(map, rows, cols, M, out, rows, cols, border);
    } else {
        memcpy(out, map, (size_t)rows * cols);
    }
}

/* envs/egocentric.py:140-160 */
void bco_goal_n_state(const double pose[3], const double *next_waypoint, int remaining, const double world_size[2],
                      const double *robot_state, int n_state, float *out)
{
    if (remaining <= 0) {
        for (int k = 0; k < 3 + n_state; ++k) out[k] = 0.0f;
        return;
    }
    /* inverse_transform (coordinate_transformations.py:57-84) */
    double c = cos(pose[2]), s = sin(pose[2]);
    double tx = -pose[0] * c - pose[1] * s;
    double ty = pose[0] * s - pose[1] * c;
    double tt = bco_normalize_angle(-pose[2]);
    /* project_poses (:310-328): homogeneous matrix times (x, y, 1) */
    double ct = cos(tt), st = sin(tt);
    double ex = ct * next_waypoint[0] + (-st) * next_waypoint[1] + tx;
    double ey = st * next_waypoint[0] + ct * next_waypoint[1] + ty;
    double eth = bco_normalize_angle(next_waypoint[2] + tt);
    double gx = clipd(ex / world_size[0], -1., 1.), gy = clipd(ey / world_size[1], -1., 1.);
    out[0] = (float)gx;
    out[1] = (float)gy;
    out[2] = (float)eth;
    for (int k = 0; k < n_state; ++k) out[3 + k] = (float)robot_state[k];
}

/* ------------------------------------------------------------------------------------ */
typedef struct {
    const bco_params *p;
    const bco_batch *b;
    int64_t lo, hi;
} batch_job;

static void *batch_worker(void *arg)
{
    batch_job *job = (batch_job *)arg;
    const bco_params *p = job->p;
    const bco_batch *b = job->b;
    for (int64_t i = job->lo; i < job->hi; ++i) {
        double st[7];
        for (int f = 0; f < 7; ++f) st[f] = b->st[f][i];
        int64_t g = b->geom ? b->geom[i] : i; /* entry of the per-env arrays this env uses */
        const uint8_t *map = b->maps + (size_t)g * (size_t)b->map_stride;
        int rows = b->rows_per_env ? b->rows_per_env[g] : b->rows;
        int cols = b->cols_per_env ? b->cols_per_env[g] : b->cols;
        const double *origin = b->origins + (size_t)g * (size_t)b->origin_stride;
        const double *path = b->paths + (size_t)g * (size_t)b->path_stride;
        int m = b->path_stride ? b->lens[g] : b->lens[0];
        double cmd[2] = {b->actions[2 * i], b->actions[2 * i + 1]};
        const double *z = b->z ? b->z + 3 * i : NULL;
        double cur_time = b->cur_time ? b->cur_time[i] : 0.0;
        uint8_t cn = 0;
        bco_delay_state d = {
            b->control_q ? b->control_q + (size_t)i * p->control_delay * 2 : NULL,
            b->pose_q ? b->pose_q + (size_t)i * p->pose_delay * 3 : NULL,
            b->state_q ? b->state_q + (size_t)i * p->state_delay * 7 : NULL,
            b->obs_pose ? b->obs_pose + (size_t)i * 3 : NULL,
            b->obs_state ? b->obs_state + (size_t)i * 7 : NULL,
        };
        int err = bco_env_step_ex(p, st, &d, &b->min_dist[i], &b->target_idx[i], &b->cur_iter[i], &cur_time,
                                  &b->collided[i], cmd, z, map, rows, cols, origin, b->resolution, path, m, &b->reward[i],
                                  &b->done[i], &cn, NULL);
        if (b->collided_now) b->collided_now[i] = cn;
        if (b->err) b->err[i] = err;
        if (b->auto_reset && b->done[i]) {
            if (b->geom) {
                if (b->next_geom) g = b->next_geom[g];
                b->geom[i] = (int32_t)g;
            }
            for (int f = 0; f < 7; ++f) st[f] = b->init_st[f][g];
            b->min_dist[i] = b->init_min_dist[g];
            b->target_idx[i] = b->init_target_idx[g];
            b->cur_iter[i] = 0;
            b->collided[i] = 0;
            cur_time = 0.0;
            /* the restored State exposes the initial pose / robot state and empty queues (k restarts at 1) */
            if (b->obs_pose)
                for (int c = 0; c < 3; ++c) b->obs_pose[(size_t)i * 3 + c] = st[c];
            if (b->obs_state)
                for (int c = 0; c < 7; ++c) b->obs_state[(size_t)i * 7 + c] = st[c];
        }
        if (b->cur_time) b->cur_time[i] = cur_time;
        for (int f = 0; f < 7; ++f) b->st[f][i] = st[f];
    }
    return NULL;
}

int bco_step_batch(const bco_params *p, const bco_batch *b, int threads)
{
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    if (threads == 1) {
        batch_job job = {p, b, 0, b->n};
        batch_worker(&job);
        return 0;
    }
    pthread_t tid[256];
    batch_job jobs[256];
    int64_t chunk = (b->n + threads - 1) / threads;
    int started = 0;
    for (int t = 0; t < threads; ++t) {
        int64_t lo = t * chunk, hi = lo + chunk;
        if (lo >= b->n) break;
        if (hi > b->n) hi = b->n;
        jobs[t].p = p;
        jobs[t].b = b;
        jobs[t].lo = lo;
        jobs[t].hi = hi;
        if (pthread_create(&tid[t], NULL, batch_worker, &jobs[t]) != 0) return -1;
        ++started;
    }
    for (int t = 0; t < started; ++t) pthread_join(tid[t], NULL);
    return 0;
}

/* Many steps back to back (bench.py's cpu_baseline): envs are independent, so every thread takes a block of envs
 * through ALL the steps on its own -- no thread is created or joined per step.  Step k reads actions_pool / z_pool
 * batch k % pool_len ([pool_len][n][2] / [pool_len][n][3], z_pool may be NULL). */
typedef struct {
    const bco_params *p;
    bco_batch b;
    int64_t lo, hi;
    int steps, pool_len;
    const double *actions_pool, *z_pool;
} run_job;

static void *run_worker(void *arg)
{
    run_job *j = (run_job *)arg;
    for (int s = 0; s < j->steps; ++s) {
        const int k = s % j->pool_len;
        j->b.actions = j->actions_pool + (int64_t)k * j->b.n * 2;
        j->b.z = j->z_pool ? j->z_pool + (int64_t)k * j->b.n * 3 : NULL;
        batch_job job = {j->p, &j->b, j->lo, j->hi};
        batch_worker(&job);
    }
    return NULL;
}

int bco_run_steps(const bco_params *p, const bco_batch *b, int threads, int steps, const double *actions_pool,
                  const double *z_pool, int pool_len)
{
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    if (steps < 0 || pool_len < 1 || !actions_pool) return -1;
    static run_job jobs[256];
    pthread_t tid[256];
    int64_t chunk = (b->n + threads - 1) / threads;
    int started = 0;
    for (int t = 0; t < threads; ++t) {
        int64_t lo = t * chunk, hi = lo + chunk;
        if (lo >= b->n) break;
        if (hi > b->n) hi = b->n;
        jobs[t].p = p;
        jobs[t].b = *b;
        jobs[t].lo = lo;
        jobs[t].hi = hi;
        jobs[t].steps = steps;
        jobs[t].pool_len = pool_len;
        jobs[t].actions_pool = actions_pool;
        jobs[t].z_pool = z_pool;
        if (threads == 1) {
            run_worker(&jobs[t]);
            return 0;
        }
        if (pthread_create(&tid[t], NULL, run_worker, &jobs[t]) != 0) return -1;
        ++started;
    }
    for (int t = 0; t < started; ++t) pthread_join(tid[t], NULL);
    return 0;
}

