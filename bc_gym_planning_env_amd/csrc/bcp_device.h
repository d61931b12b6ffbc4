// bcp_device.h -- device-side arithmetic of the PlanEnv.step() path (gfx950, fp64, no FMA contraction).
//
// Every function states the reference lines it implements (paths relative to bc_gym_planning_env/).
// The translation unit is compiled with -ffp-contract=off: products and sums round separately exactly as
// numpy's ufuncs do; the only fused operations are the explicit fma() calls in footprint_vertex().
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bcplan.h"

namespace bcp {

// Pointers that reach a kernel through memory (the device-resident parameter block, the kernel-argument segment read
// through a constant-address-space reference) are generic to the compiler, which then emits FLAT loads and stores: they
// count against both the vector-memory and the LDS counters, so an LDS operation behind a few flat stores waits for
// memory.  Everything these kernels reach through such pointers is global memory: say so.
template <typename T>
using GlobalPtr = T __attribute__((address_space(1)))*;

template <typename T>
__device__ __forceinline__ GlobalPtr<T> as_global(T* p)
{
    return (GlobalPtr<T>)p;
}

constexpr double kPi = 3.14159265358979323846;
constexpr double kTwoPi = 2.0 * kPi;

// Device copy of the parameters (kernel argument, lives in SGPRs / scalar cache).
struct DevParams {
    int32_t model, n_verts, dynamic_model, model_front_column_pid, noise_on, iteration_timeout;
    double dt, L, max_wheel_angle, max_wheel_speed, max_lin_acc, max_ang_acc, p_gain;
    double inv_dt, inv_L;    // 1 / dt, 1 / L, correctly rounded (host): div_by_const; directly behind p_gain (step_local_kernel
                             // fetches dt .. inv_L as nine adjacent values)
    double alpha[6];
    double sp, ap, progress_mult;
    double par_thr;          // -sp / 9, utilities/path_tools.py:423
    double sp2_lo, sp2_hi;   // sp^2 (1 -+ 1e-13): dx^2+dy^2 outside this band decides hypot(dx,dy) < sp on its own
    double sp_prune;         // sp nudged up two ulps: |dx| > sp_prune  =>  hypot(dx,dy) >= sp for any faithful hypot
    double qverts[BCP_MAX_VERTS][2];  // footprint / resolution (path_tools.py:145), divided on the host in fp64
    float qbox[4];                    // bounding box of qverts in the robot frame: xmin, xmax, ymin, ymax (pixels)
    int32_t reward_provider;          // BCP_REWARD_*
    int32_t control_delay, pose_delay, state_delay;   // EnvParams delays (envs/base/params.py:28-30)
    float ap_cos_min;                 // cos(ap) - 1e-4 (-2 when ap >= pi): the heading test of the quantised prefilter records
};

// numpy float `%`: the result takes the sign of the divisor (npy_divmod)
__device__ __forceinline__ double py_mod(double a, double b)
{
    double r = fmod(a, b);
    if (r != 0.0) {
        if ((b < 0.0) != (r < 0.0)) r += b;
    } else {
        r = copysign(0.0, b);
    }
    return r;
}

// fmod(a, 2 pi) for far-out arguments (|a| >= 4 pi: an angle that went through normalize_angle once never gets here).
// ONE copy per kernel instead of one per use: ocml's fmod is a loop of ~100 instructions, normalize_angle is inlined two
// dozen times into the step kernels, and their code (74 KB) is larger than the instruction cache (64 KB).  Measured in
// round 4 (same box, builds side by side): 11.77 against 11.95 us per step, no register or spill changes.
__device__ __attribute__((noinline)) double fmod_two_pi_far(double a) { return fmod(a, kTwoPi); }

// a % (2 pi) with numpy semantics.  fmod is exact (a - k*b is representable); for |a| < 2b it is a itself or one
// exact subtraction / addition (Sterbenz), so the library call is only needed for far-out arguments.
__device__ __forceinline__ double py_mod_two_pi(double a)
{
    const double b = kTwoPi;
    double r;
    if (a >= 0.0) {
        r = a < b ? a : (a < 2.0 * b ? a - b : fmod_two_pi_far(a));
    } else {
        r = a > -b ? a : (a > -2.0 * b ? a + b : fmod_two_pi_far(a));
    }
    if (r != 0.0) {
        if (r < 0.0) r += b;
    } else {
        r = 0.0;
    }
    return r;
}

// x / d for a divisor that is a constant of the handle (dt, L) or of the program (pi), with rd = 1 / d correctly rounded: the
// quotient estimate, its exact residual (one fma) and the correction (one fma) -- Markstein's sequence, which returns the
// correctly rounded quotient, i.e. the very double x / d is (for every divisor whose significand is not all ones: the host
// refuses such a dt / L; 3 x 10^6 random and swept dividends per divisor checked against exact rational arithmetic, and the
// reference trajectories are the standing check).  Three instructions instead of the ~28 of a float64 division, six times on
// the movers' chain of step_local_kernel.
__device__ __forceinline__ double div_by_const(double x, double d, double rd)
{
    const double q = x * rd;
    return fma(fma(-q, d, x), rd, q);
}
constexpr double kInvPi = 1.0 / kPi;   // (constant expression: IEEE division, correctly rounded)

// utilities/coordinate_transformations.py:28-36
__device__ __forceinline__ double normalize_angle(double z) { return py_mod_two_pi(z + kPi) - kPi; }

// np.clip == minimum(maximum(a, lo), hi)
__device__ __forceinline__ double clipd(double a, double lo, double hi)
{
    double t = a < lo ? lo : a;
    return t > hi ? hi : t;
}

__device__ __forceinline__ double signd(double a) { return (double)((a > 0.0) - (a < 0.0)); }

struct Pose {
    double x, y, th;
};

// cos and sin of one angle with ONE range reduction: ocml's sincos runs the reduction and the two kernel polynomials its
// sin and cos run, once -- the values are the ones cos(x) / sin(x) return (the golden trajectories are the check)
__device__ __forceinline__ void cos_sin(double x, double& c, double& s) { sincos(x, &s, &c); }

// robot_models/differential_drive.py:21-40
__device__ __forceinline__ Pose kinematic_step(Pose p, double v, double w, double dt)
{
    double half_wdt = 0.5 * w * dt;
    double t = div_by_const(half_wdt, kPi, kInvPi);   // np.sinc(half_wdt / np.pi)
    double yy = kPi * (t == 0.0 ? 1.0e-20 : t);
    double sinc = sin(yy) / yy;
    double v_factor = v * dt * sinc;
    double a = p.th + half_wdt;
    Pose o;
    double ca, sa;
    cos_sin(a, ca, sa);
    o.x = p.x + v_factor * ca;
    o.y = p.y + v_factor * sa;
    o.th = normalize_angle(p.th + w * dt);
    return o;
}

// robot_models/differential_drive.py:43-74.  z[k] is consumed only when variance_k > 0; `drawn` gets the mask.
__device__ __forceinline__ Pose kinematic_step_noise(Pose p, double v, double w, double dt, const double* alpha,
                                                     const double z[3], int& drawn)
{
    double var0 = alpha[0] * (v * v) + alpha[1] * (w * w);
    if (var0 > 0.0) {
        v = v + (0.0 + sqrt(var0) * z[0]);
        drawn |= 1;
    } else {
        v = v + 0.0;
    }
    double var1 = alpha[2] * (v * v) + alpha[3] * (w * w);
    if (var1 > 0.0) {
        w = w + (0.0 + sqrt(var1) * z[1]);
        drawn |= 2;
    } else {
        w = w + 0.0;
    }
    double var2 = alpha[4] * (v * v) + alpha[5] * (w * w);
    double rot = 0.0;
    if (var2 > 0.0) {
        rot = 0.0 + sqrt(var2) * z[2];
        drawn |= 4;
    }
    Pose o = kinematic_step(p, v, w, dt);
    o.th = normalize_angle(o.th + rot * dt);
    return o;
}

// utilities/path_tools.py:298-323 for the two-row call of tricycle_model.py:520-529
// cos / sin of the heading before the step, when the caller already holds them (by value: a pointer to a local would
// send them through scratch memory)
struct KnownHeading {
    double c0, s0;
    bool known;
};

__device__ __forceinline__ KnownHeading no_known_heading()
{
    KnownHeading k;
    k.c0 = k.s0 = 0.0;
    k.known = false;
    return k;
}

// (BY_RCP: the caller holds inv_dt = 1 / dt correctly rounded -- div_by_const --; otherwise the two quotients are divisions)
template <bool BY_RCP = false>
__device__ __forceinline__ int path_velocity(Pose p0, Pose p1, double dt, double& v, double& w,
                                             KnownHeading old_heading = no_known_heading(), double inv_dt = 0.0)
{
    double dx = p1.x - p0.x, dy = p1.y - p0.y;
    double c0, s0;
    if (old_heading.known) {
        c0 = old_heading.c0;
        s0 = old_heading.s0;
    } else {
        cos_sin(p0.th, c0, s0);
    }
    double sign = signd(c0 * dx + s0 * dy);
    if (sign == 0.0) sign = signd(s0 * dy);
    double ds = sqrt(dx * dx + dy * dy) * sign;
    double da = p1.th - p0.th;
    if (da < -kPi) da += kTwoPi;
    if (da > kPi) da -= kTwoPi;
    if (BY_RCP) {
        v = div_by_const(ds, dt, inv_dt);
        w = div_by_const(da, dt, inv_dt);
    } else {
        v = ds / dt;
        w = da / dt;
    }
    return (fabs(da) < kPi) ? 0 : BCP_ERR_ANGLE_JUMP;
}

struct Robot {
    Pose p;
    double v, w, steer, wheel;
};

// TricycleRobot.step (robot_models/tricycle_model.py:478-538, with :71-188) and
// DiffDriveRobot.step (robot_models/differential_drive.py:236-265), in two halves:
//   robot_step_begin: everything that needs neither the odometry noise nor the old heading's cos / sin -- the tricycle's
//                     front-wheel column, cos / sin of the new wheel angle and the velocity model (tricycle_model.py:127-188);
//   robot_step_end:   the (noisy) kinematic step and the measured velocities (differential_drive.py:21-74,
//                     path_tools.py:298-323).
// step_local_kernel runs the first half while the noise and the old heading are still being computed by other waves.
// the robot constants of DevParams by value: a kernel fetches them once, ahead of time (step_local_kernel)
struct RobotConsts {
    int32_t model, dynamic_model, model_front_column_pid, noise_on;
    double dt, L, max_wheel_angle, max_wheel_speed, max_lin_acc, max_ang_acc, p_gain, inv_dt, inv_L;
    double alpha[6];
};

__device__ __forceinline__ RobotConsts robot_consts(const DevParams& P)
{
    RobotConsts c;
    c.model = P.model;
    c.dynamic_model = P.dynamic_model;
    c.model_front_column_pid = P.model_front_column_pid;
    c.noise_on = P.noise_on;
    c.dt = P.dt;
    c.L = P.L;
    c.max_wheel_angle = P.max_wheel_angle;
    c.max_wheel_speed = P.max_wheel_speed;
    c.max_lin_acc = P.max_lin_acc;
    c.max_ang_acc = P.max_ang_acc;
    c.p_gain = P.p_gain;
    c.inv_dt = P.inv_dt;
    c.inv_L = P.inv_L;
#pragma unroll
    for (int k = 0; k < 6; ++k) c.alpha[k] = P.alpha[k];
    return c;
}

struct RobotDrive {
    double v, w;          // what goes into the kinematic step
    bool noisy;           // ... through kinematic_step_noise (tricycle: only the dynamic model is noisy, tricycle_model.py:489-519)
};

template <typename Params>   // DevParams or RobotConsts
__device__ __forceinline__ RobotDrive robot_step_begin(const Params& P, Robot& r, double cmd0, double cmd1)
{
    RobotDrive d;
    if (P.model == BCP_MODEL_TRICYCLE) {
        double wa = r.wheel, new_wa;
        if (P.model_front_column_pid) {   // tricycle_model.py:127-154
            double max_delta = P.max_wheel_speed * P.dt;
            double delta = clipd(P.p_gain * (cmd1 - wa), -max_delta, max_delta);
            new_wa = clipd(wa + delta, -P.max_wheel_angle, P.max_wheel_angle);
        } else {
            new_wa = clipd(cmd1, -P.max_wheel_angle, P.max_wheel_angle);
        }
        double cw, sw;
        cos_sin(new_wa, cw, sw);
        double des_v = cmd0 * cw;
        double des_w = div_by_const(cmd0 * sw, P.L, P.inv_L);
        if (P.dynamic_model) {            // tricycle_model.py:157-188
            double acc_v = div_by_const(des_v - r.v, P.dt, P.inv_dt);
            double acc_w = div_by_const(des_w - r.w, P.dt, P.inv_dt);
            double lin = clipd(acc_v, -2 * P.max_lin_acc, P.max_lin_acc);
            double ang = clipd(acc_w, -P.max_ang_acc, P.max_ang_acc);
            double nv = r.v + lin * P.dt;
            double nw = r.w + ang * P.dt;
            if (0.0 > nv) nv = 0.0;
            d.v = nv;
            d.w = nw;
            d.noisy = P.noise_on != 0;
        } else {                          // tricycle_kinematic_step :38-68
            d.v = des_v;
            d.w = des_w;
            d.noisy = false;
        }
        r.steer = wa - cmd1;              // :532
        r.wheel = new_wa;
    } else {
        d.v = cmd0;
        d.w = cmd1;
        d.noisy = P.noise_on != 0;
    }
    return d;
}

// (old_heading: step_local_kernel has an idle wave compute cos / sin of the old heading while the state loads are in flight)
template <typename Params>   // DevParams or RobotConsts
__device__ __forceinline__ int robot_step_end(const Params& P, Robot& r, RobotDrive d, const double z[3], int& drawn,
                                              KnownHeading old_heading = no_known_heading())
{
    const Pose last = r.p;
    const Pose np_ = d.noisy ? kinematic_step_noise(last, d.v, d.w, P.dt, P.alpha, z, drawn) : kinematic_step(last, d.v, d.w, P.dt);
    double mv, mw;
    const int err = path_velocity<true>(last, np_, P.dt, mv, mw, old_heading, P.inv_dt);
    r.p = np_;
    r.v = mv;
    r.w = mw;
    return err;
}

// robot_step_end in two parts: the new pose first -- what the other waves of step_local_kernel wait for -- then the
// measured velocities (path_velocity: a square root and two divisions that nobody else needs)
template <typename Params>
__device__ __forceinline__ Pose robot_step_pose(const Params& P, const Robot& r, RobotDrive d, const double z[3], int& drawn)
{
    return d.noisy ? kinematic_step_noise(r.p, d.v, d.w, P.dt, P.alpha, z, drawn) : kinematic_step(r.p, d.v, d.w, P.dt);
}

template <typename Params>
__device__ __forceinline__ int robot_step_measure(const Params& P, Robot& r, Pose np_, KnownHeading old_heading = no_known_heading())
{
    double mv, mw;
    const int err = path_velocity<true>(r.p, np_, P.dt, mv, mw, old_heading, P.inv_dt);
    r.p = np_;
    r.v = mv;
    r.w = mw;
    return err;
}

__device__ __forceinline__ int robot_step(const DevParams& P, Robot& r, double cmd0, double cmd1, const double z[3],
                                          int& drawn, KnownHeading old_heading = no_known_heading())
{
    const RobotDrive d = robot_step_begin(P, r, cmd0, cmd1);
    return robot_step_end(P, r, d, z, drawn, old_heading);
}

// Philox4x32-10 (Salmon et al., SC'11), counter = (env_lo, env_hi, step_lo, step_hi), key = seed.
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

// Three standard normals for (seed, env, step): Box-Muller in float64 on 32-bit uniforms -- the stand-in for the float64
// np.random.normal draws of robot_models/differential_drive.py:43-52 (parity runs replay exact values through noise_z).
// u = (k + 1/2) 2^-32 lies in (0, 1); r = sqrt(-2 ln u) <= 6.66; the angle 2 pi k 2^-32 goes through sincospi (exact
// argument reduction).  The only departure from N(0, 1) is the 2^-32 lattice of the uniforms: |z| <= 6.66, mass beyond
// that 2.7e-11 (numpy's 53-bit uniforms reach 8.6 sigma).  tests/test_gpu_noise.py checks the distribution.
//
// The stream (round 3): Philox words (c0, c1) make Box-Muller pair A, (c2, c3) pair B, and
//     z[1] = A.cos,  z[2] = A.sin,  z[0] = B.cos
// -- the two slots PlanEnv's noise model draws (alpha1 = alpha2 = 0: slot 0 is never consumed, envs/base/env.py:228-231) are
// the two halves of ONE pair, so a step that cannot draw slot 0 computes one logarithm, one square root and one sincospi
// per env instead of two each.  (Rounds 1-2: z[0], z[1] = pair A, z[2] = B.sin.)
template <int PAIR>   // 0 = A, 1 = B
__device__ __forceinline__ void box_muller_pair(const uint32_t c[4], double& cos_part, double& sin_part)
{
    const double k = 2.3283064365386963e-10;  // 2^-32
    const double u = ((double)c[2 * PAIR] + 0.5) * k;
    const double r = sqrt(-2.0 * log(u));
    double sn, cs;
    sincospi(2.0 * ((double)c[2 * PAIR + 1] * k), &sn, &cs);
    cos_part = r * cs;
    sin_part = r * sn;
}

// slots 1 and 2 (pair A)
__device__ __forceinline__ void device_normals_12(uint64_t seed, uint64_t env, uint64_t step, double& z1, double& z2)
{
    uint32_t c[4] = {(uint32_t)env, (uint32_t)(env >> 32), (uint32_t)step, (uint32_t)(step >> 32)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    box_muller_pair<0>(c, z1, z2);
}

// slot 0 (pair B)
__device__ __forceinline__ double device_normal_0(uint64_t seed, uint64_t env, uint64_t step)
{
    uint32_t c[4] = {(uint32_t)env, (uint32_t)(env >> 32), (uint32_t)step, (uint32_t)(step >> 32)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    double z0, twin;
    box_muller_pair<1>(c, z0, twin);
    return z0;
}

__device__ __forceinline__ void device_normals(uint64_t seed, uint64_t env, uint64_t step, double z[3])
{
    uint32_t c[4] = {(uint32_t)env, (uint32_t)(env >> 32), (uint32_t)step, (uint32_t)(step >> 32)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    double twin;
    box_muller_pair<0>(c, z[1], z[2]);
    box_muller_pair<1>(c, z[0], twin);
}

}  // namespace bcp
