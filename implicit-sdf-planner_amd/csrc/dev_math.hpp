// Device-side fp64 math of the collision sweeps (gfx950).  Product code: hand-written for the GPU, it shares
// NOTHING with oracle/ — parity is established by tests, not by construction.
//   quat_to_rot          Eigen::Quaterniond::toRotationMatrix as used at back_end_optimizer.hpp:492,614
//   smoothed_l1          back_end_optimizer.hpp:332-356
//   flat_forward/backward src/utils/include/utils/flatness.hpp:53-148, :230-406
//   poly_basis           back_end_optimizer.hpp:472-488 (beta0..beta4)
#pragma once
#include <hip/hip_runtime.h>

namespace isdf {

// small fixed-size vectors, templated on the scalar so the conservative fp32 pre-filter of tile_kernel can reuse the
// shape code that the exact fp64 path uses
template <typename T> struct v3 { T x, y, z; };
using d3 = v3<double>;
using f3 = v3<float>;
struct d4 { double w, x, y, z; };
struct m3 { double a[9]; };   // row-major

template <typename T> __device__ __forceinline__ v3<T> mkv(T x, T y, T z) { v3<T> r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ d3 mk3(double x, double y, double z) { return mkv<double>(x, y, z); }
template <typename T> __device__ __forceinline__ v3<T> operator+(v3<T> a, v3<T> b) { return mkv<T>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <typename T> __device__ __forceinline__ v3<T> operator-(v3<T> a, v3<T> b) { return mkv<T>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <typename T> __device__ __forceinline__ v3<T> operator*(T s, v3<T> a) { return mkv<T>(s * a.x, s * a.y, s * a.z); }
template <typename T> __device__ __forceinline__ T dot3(v3<T> a, v3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// overloaded scalar math (float versions map to the single-instruction f32 ops)
#ifdef ISDF_LEAN_MATH
// Translation units that define ISDF_LEAN_MATH (the exact pass of the tile sweep, where fp64 instruction issue is the
// bound) get the compiler's own rsq + Newton sequence for sqrt WITHOUT its range scaling and class handling: same
// arithmetic, hence the same bits, for arguments in the normal range - which sums of squares of metre-scale lengths are -
// and 0 -> 0.  10 instructions instead of 18.
__device__ __forceinline__ double m_sqrt(double x) {
    // the reciprocal root of max(|x|, DBL_MIN) (one instruction, |.| is an operand modifier) instead of a compare and two
    // selects behind the sequence: x = 0 gives 0 * finite = 0 through every step, a NaN stays a NaN, x > 0 is untouched
    const double y = __builtin_amdgcn_rsq(fmax(fabs(x), 2.2250738585072014e-308));
    double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    double d = fma(-g, g, x);
    g = fma(d, h, g);
    d = fma(-g, g, x);
    g = fma(d, h, g);
    return g;
}
// 1 / x for the shared-divisor quotients below: rcp + two Newton steps (<= 1 ulp, normal range; 5 instructions instead of
// the 11 of the correctly-rounded division with its scaling and fix-up).
__device__ __forceinline__ double m_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    e = fma(-x, r, 1.0);
    return fma(r, e, r);
}
#else
__device__ __forceinline__ double m_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ double m_rcp(double x) { return 1.0 / x; }
#endif
__device__ __forceinline__ float m_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double m_abs(double x) { return fabs(x); }
__device__ __forceinline__ float m_abs(float x) { return fabsf(x); }
__device__ __forceinline__ double m_max(double a, double b) { return fmax(a, b); }
__device__ __forceinline__ float m_max(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ double m_min(double a, double b) { return fmin(a, b); }
__device__ __forceinline__ float m_min(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ double m_floor(double x) { return floor(x); }
__device__ __forceinline__ float m_floor(float x) { return floorf(x); }
__device__ __forceinline__ double m_atan2(double y, double x) { return atan2(y, x); }
__device__ __forceinline__ float m_atan2(float y, float x) { return atan2f(y, x); }
__device__ __forceinline__ void m_sincos(double a, double &s, double &c) { sincos(a, &s, &c); }
__device__ __forceinline__ void m_sincos(float a, float &s, float &c) { sincosf(a, &s, &c); }

template <typename T> __device__ __forceinline__ T norm3(v3<T> a) { return m_sqrt(dot3(a, a)); }
// Eigen normalized(): unchanged when the squared norm is not > 0
// a / |a| through one refined reciprocal square root (no sqrt, no division): |error| ~ 1 ulp.  Zero stays zero
// (Eigen's normalized()).
__device__ __forceinline__ d3 normalized3_lean(d3 a) {
    const double z = dot3(a, a);
    if (!(z > 0.0)) return a;
    const double y = __builtin_amdgcn_rsq(z);
    const double g = z * y;
    double h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    h = fma(h, r, h);                       // 0.5 / sqrt(z) after one Newton step
    const double r2 = fma(-h, z * (h + h), 0.5);   // second step on the reciprocal root itself
    h = fma(h, r2, h);
    const double inv = h + h;
    return mk3(a.x * inv, a.y * inv, a.z * inv);
}
__device__ __forceinline__ d3 normalized3(d3 a) {
    const double z = dot3(a, a);
    if (z > 0.0) { const double n = sqrt(z); return mk3(a.x / n, a.y / n, a.z / n); }
    return a;
}
__device__ __forceinline__ d3 mat_mul(const m3 &R, d3 v) {
    return mk3(R.a[0] * v.x + R.a[1] * v.y + R.a[2] * v.z, R.a[3] * v.x + R.a[4] * v.y + R.a[5] * v.z,
               R.a[6] * v.x + R.a[7] * v.y + R.a[8] * v.z);
}
__device__ __forceinline__ d3 mat_tmul(const m3 &R, d3 v) {   // R^T v
    return mk3(R.a[0] * v.x + R.a[3] * v.y + R.a[6] * v.z, R.a[1] * v.x + R.a[4] * v.y + R.a[7] * v.z,
               R.a[2] * v.x + R.a[5] * v.y + R.a[8] * v.z);
}

__device__ __forceinline__ m3 quat_to_rot(d4 q) {
    const double tx = 2.0 * q.x, ty = 2.0 * q.y, tz = 2.0 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    m3 R;
    R.a[0] = 1.0 - (tyy + tzz); R.a[1] = txy - twz;         R.a[2] = txz + twy;
    R.a[3] = txy + twz;         R.a[4] = 1.0 - (txx + tzz); R.a[5] = tyz - twx;
    R.a[6] = txz - twy;         R.a[7] = tyz + twx;         R.a[8] = 1.0 - (txx + tyy);
    return R;
}

// returns false (f, df untouched) for x < 0
// the same with the reciprocal of mu precomputed (two divisions fewer per voxel in the exact tile pass)
__device__ __forceinline__ bool smoothed_l1_inv(double x, double mu, double inv_mu, double &f, double &df) {
    if (x < 0.0) return false;
    if (x > mu) { f = x - 0.5 * mu; df = 1.0; return true; }
    const double xdmu = x * inv_mu;
    const double sq = xdmu * xdmu;
    const double mumxd2 = mu - 0.5 * x;
    f = mumxd2 * sq * xdmu;
    df = sq * ((-0.5) * xdmu + 3.0 * mumxd2 * inv_mu);
    return true;
}
__device__ __forceinline__ bool smoothed_l1(double x, double mu, double &f, double &df) {
    if (x < 0.0) return false;
    if (x > mu) { f = x - 0.5 * mu; df = 1.0; return true; }
    const double xdmu = x / mu;
    const double sq = xdmu * xdmu;
    const double mumxd2 = mu - 0.5 * x;
    f = mumxd2 * sq * xdmu;
    df = sq * ((-0.5) * xdmu + 3.0 * mumxd2 / mu);
    return true;
}

// g^T * dR^T/dq_k * d for k = w,x,y,z (the four matrices of back_end_optimizer.hpp:127-173, multiplied out)
__device__ __forceinline__ d4 quat_step_grad(d4 q, d3 g, d3 d) {
    // With u = (x, y, z), c = g x d the four products g^T (dR^T/dq_k) d of back_end_optimizer.hpp:127-173 are
    //   k = w:      2 u.c
    //   k = x,y,z:  -4 (g.d) u + 2 (u.d) g + 2 (u.g) d + 2 w c
    // (term by term the same monomials as the matrices multiplied out; 33 operations instead of 45, rounding-level differences)
    const d3 c = mk3(g.y * d.z - g.z * d.y, g.z * d.x - g.x * d.z, g.x * d.y - g.y * d.x);
    const double gd = g.x * d.x + g.y * d.y + g.z * d.z;
    const double ud = q.x * d.x + q.y * d.y + q.z * d.z;
    const double ug = q.x * g.x + q.y * g.y + q.z * g.z;
    const double A = -4.0 * gd, B = 2.0 * ud, C = 2.0 * ug, W = 2.0 * q.w;
    d4 r;
    r.w = 2.0 * (q.x * c.x + q.y * c.y + q.z * c.z);
    r.x = A * q.x + B * g.x + C * d.x + W * c.x;
    r.y = A * q.y + B * g.y + C * d.y + W * c.y;
    r.z = A * q.z + B * g.z + C * d.z + W * c.z;
    return r;
}

struct FlatP { double mass, grav, dh, dv, cp, veps, dh_over_m; };

// a / b where several quotients share the divisor b: translation units that define ISDF_LEAN_MATH (the tile sweep, bound by
// fp64 instruction issue) form 1 / b once and multiply (<= 1 ulp per quotient); everywhere else it is the plain division.
#ifdef ISDF_LEAN_MATH
#define ISDF_QUOT(a, b, inv_b) ((a) * (inv_b))
#else
#define ISDF_QUOT(a, b, inv_b) ((a) / (b))
#endif

// Shared forward intermediates of the flatness map.
struct FlatS {
    double cp_term, w_term, zu0, zu1, zu2, s0, s1, s2, zu_sqr_norm, zu_norm, z0, z1, z2, tilt_den;
};
__device__ __forceinline__ void flat_core(const FlatP &P, d3 v, d3 a, FlatS &s) {
    s.cp_term = sqrt(v.x * v.x + v.y * v.y + v.z * v.z + P.veps);
    s.w_term = 1.0 + P.cp * s.cp_term;
    s.zu0 = a.x + P.dh_over_m * (s.w_term * v.x);
    s.zu1 = a.y + P.dh_over_m * (s.w_term * v.y);
    s.zu2 = a.z + P.dh_over_m * (s.w_term * v.z) + P.grav;
    s.s0 = s.zu0 * s.zu0; s.s1 = s.zu1 * s.zu1; s.s2 = s.zu2 * s.zu2;
    s.zu_sqr_norm = s.s0 + s.s1 + s.s2;
    s.zu_norm = sqrt(s.zu_sqr_norm);
    [[maybe_unused]] const double inv_zn = m_rcp(s.zu_norm);
    s.z0 = ISDF_QUOT(s.zu0, s.zu_norm, inv_zn); s.z1 = ISDF_QUOT(s.zu1, s.zu_norm, inv_zn); s.z2 = ISDF_QUOT(s.zu2, s.zu_norm, inv_zn);
    s.tilt_den = sqrt(2.0 * (1.0 + s.z2));
}
__device__ __forceinline__ d4 flat_quat(const FlatS &s) {
    [[maybe_unused]] const double inv_t = m_rcp(s.tilt_den);
    d4 q; q.w = 0.5 * s.tilt_den; q.x = ISDF_QUOT(-s.z1, s.tilt_den, inv_t); q.y = ISDF_QUOT(s.z0, s.tilt_den, inv_t); q.z = 0.0; return q;
}
struct FlatS2 {
    double ng_den, ng00, ng01, ng02, ng11, ng12, ng22, zu01, zu12, zu02, v_dot_a, dw_term;
    double dw0, dw1, dw2, dzt0, dzt1, dzt2, dz0, dz1, dz2, omg_den, omg_term;
};
__device__ __forceinline__ void flat_core2(const FlatP &P, d3 v, d3 a, d3 j, const FlatS &s, FlatS2 &t) {
    t.ng_den = s.zu_sqr_norm * s.zu_norm;
    [[maybe_unused]] const double inv_ng = m_rcp(t.ng_den);
    t.ng00 = ISDF_QUOT(s.s1 + s.s2, t.ng_den, inv_ng);
    t.zu01 = s.zu0 * s.zu1; t.zu12 = s.zu1 * s.zu2; t.zu02 = s.zu0 * s.zu2;
    t.ng01 = ISDF_QUOT(-t.zu01, t.ng_den, inv_ng);
    t.ng02 = ISDF_QUOT(-t.zu02, t.ng_den, inv_ng);
    t.ng11 = ISDF_QUOT(s.s0 + s.s2, t.ng_den, inv_ng);
    t.ng12 = ISDF_QUOT(-t.zu12, t.ng_den, inv_ng);
    t.ng22 = ISDF_QUOT(s.s0 + s.s1, t.ng_den, inv_ng);
    t.v_dot_a = v.x * a.x + v.y * a.y + v.z * a.z;
    [[maybe_unused]] const double inv_cp = m_rcp(s.cp_term);
    t.dw_term = ISDF_QUOT(P.cp * t.v_dot_a, s.cp_term, inv_cp);
    t.dw0 = s.w_term * a.x + t.dw_term * v.x;
    t.dw1 = s.w_term * a.y + t.dw_term * v.y;
    t.dw2 = s.w_term * a.z + t.dw_term * v.z;
    t.dzt0 = j.x + P.dh_over_m * t.dw0;
    t.dzt1 = j.y + P.dh_over_m * t.dw1;
    t.dzt2 = j.z + P.dh_over_m * t.dw2;
    t.dz0 = t.ng00 * t.dzt0 + t.ng01 * t.dzt1 + t.ng02 * t.dzt2;
    t.dz1 = t.ng01 * t.dzt0 + t.ng11 * t.dzt1 + t.ng12 * t.dzt2;
    t.dz2 = t.ng02 * t.dzt0 + t.ng12 * t.dzt1 + t.ng22 * t.dzt2;
    t.omg_den = s.z2 + 1.0;
    [[maybe_unused]] const double inv_omg = m_rcp(t.omg_den);
    t.omg_term = ISDF_QUOT(t.dz2, t.omg_den, inv_omg);
}
__device__ __forceinline__ d3 flat_omg(const FlatS &s, const FlatS2 &t) {
    [[maybe_unused]] const double inv_omg = m_rcp(t.omg_den);
    return mk3(-t.dz1 + s.z1 * t.omg_term, t.dz0 - s.z0 * t.omg_term, ISDF_QUOT(s.z1 * t.dz0 - s.z0 * t.dz1, t.omg_den, inv_omg));
}

// Reverse mode of (v,a,j) -> (quat, omega) with psi == 0; inputs are dL/dpos, dL/dvel, dL/dquat, dL/domega.
// flat_backward_from takes the forward intermediates the caller already holds (flat_core + flat_core2 of the same v, a, j).
__device__ __forceinline__ void flat_backward_from(const FlatP &P, d3 v, d3 a, const FlatS &s, const FlatS2 &t, d3 pos_grad, d3 vel_grad,
                                                   d4 quat_grad, d3 omg_grad, d3 &gP, d3 &gV, d3 &gA, d3 &gJ) {
    [[maybe_unused]] const double inv_tilt = m_rcp(s.tilt_den), inv_omg = m_rcp(t.omg_den), inv_ng = m_rcp(t.ng_den), inv_zn = m_rcp(s.zu_norm), inv_cp = m_rcp(s.cp_term);
    const double tilt0b = quat_grad.w, tilt1b = quat_grad.x, tilt2b = quat_grad.y;
    [[maybe_unused]] const double tilt_den_sqr = s.tilt_den * s.tilt_den;
    const double tilt_denb = ISDF_QUOT(s.z1 * tilt1b - s.z0 * tilt2b, tilt_den_sqr, inv_tilt * inv_tilt) + 0.5 * tilt0b;
    const double omg_termb = -s.z0 * omg_grad.y + s.z1 * omg_grad.x;
    double tb = ISDF_QUOT(omg_grad.z, t.omg_den, inv_omg);
    double z1b = t.dz0 * tb;
    const double dz0b = s.z1 * tb + omg_grad.y;
    double z0b = -(t.dz1 * tb);
    const double dz1b = -s.z0 * tb - omg_grad.x;
    const double omg_denb = -ISDF_QUOT((s.z1 * t.dz0 - s.z0 * t.dz1) * tb, t.omg_den, inv_omg) - ISDF_QUOT(ISDF_QUOT(t.dz2 * omg_termb, t.omg_den, inv_omg), t.omg_den, inv_omg);
    tb = -(t.omg_term * omg_grad.y);
    z0b += tb;
    tb = -(t.omg_term * omg_grad.x);
    z0b += ISDF_QUOT(tilt2b, s.tilt_den, inv_tilt);
    z1b += -tb - ISDF_QUOT(tilt1b, s.tilt_den, inv_tilt);
    const double dz2b = ISDF_QUOT(omg_termb, t.omg_den, inv_omg);
    const double z2b = omg_denb + ISDF_QUOT(tilt_denb, s.tilt_den, inv_tilt);

    const double ng02b = t.dzt0 * dz2b + t.dzt2 * dz0b;
    const double dzt0b = t.ng02 * dz2b + t.ng01 * dz1b + t.ng00 * dz0b;
    const double ng12b = t.dzt1 * dz2b + t.dzt2 * dz1b;
    const double dzt1b = t.ng12 * dz2b + t.ng11 * dz1b + t.ng01 * dz0b;
    const double ng22b = t.dzt2 * dz2b;
    const double dzt2b = t.ng22 * dz2b + t.ng12 * dz1b + t.ng02 * dz0b;
    const double ng01b = t.dzt0 * dz1b + t.dzt1 * dz0b;
    const double ng11b = t.dzt1 * dz1b;
    const double ng00b = t.dzt0 * dz0b;
    gJ = mk3(dzt0b, dzt1b, dzt2b);
    const double dw0b = P.dh_over_m * dzt0b, dw1b = P.dh_over_m * dzt1b, dw2b = P.dh_over_m * dzt2b;
    tb = ISDF_QUOT(P.cp * (v.z * dw2b + v.y * dw1b + v.x * dw0b), s.cp_term, inv_cp);
    gA = mk3(s.w_term * dw0b + v.x * tb, s.w_term * dw1b + v.y * tb, s.w_term * dw2b + v.z * tb);
    gV = mk3(t.dw_term * dw0b + a.x * tb, t.dw_term * dw1b + a.y * tb, t.dw_term * dw2b + a.z * tb);
    double cp_termb = -ISDF_QUOT(t.v_dot_a * tb, s.cp_term, inv_cp);
    tb = ISDF_QUOT(ng22b, t.ng_den, inv_ng);
    double zs0b = tb, zs1b = tb;
    double ng_denb = -ISDF_QUOT((s.s0 + s.s1) * tb, t.ng_den, inv_ng);
    const double zu12b = -ISDF_QUOT(ng12b, t.ng_den, inv_ng);
    tb = ISDF_QUOT(ng11b, t.ng_den, inv_ng);
    ng_denb += ISDF_QUOT(ISDF_QUOT(t.zu12 * ng12b, t.ng_den, inv_ng), t.ng_den, inv_ng) - ISDF_QUOT((s.s0 + s.s2) * tb, t.ng_den, inv_ng);
    zs0b += tb;
    double zs2b = tb;
    const double zu02b = -ISDF_QUOT(ng02b, t.ng_den, inv_ng);
    const double zu01b = -ISDF_QUOT(ng01b, t.ng_den, inv_ng);
    tb = ISDF_QUOT(ng00b, t.ng_den, inv_ng);
    ng_denb += ISDF_QUOT(ISDF_QUOT(t.zu02 * ng02b, t.ng_den, inv_ng), t.ng_den, inv_ng) + ISDF_QUOT(ISDF_QUOT(t.zu01 * ng01b, t.ng_den, inv_ng), t.ng_den, inv_ng) - ISDF_QUOT((s.s1 + s.s2) * tb, t.ng_den, inv_ng);
    const double zu_normb = s.zu_sqr_norm * ng_denb - (s.zu2 * z2b + s.zu1 * z1b + s.zu0 * z0b) / s.zu_sqr_norm;
    const double zu_sqr_normb = s.zu_norm * ng_denb + ISDF_QUOT(0.5 * zu_normb, s.zu_norm, inv_zn);
    tb += zu_sqr_normb;
    zs1b += tb;
    zs2b += tb;
    const double zu2b = ISDF_QUOT(z2b, s.zu_norm, inv_zn) + s.zu0 * zu02b + s.zu1 * zu12b + 2 * s.zu2 * zs2b;
    const double w2b = P.dh_over_m * zu2b;
    const double zu1b = ISDF_QUOT(z1b, s.zu_norm, inv_zn) + s.zu2 * zu12b + s.zu0 * zu01b + 2 * s.zu1 * zs1b;
    const double w1b = P.dh_over_m * zu1b;
    zs0b += zu_sqr_normb;
    const double zu0b = ISDF_QUOT(z0b, s.zu_norm, inv_zn) + s.zu2 * zu02b + s.zu1 * zu01b + 2 * s.zu0 * zs0b;
    const double w0b = P.dh_over_m * zu0b;
    const double w_termb = a.z * dw2b + a.y * dw1b + a.x * dw0b + v.z * w2b + v.y * w1b + v.x * w0b;
    gA.x += zu0b; gA.y += zu1b; gA.z += zu2b;
    cp_termb += P.cp * w_termb;
    const double vsb = ISDF_QUOT(0.5 * cp_termb, s.cp_term, inv_cp);
    gV.x += s.w_term * w0b + 2 * v.x * vsb + vel_grad.x;
    gV.y += s.w_term * w1b + 2 * v.y * vsb + vel_grad.y;
    gV.z += s.w_term * w2b + 2 * v.z * vsb + vel_grad.z;
    gP = pos_grad;
}
__device__ __forceinline__ void flat_backward(const FlatP &P, d3 v, d3 a, d3 jer, d3 pos_grad, d3 vel_grad,
                                              d4 quat_grad, d3 omg_grad, d3 &gP, d3 &gV, d3 &gA, d3 &gJ) {
    FlatS s; FlatS2 t;
    flat_core(P, v, a, s);
    flat_core2(P, v, a, jer, s, t);
    flat_backward_from(P, v, a, s, t, pos_grad, vel_grad, quat_grad, omg_grad, gP, gV, gA, gJ);
}

// Piece-local monomial bases at local time s1.
struct Basis { double b0[6], b1[6], b2[6], b3[6], b4[6]; };
__device__ __forceinline__ void poly_basis(double s1, Basis &B) {
    const double s2 = s1 * s1, s3 = s2 * s1, s4 = s2 * s2, s5 = s4 * s1;
    B.b0[0] = 1.0; B.b0[1] = s1; B.b0[2] = s2; B.b0[3] = s3; B.b0[4] = s4; B.b0[5] = s5;
    B.b1[0] = 0.0; B.b1[1] = 1.0; B.b1[2] = 2.0 * s1; B.b1[3] = 3.0 * s2; B.b1[4] = 4.0 * s3; B.b1[5] = 5.0 * s4;
    B.b2[0] = 0.0; B.b2[1] = 0.0; B.b2[2] = 2.0; B.b2[3] = 6.0 * s1; B.b2[4] = 12.0 * s2; B.b2[5] = 20.0 * s3;
    B.b3[0] = 0.0; B.b3[1] = 0.0; B.b3[2] = 0.0; B.b3[3] = 6.0; B.b3[4] = 24.0 * s1; B.b3[5] = 60.0 * s2;
    B.b4[0] = 0.0; B.b4[1] = 0.0; B.b4[2] = 0.0; B.b4[3] = 0.0; B.b4[4] = 24.0; B.b4[5] = 120.0 * s1;
}
// c^T beta for one piece; c points at the piece's row block, `ld` = 6N is the column stride.  K0 = the derivative order of
// the basis: its first K0 entries are structural zeros (poly_basis) and `s + c * 0.0 == s` for every finite c, so those
// terms may be skipped without changing a bit.
template <int K0 = 0> __device__ __forceinline__ d3 ct_beta(const double *__restrict__ c, int ld, const double *b) {
    d3 r;
    double s = 0.0;
#pragma unroll
    for (int k = K0; k < 6; k++) s += c[k] * b[k];
    r.x = s; s = 0.0;
#pragma unroll
    for (int k = K0; k < 6; k++) s += c[ld + k] * b[k];
    r.y = s; s = 0.0;
#pragma unroll
    for (int k = K0; k < 6; k++) s += c[2 * ld + k] * b[k];
    r.z = s;
    return r;
}

// Piecewise-quintic trajectory as the swept-volume path evaluates it.
//   traj_locate follows Trajectory::locatePieceIdx (src/utils/include/utils/trajectory.hpp:545-563): sequential
//     subtraction of durations, strict '>' (a junction time belongs to the earlier piece), overflow clamps to the end.
//   traj_eval follows Piece::getPos_Vel_Acc_Jerk (:105-149): running powers of t, ascending-power walk.
struct TrajView { const double *T; const double *C; int N; };
__device__ __forceinline__ int traj_locate(const TrajView &tr, double &t) {
    // Same arithmetic as the reference's loop (t -= dur while t > dur), written as a wave-uniform walk over the pieces:
    // the duration of piece k is one scalar load for the whole wavefront and lanes that have found their piece idle.
    int idx = 0;
    bool walking = true;
    double last = 0.0;
    for (int k = 0; k < tr.N; k++) {
        const double dur = tr.T[k];
        last = dur;
        walking = walking && (t > dur);
        if (walking) { t -= dur; idx = k + 1; }
        if (!__builtin_amdgcn_ballot_w64(walking)) break;
    }
    if (idx == tr.N) { idx--; t += last; }       // ran off the end: clamp to the last piece (last == T[N-1] here)
    return idx;
}
__device__ __forceinline__ void traj_eval(const TrajView &tr, int piece, double t, d3 &pos, d3 &vel, d3 &acc, d3 &jer) {
    pos = mk3(0, 0, 0); vel = pos; acc = pos; jer = pos;
    double pos_tn = 1.0, vel_tn = 1.0, acc_tn = 1.0, jer_tn = 1.0;
    int vel_n = 1, acc_m = 1, acc_n = 2, jl = 1, jm = 2, jn = 3;
    const int ld = 6 * tr.N;
    const double *c = tr.C + 6 * piece;
#pragma unroll
    for (int power = 0; power <= 5; power++) {
        const d3 ck = mk3(c[power], c[ld + power], c[2 * ld + power]);
        pos = pos + pos_tn * ck;
        pos_tn *= t;
        if (power >= 1) { vel = vel + (vel_n * vel_tn) * ck; vel_tn *= t; vel_n++; }
        if (power >= 2) { acc = acc + (acc_m * acc_n * acc_tn) * ck; acc_tn *= t; acc_m++; acc_n++; }
        if (power >= 3) { jer = jer + (jl * jm * jn * jer_tn) * ck; jer_tn *= t; jl++; jm++; jn++; }
    }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Sum of 8 per-lane doubles over the wavefront in one transposing butterfly: after the call, component k's total is
// in `v[0]` of lane wave_sum8_lane(k) (and of every lane with the same bits 3..5).  Halves the register set at each
// of the first three steps (xor 32, 16, 8), so it needs 7 + 3 exchanges instead of 8 x 6.  Fixed summation order.
__device__ __forceinline__ int wave_sum8_lane(int k) { return ((k & 1) << 3) | (((k >> 1) & 1) << 4) | ((k >> 2) << 5); }
// Lane exchanges without LDS round trips: gfx950's v_permlane32_swap / v_permlane16_swap trade half wavefronts / odd and even
// rows of two registers in place - with A = the component the lower half keeps and B = the one the upper half keeps,
// swap(A, B) leaves {A_low, B_low} and {A_high, B_high}, whose lane-wise sum IS the butterfly step (same operands as
// keep + shfl_xor(send): same bits) - and DPP row operations inside a row of 16.
template <int CTRL> __device__ __forceinline__ double dpp_mov_f64(double v) {
    // (every lane of these row patterns has a source, so there is no "old" operand to preserve: v_mov_dpp straight from the source
    // register - with an old operand, 0 or v itself, the compiler emits a v_mov in front of every DPP move)
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double swap32_sum(double a, double b) {
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ double swap16_sum(double a, double b) {
    const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ void wave_sum8(double (&v)[8], int lane) {
    const bool b3 = (lane & 8) != 0;
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] = swap32_sum(v[i], v[4 + i]);      // lanes 0-31 keep components i, lanes 32-63 components 4 + i
#pragma unroll
    for (int i = 0; i < 2; i++) v[i] = swap16_sum(v[i], v[2 + i]);      // even rows keep i, odd rows 2 + i
    {                                   // keep component 4*b5 + 2*b4 + b3: partner = lane ^ 8 = the row rotated by 8
        const double send = b3 ? v[0] : v[1], keep = b3 ? v[1] : v[0];
        v[0] = keep + dpp_mov_f64<0x128>(send);       // row_ror:8
    }
    v[0] += dpp_mov_f64<0x141>(v[0]);   // row_half_mirror: lane i + lane 7 - i of its group of 8
    v[0] += dpp_mov_f64<0x4E>(v[0]);    // quad_perm [2,3,0,1]
    v[0] += dpp_mov_f64<0xB1>(v[0]);    // quad_perm [1,0,3,2]
}

// The same butterfly for a wavefront that holds TWO independent blocks of 32 lanes / FOUR blocks of 16 (the rows): the levels
// that would combine lanes of different blocks (xor 32 / xor 32 and 16) are left out and the register set is halved at the
// following levels instead.  Per component the operands of every remaining addition are those of wave_sum8 run on the block
// alone in lanes 0.. of an otherwise zero wavefront (the levels left out only add those zeros), so the sums carry the same bits.
// halves: component k of a block -> v[0] of its lanes with ((lane >> 2) & 7) == k;  rows: ((lane >> 1) & 7) == k.
__device__ __forceinline__ void wave_sum8_halves(double (&v)[8], int lane) {
    const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0;
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] = swap16_sum(v[i], v[4 + i]);      // even rows keep components i, odd rows 4 + i
#pragma unroll
    for (int j = 0; j < 2; j++) {                                       // partner lane ^ 8: keep j (bit 3 clear) or 2 + j
        const double send = b3 ? v[j] : v[2 + j], keep = b3 ? v[2 + j] : v[j];
        v[j] = keep + dpp_mov_f64<0x128>(send);
    }
    {                                                                   // partner 7 - i of the group of 8 (the other bit 2): keep 0 or 1
        const double send = b2 ? v[0] : v[1], keep = b2 ? v[1] : v[0];
        v[0] = keep + dpp_mov_f64<0x141>(send);
    }
    v[0] += dpp_mov_f64<0x4E>(v[0]);
    v[0] += dpp_mov_f64<0xB1>(v[0]);
}
__device__ __forceinline__ void wave_sum8_rows(double (&v)[8], int lane) {
    const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0, b1 = (lane & 2) != 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {                                       // partner lane ^ 8: keep i (bit 3 clear) or 4 + i
        const double send = b3 ? v[i] : v[4 + i], keep = b3 ? v[4 + i] : v[i];
        v[i] = keep + dpp_mov_f64<0x128>(send);
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {                                       // partner 7 - i: keep j (bit 2 clear) or 2 + j
        const double send = b2 ? v[j] : v[2 + j], keep = b2 ? v[2 + j] : v[j];
        v[j] = keep + dpp_mov_f64<0x141>(send);
    }
    {                                                                   // partner lane ^ 2: keep 0 (bit 1 clear) or 1
        const double send = b1 ? v[0] : v[1], keep = b1 ? v[1] : v[0];
        v[0] = keep + dpp_mov_f64<0x4E>(send);
    }
    v[0] += dpp_mov_f64<0xB1>(v[0]);
}

} // namespace isdf
