// MINCO (s = 3, non-uniform time) in LOG DEPTH - the arithmetic of the device-resident objective callback (minco_dev.hip).
//
// What it replaces: minco.setParameters / getEnergy* / propogateGrad as TrajOptimizer::costFunctionLmbm calls them
// (src/utils/include/utils/minco.hpp:43-198 banded LU, :397-655; back_end_optimizer.hpp:358-430).  The reference factors the
// 6N x 6N band matrix of ALL polynomial coefficients - a chain of 6N dependent pivots (csrc/minco_host.hpp keeps that form,
// bitwise).  The same trajectory written in its junction states is a much smaller problem:
//
//   * piece k (duration T_k, h = 1 / T_k) is the quintic Hermite interpolant of its end states e_k = (p_k, v_k, a_k, p_k+1,
//     v_k+1, a_k+1) - closed form, piece by piece (`hermite`);
//   * the unknowns are u_j = (v_j, a_j) at the N - 1 inner waypoints; the reference's rows "3rd and 4th derivative continuous
//     at the junction" (minco.hpp:446-449) become a BLOCK-TRIDIAGONAL system K u = r with 2 x 2 blocks (`junction_row`).
//     Written as [-(snap jump); +(jerk jump)] it is the gradient of the jerk energy in (v_j, a_j): K is SYMMETRIC POSITIVE
//     DEFINITE, so block elimination needs no pivoting and the adjoint system of propogateGrad is the SAME matrix;
//   * K u = r is solved by parallel cyclic reduction: ceil(log2(N - 1)) rounds in which every row eliminates its two
//     neighbours at distance s = 1, 2, 4 ... (`pcr_normalise` / `pcr_combine`), one thread per row - 6 rounds at N = 40
//     instead of 240 dependent pivots;
//   * the gradient w.r.t. (waypoints, durations) follows by the implicit-function rule with ONE more solve, mu = K^-1 g_u
//     (`piece_adjoint` / `piece_grad_T`).
//
// The operation order differs from the band LU, so results agree with the reference to rounding (tests pin 1e-10 relative
// against the reference's own minco.hpp compiled here, oracle/_ref/libref_minco.so), not bit for bit.
// Everything is plain inline arithmetic on small structs so that tests/native/minco_pcr_shim.cpp runs the identical code on
// the CPU (one loop per round) and minco_dev.hip runs it with one thread per row.
#pragma once
#ifdef __HIPCC__
#define MPCR_HD __host__ __device__ __forceinline__
#else
#define MPCR_HD inline
#endif

namespace mpcr {

// 1 / x.  On the device: v_rcp_f64 + two Newton steps (<= 1 ulp in the normal range, where determinants of the junction blocks
// and durations live) - 5 dependent instructions instead of the ~11 of the correctly rounded quotient, on the critical path of
// every elimination round.
MPCR_HD double rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, e, r);
#else
    return 1.0 / x;
#endif
}

struct M2 { double a, b, c, d; };                    // [[a, b], [c, d]]
MPCR_HD M2 mul(const M2 &x, const M2 &y) { return {x.a * y.a + x.b * y.c, x.a * y.b + x.b * y.d, x.c * y.a + x.d * y.c, x.c * y.b + x.d * y.d}; }
MPCR_HD M2 inv(const M2 &x) {
    const double r = rcp(x.a * x.d - x.b * x.c);
    return {x.d * r, -x.b * r, -x.c * r, x.a * r};
}

// one block row of K u = r for the three axes: L u_(j-s) + D u_j + U u_(j+s) = r;  r[axis] = (v-equation, a-equation)
struct Row { M2 L, D, U; double r[3][2]; };
// the row divided by its diagonal block: what a neighbour needs of it
struct Norm { M2 L, U; double r[3][2]; };

// Row j (1 <= j <= N - 1) of the junction system.  hl / hr: 1 / T of the pieces left and right of waypoint j; dpl = p_j -
// p_(j-1), dpr = p_(j+1) - p_j per axis.  Equations: [-(snap_left(T) - snap_right(0)); jerk_left(T) - jerk_right(0)] = 0.
MPCR_HD void junction_blocks(double hl, double hr, M2 &L, M2 &D, M2 &U) {
    const double hl2 = hl * hl, hr2 = hr * hr, hl3 = hl2 * hl, hr3 = hr2 * hr;
    L = {168.0 * hl3, 24.0 * hl2, -24.0 * hl2, -3.0 * hl};
    D = {192.0 * (hl3 + hr3), 36.0 * (hr2 - hl2), 36.0 * (hr2 - hl2), 9.0 * (hl + hr)};
    U = {168.0 * hr3, -24.0 * hr2, 24.0 * hr2, -3.0 * hr};
}
MPCR_HD void junction_rhs1(double hl, double hr, double dpl, double dpr, double r[2]) {      // one axis
    const double hl3 = hl * hl * hl, hr3 = hr * hr * hr, hl4 = hl3 * hl, hr4 = hr3 * hr;
    r[0] = 360.0 * (hl4 * dpl + hr4 * dpr);
    r[1] = 60.0 * (hr3 * dpr - hl3 * dpl);
}
MPCR_HD void junction_rhs(double hl, double hr, const double dpl[3], const double dpr[3], double r[3][2]) {
    for (int d = 0; d < 3; d++) junction_rhs1(hl, hr, dpl[d], dpr[d], r[d]);
}
// known end state (v, a) on the left of the first row / on the right of the last row moves to the right-hand side
MPCR_HD void rhs_minus(const M2 &B, const double va[3][2], double r[3][2]) {
    for (int d = 0; d < 3; d++) {
        r[d][0] -= B.a * va[d][0] + B.b * va[d][1];
        r[d][1] -= B.c * va[d][0] + B.d * va[d][1];
    }
}

MPCR_HD Norm pcr_normalise(const Row &w) {
    const M2 i = inv(w.D);
    Norm n;
    n.L = mul(i, w.L); n.U = mul(i, w.U);
    for (int d = 0; d < 3; d++) {
        n.r[d][0] = i.a * w.r[d][0] + i.b * w.r[d][1];
        n.r[d][1] = i.c * w.r[d][0] + i.d * w.r[d][1];
    }
    return n;
}
// eliminates the neighbours at distance s: lo = normalised row j - s (null: none), hi = row j + s
MPCR_HD void pcr_combine(Row &w, const Norm *lo, const Norm *hi) {
    M2 D = w.D;
    M2 Ln = {0, 0, 0, 0}, Un = {0, 0, 0, 0};
    if (lo) {
        const M2 t = mul(w.L, lo->U);
        D.a -= t.a; D.b -= t.b; D.c -= t.c; D.d -= t.d;
        const M2 l = mul(w.L, lo->L);
        Ln = {-l.a, -l.b, -l.c, -l.d};
        for (int d = 0; d < 3; d++) {
            const double r0 = lo->r[d][0], r1 = lo->r[d][1];
            w.r[d][0] -= w.L.a * r0 + w.L.b * r1;
            w.r[d][1] -= w.L.c * r0 + w.L.d * r1;
        }
    }
    if (hi) {
        const M2 t = mul(w.U, hi->L);
        D.a -= t.a; D.b -= t.b; D.c -= t.c; D.d -= t.d;
        const M2 u = mul(w.U, hi->U);
        Un = {-u.a, -u.b, -u.c, -u.d};
        for (int d = 0; d < 3; d++) {
            const double r0 = hi->r[d][0], r1 = hi->r[d][1];
            w.r[d][0] -= w.U.a * r0 + w.U.b * r1;
            w.r[d][1] -= w.U.c * r0 + w.U.d * r1;
        }
    }
    w.D = D; w.L = Ln; w.U = Un;
}
// after the last round the row stands alone: u = D^-1 r
MPCR_HD void pcr_finish(const Row &w, double u[3][2]) {
    const M2 i = inv(w.D);
    for (int d = 0; d < 3; d++) {
        u[d][0] = i.a * w.r[d][0] + i.b * w.r[d][1];
        u[d][1] = i.c * w.r[d][0] + i.d * w.r[d][1];
    }
}

// ---- one piece: end states -> the six coefficients (ascending powers, the layout of minco.hpp:402,545) --------------------
MPCR_HD void hermite(double T, double h, double p0, double v0, double a0, double p1, double v1, double a1, double c[6]) {
    const double dlt = p1 - p0 - (v0 + 0.5 * a0 * T) * T, dv = (v1 - v0 - a0 * T) * T, da = (a1 - a0) * (T * T);
    const double h3 = h * h * h;
    c[0] = p0; c[1] = v0; c[2] = 0.5 * a0;
    c[3] = (10.0 * dlt - 4.0 * dv + 0.5 * da) * h3;
    c[4] = (-15.0 * dlt + 7.0 * dv - da) * (h3 * h);
    c[5] = (6.0 * dlt - 3.0 * dv + 0.5 * da) * (h3 * h * h);
}

// jerk energy of a piece and its partials (minco.hpp:584-654, one piece, one axis at a time is NOT how the reference sums it:
// row_dot runs over the axes first - the same association is kept here)
MPCR_HD double piece_energy(double T, const double c[3][6], double gc[3][6], double &gT) {
    const double t1 = T, t2 = t1 * t1, t3 = t2 * t1, t4 = t2 * t2, t5 = t4 * t1;
    double d33 = 0, d43 = 0, d44 = 0, d53 = 0, d54 = 0, d55 = 0;
    for (int d = 0; d < 3; d++) {
        d33 += c[d][3] * c[d][3]; d43 += c[d][4] * c[d][3]; d44 += c[d][4] * c[d][4];
        d53 += c[d][5] * c[d][3]; d54 += c[d][5] * c[d][4]; d55 += c[d][5] * c[d][5];
        gc[d][0] = gc[d][1] = gc[d][2] = 0.0;
        gc[d][3] = 72.0 * c[d][3] * t1 + 144.0 * c[d][4] * t2 + 240.0 * c[d][5] * t3;
        gc[d][4] = 144.0 * c[d][3] * t2 + 384.0 * c[d][4] * t3 + 720.0 * c[d][5] * t4;
        gc[d][5] = 240.0 * c[d][3] * t3 + 720.0 * c[d][4] * t4 + 1440.0 * c[d][5] * t5;
    }
    gT = 36.0 * d33 + 288.0 * d43 * t1 + 576.0 * d44 * t2 + 720.0 * d53 * t2 + 2880.0 * d54 * t3 + 3600.0 * d55 * t4;
    return 36.0 * d33 * t1 + 144.0 * d43 * t2 + 192.0 * d44 * t3 + 240.0 * d53 * t3 + 720.0 * d54 * t4 + 720.0 * d55 * t5;
}

// the same, ONE axis at a time (the device's axis-per-thread form): the axes' shares are added afterwards, so the sums associate
// differently from the reference's row_dot - equal to rounding
MPCR_HD double piece_energy1(double T, const double c[6], double gc[6], double &gT) {
    const double t1 = T, t2 = t1 * t1, t3 = t2 * t1, t4 = t2 * t2, t5 = t4 * t1;
    const double d33 = c[3] * c[3], d43 = c[4] * c[3], d44 = c[4] * c[4], d53 = c[5] * c[3], d54 = c[5] * c[4], d55 = c[5] * c[5];
    gc[0] = gc[1] = gc[2] = 0.0;
    gc[3] = 72.0 * c[3] * t1 + 144.0 * c[4] * t2 + 240.0 * c[5] * t3;
    gc[4] = 144.0 * c[3] * t2 + 384.0 * c[4] * t3 + 720.0 * c[5] * t4;
    gc[5] = 240.0 * c[3] * t3 + 720.0 * c[4] * t4 + 1440.0 * c[5] * t5;
    gT = 36.0 * d33 + 288.0 * d43 * t1 + 576.0 * d44 * t2 + 720.0 * d53 * t2 + 2880.0 * d54 * t3 + 3600.0 * d55 * t4;
    return 36.0 * d33 * t1 + 144.0 * d43 * t2 + 192.0 * d44 * t3 + 240.0 * d53 * t3 + 720.0 * d54 * t4 + 720.0 * d55 * t5;
}

// ---- adjoint of one piece ------------------------------------------------------------------------------------------------
// G[6] = dCost/dc of the piece (one axis) -> dCost/d(p0, v0, a0, p1, v1, a1) at fixed T (transpose of `hermite`)
MPCR_HD void piece_adjoint(double h, const double G[6], double ge[6]) {
    const double h2 = h * h, h3 = h2 * h, h4 = h2 * h2, h5 = h4 * h;
    const double gp = 10.0 * h3 * G[3] - 15.0 * h4 * G[4] + 6.0 * h5 * G[5];
    ge[0] = G[0] - gp;
    ge[1] = G[1] - 6.0 * h2 * G[3] + 8.0 * h3 * G[4] - 3.0 * h4 * G[5];
    ge[2] = 0.5 * G[2] - 1.5 * h * G[3] + 1.5 * h2 * G[4] - 0.5 * h3 * G[5];
    ge[3] = gp;
    ge[4] = -4.0 * h2 * G[3] + 7.0 * h3 * G[4] - 3.0 * h4 * G[5];
    ge[5] = 0.5 * h * G[3] - h2 * G[4] + 0.5 * h3 * G[5];
}
// dCost/dT of the piece (one axis), everything but the caller's partial: the coefficients' own dependence on T at fixed end
// states (G) plus the junction equations' (mu_l = adjoint state at the piece's left waypoint, mu_r at its right; zero at the
// trajectory's ends).  dp = p1 - p0.
MPCR_HD double piece_grad_T(double h, const double G[6], double dp, double v0, double a0, double v1, double a1,
                            const double mu_l[2], const double mu_r[2]) {
    const double h2 = h * h, h3 = h2 * h, h4 = h2 * h2;
    // d(c3, c4, c5)/dh at fixed end states
    const double c3h = 30.0 * h2 * dp - 12.0 * h * v0 - 1.5 * a0 - 8.0 * h * v1 + 0.5 * a1;
    const double c4h = -60.0 * h3 * dp + 24.0 * h2 * v0 + 3.0 * h * a0 + 21.0 * h2 * v1 - 2.0 * h * a1;
    const double c5h = 30.0 * h4 * dp - 12.0 * h3 * v0 - 1.5 * h2 * a0 - 12.0 * h3 * v1 + 1.5 * h2 * a1;
    // d/dh of the piece's jerk / snap at its two ends
    const double j0h = 180.0 * h2 * dp - 72.0 * h * v0 - 9.0 * a0 - 48.0 * h * v1 + 3.0 * a1;
    const double jTh = 180.0 * h2 * dp - 48.0 * h * v0 - 3.0 * a0 - 72.0 * h * v1 + 9.0 * a1;
    const double s0h = -1440.0 * h3 * dp + 576.0 * h2 * v0 + 72.0 * h * a0 + 504.0 * h2 * v1 - 48.0 * h * a1;
    const double sTh = 1440.0 * h3 * dp - 504.0 * h2 * v0 - 48.0 * h * a0 - 576.0 * h2 * v1 + 72.0 * h * a1;
    const double direct = G[3] * c3h + G[4] * c4h + G[5] * c5h;
    const double implicit = mu_l[0] * s0h - mu_l[1] * j0h - mu_r[0] * sTh + mu_r[1] * jTh;
    return h2 * (implicit - direct);
}
// dCost/d(p1 - p0) through the junction equations (one axis): the waypoint gradient gets +w of the piece on its left, -w of
// the piece on its right
MPCR_HD double piece_w(double h, const double mu_l[2], const double mu_r[2]) {
    const double h3 = h * h * h;
    return 360.0 * h3 * h * (mu_l[0] + mu_r[0]) + 60.0 * h3 * (mu_l[1] - mu_r[1]);
}

// tau <-> T of the optimizer (back_end_optimizer.hpp:216-300)
MPCR_HD double tau_to_T(double tau) { return tau > 0.0 ? ((0.5 * tau + 1.0) * tau + 1.0) : 1.0 / ((0.5 * tau - 1.0) * tau + 1.0); }
// T and 1 / T together (one reciprocal): the device's form; 1 ulp off the quotient above for tau <= 0
MPCR_HD void tau_to_T_h(double tau, double &T, double &h) {
    if (tau > 0.0) { T = (0.5 * tau + 1.0) * tau + 1.0; h = rcp(T); }
    else { h = (0.5 * tau - 1.0) * tau + 1.0; T = rcp(h); }
}
MPCR_HD double grad_T_to_tau(double tau, double gT) {
    if (tau > 0) return gT * (tau + 1.0);
    const double den = (0.5 * tau - 1.0) * tau + 1.0;
    return gT * (1.0 - tau) / (den * den);
}

}  // namespace mpcr
