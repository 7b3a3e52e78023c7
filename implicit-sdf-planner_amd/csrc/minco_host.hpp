// Host side of the full objective callback: MINCO (s = 3, non-uniform time) as TrajOptimizer::costFunctionLmbm uses
// it around the two device sweeps (back_end_optimizer.hpp:358-430):
//   minco.setParameters / getEnergy / getEnergyPartialGradByCoeffs / getEnergyPartialGradByTimes / propogateGrad
//   (src/utils/include/utils/minco.hpp:397-655) on the banded LU of :43-198 (no pivoting, 6 sub- and super-diagonals).
// This is the HOST form of the callback's MINCO half (isdf_set_minco_mode(ctx, 1), and the library's own choice for a single
// trajectory of <= 64 pieces on the tile sweep): the reference's banded factorisation, a chain of 6N dependent pivots that one
// CPU core walks in ~10 us at N = 40 - written so that every element sees the same sequence of floating-point operations as the
// reference's loops (same elimination order, same skipped zeros).  The DEVICE form (csrc/minco_dev.hip, minco_pcr.hpp) solves the
// same trajectory in its junction states by parallel cyclic reduction and agrees with this one to rounding (1e-10 relative).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <vector>

namespace isdf_host {

// Band storage of an n x n matrix with `lo` sub- and `up` super-diagonals: element (i, j) at d[(i - j + up) * n + j].
class BandLU {
public:
    void create(int n_, int lo_, int up_) { n = n_; lo = lo_; up = up_; d.assign((size_t)n * (lo + up + 1), 0.0); }
    void clear() { std::fill(d.begin(), d.end(), 0.0); }
    double &operator()(int i, int j) { return d[(size_t)(i - j + up) * n + j]; }
    double operator()(int i, int j) const { return d[(size_t)(i - j + up) * n + j]; }

    // in-place LU without pivoting: column k is scaled by the pivot, then the trailing band gets the rank-1 update
    void factorize() {
        BandLU &A = *this;
        for (int k = 0; k + 1 < n; k++) {
            const int i_end = std::min(k + lo, n - 1), j_end = std::min(k + up, n - 1);
            const double piv = A(k, k);
            for (int i = k + 1; i <= i_end; i++)
                if (A(i, k) != 0.0) A(i, k) /= piv;
            for (int j = k + 1; j <= j_end; j++) {
                const double ukj = A(k, j);
                if (ukj == 0.0) continue;
                for (int i = k + 1; i <= i_end; i++)
                    if (A(i, k) != 0.0) A(i, j) -= A(i, k) * ukj;
            }
        }
    }
    // x: n x m, column-major with leading dimension n (Eigen::MatrixX3d); solves A x = b in place
    void solve(double *x, int m) const {
        const BandLU &A = *this;
        for (int j = 0; j < n; j++) {
            const int i_end = std::min(j + lo, n - 1);
            for (int i = j + 1; i <= i_end; i++) {
                const double l = A(i, j);
                if (l != 0.0)
                    for (int c = 0; c < m; c++) x[(size_t)c * n + i] -= l * x[(size_t)c * n + j];
            }
        }
        for (int j = n - 1; j >= 0; j--) {
            const double piv = A(j, j);
            for (int c = 0; c < m; c++) x[(size_t)c * n + j] /= piv;
            for (int i = std::max(0, j - up); i < j; i++) {
                const double u = A(i, j);
                if (u != 0.0)
                    for (int c = 0; c < m; c++) x[(size_t)c * n + i] -= u * x[(size_t)c * n + j];
            }
        }
    }
    // solves A^T x = b in place
    void solve_transposed(double *x, int m) const {
        const BandLU &A = *this;
        for (int j = 0; j < n; j++) {
            const double piv = A(j, j);
            for (int c = 0; c < m; c++) x[(size_t)c * n + j] /= piv;
            const int i_end = std::min(j + up, n - 1);
            for (int i = j + 1; i <= i_end; i++) {
                const double u = A(j, i);
                if (u != 0.0)
                    for (int c = 0; c < m; c++) x[(size_t)c * n + i] -= u * x[(size_t)c * n + j];
            }
        }
        for (int j = n - 1; j >= 0; j--)
            for (int i = std::max(0, j - lo); i < j; i++) {
                const double l = A(j, i);
                if (l != 0.0)
                    for (int c = 0; c < m; c++) x[(size_t)c * n + i] -= l * x[(size_t)c * n + j];
            }
    }

private:
    int n = 0, lo = 0, up = 0;
    std::vector<double> d;
};

// MINCO_S3NU: minimum-jerk... (s = 3: quintic pieces) trajectory through fixed waypoints with free durations.
class MincoS3 {
public:
    int N = 0;
    std::vector<double> c;               // coefficient matrix b: 6N x 3, column-major, rows 6i..6i+5 = c0..c5 of piece i

    // head / tail: 3x3 column-major, columns = position, velocity, acceleration (Eigen::Matrix3d of setConditions)
    void set_conditions(const double *head9, const double *tail9, int pieces) {
        N = pieces;
        std::copy(head9, head9 + 9, head);
        std::copy(tail9, tail9 + 9, tail);
        A.create(6 * N, 6, 6);
        c.assign((size_t)18 * N, 0.0);
        t1.assign(N, 0.0); t2 = t1; t3 = t1; t4 = t1; t5 = t1;
    }
    double &C(int r, int col) { return c[(size_t)col * 6 * N + r]; }
    double C(int r, int col) const { return c[(size_t)col * 6 * N + r]; }

    // waypoints: 3 x (N-1) column-major (point i at waypoints[3i..3i+2]); durations: N
    void set_parameters(const double *waypoints, const double *durations) {
        for (int i = 0; i < N; i++) {
            t1[i] = durations[i];
            t2[i] = t1[i] * t1[i];
            t3[i] = t2[i] * t1[i];
            t4[i] = t2[i] * t2[i];
            t5[i] = t4[i] * t1[i];
        }
        A.clear();
        std::fill(c.begin(), c.end(), 0.0);
        // start: position, velocity, acceleration of piece 0 at t = 0
        A(0, 0) = 1.0; A(1, 1) = 1.0; A(2, 2) = 2.0;
        for (int d = 0; d < 3; d++) { C(0, d) = head[d]; C(1, d) = head[3 + d]; C(2, d) = head[6 + d]; }
        for (int i = 0; i + 1 < N; i++) {
            const int r = 6 * i, q = 6 * i;       // equation rows r+3..r+8, unknown columns q..q+11
            const double T1 = t1[i], T2 = t2[i], T3 = t3[i], T4 = t4[i], T5 = t5[i];
            // continuity of the 3rd and 4th derivative at the junction
            A(r + 3, q + 3) = 6.0;  A(r + 3, q + 4) = 24.0 * T1;  A(r + 3, q + 5) = 60.0 * T2;  A(r + 3, q + 9) = -6.0;
            A(r + 4, q + 4) = 24.0; A(r + 4, q + 5) = 120.0 * T1; A(r + 4, q + 10) = -24.0;
            // piece i ends at the waypoint
            A(r + 5, q) = 1.0; A(r + 5, q + 1) = T1; A(r + 5, q + 2) = T2; A(r + 5, q + 3) = T3; A(r + 5, q + 4) = T4; A(r + 5, q + 5) = T5;
            // continuity of position, velocity, acceleration
            A(r + 6, q) = 1.0; A(r + 6, q + 1) = T1; A(r + 6, q + 2) = T2; A(r + 6, q + 3) = T3; A(r + 6, q + 4) = T4; A(r + 6, q + 5) = T5;
            A(r + 6, q + 6) = -1.0;
            A(r + 7, q + 1) = 1.0; A(r + 7, q + 2) = 2 * T1; A(r + 7, q + 3) = 3 * T2; A(r + 7, q + 4) = 4 * T3; A(r + 7, q + 5) = 5 * T4;
            A(r + 7, q + 7) = -1.0;
            A(r + 8, q + 2) = 2.0; A(r + 8, q + 3) = 6 * T1; A(r + 8, q + 4) = 12 * T2; A(r + 8, q + 5) = 20 * T3;
            A(r + 8, q + 8) = -2.0;
            for (int d = 0; d < 3; d++) C(r + 5, d) = waypoints[3 * i + d];
        }
        {   // end: position, velocity, acceleration of the last piece at t = T
            const int e = 6 * N, l = N - 1;
            A(e - 3, e - 6) = 1.0; A(e - 3, e - 5) = t1[l]; A(e - 3, e - 4) = t2[l]; A(e - 3, e - 3) = t3[l]; A(e - 3, e - 2) = t4[l]; A(e - 3, e - 1) = t5[l];
            A(e - 2, e - 5) = 1.0; A(e - 2, e - 4) = 2 * t1[l]; A(e - 2, e - 3) = 3 * t2[l]; A(e - 2, e - 2) = 4 * t3[l]; A(e - 2, e - 1) = 5 * t4[l];
            A(e - 1, e - 4) = 2; A(e - 1, e - 3) = 6 * t1[l]; A(e - 1, e - 2) = 12 * t2[l]; A(e - 1, e - 1) = 20 * t3[l];
            for (int d = 0; d < 3; d++) { C(e - 3, d) = tail[d]; C(e - 2, d) = tail[3 + d]; C(e - 1, d) = tail[6 + d]; }
        }
        A.factorize();
        A.solve(c.data(), 3);
    }

    double row_dot(int ra, int rb) const { return C(ra, 0) * C(rb, 0) + C(ra, 1) * C(rb, 1) + C(ra, 2) * C(rb, 2); }

    // jerk energy and its partials; grad_c (6N x 3 col-major) and grad_t (N) are OVERWRITTEN
    double energy(double *grad_c, double *grad_t) const {
        double e = 0.0;
        for (int i = 0; i < N; i++) {
            const int r = 6 * i;
            e += 36.0 * row_dot(r + 3, r + 3) * t1[i] + 144.0 * row_dot(r + 4, r + 3) * t2[i] + 192.0 * row_dot(r + 4, r + 4) * t3[i] +
                 240.0 * row_dot(r + 5, r + 3) * t3[i] + 720.0 * row_dot(r + 5, r + 4) * t4[i] + 720.0 * row_dot(r + 5, r + 5) * t5[i];
            for (int d = 0; d < 3; d++) {
                double *g = grad_c + (size_t)d * 6 * N + r;
                g[5] = 240.0 * C(r + 3, d) * t3[i] + 720.0 * C(r + 4, d) * t4[i] + 1440.0 * C(r + 5, d) * t5[i];
                g[4] = 144.0 * C(r + 3, d) * t2[i] + 384.0 * C(r + 4, d) * t3[i] + 720.0 * C(r + 5, d) * t4[i];
                g[3] = 72.0 * C(r + 3, d) * t1[i] + 144.0 * C(r + 4, d) * t2[i] + 240.0 * C(r + 5, d) * t3[i];
                g[0] = g[1] = g[2] = 0.0;
            }
            grad_t[i] = 36.0 * row_dot(r + 3, r + 3) + 288.0 * row_dot(r + 4, r + 3) * t1[i] + 576.0 * row_dot(r + 4, r + 4) * t2[i] +
                        720.0 * row_dot(r + 5, r + 3) * t2[i] + 2880.0 * row_dot(r + 5, r + 4) * t3[i] + 3600.0 * row_dot(r + 5, r + 5) * t4[i];
        }
        return e;
    }

    // dCost/d(c, T) -> dCost/d(waypoints, T).  grad_c: 6N x 3 col-major (not modified), grad_t_partial: N.
    // grad_points: 3 x (N-1) col-major, grad_t: N.
    void propagate_grad(const double *grad_c, const double *grad_t_partial, double *grad_points, double *grad_t) {
        adj.assign(grad_c, grad_c + (size_t)18 * N);
        A.solve_transposed(adj.data(), 3);
        const int ld = 6 * N;
        for (int i = 0; i + 1 < N; i++)
            for (int d = 0; d < 3; d++) grad_points[3 * i + d] = adj[(size_t)d * ld + 6 * i + 5];
        for (int i = 0; i + 1 < N; i++) {
            const int r = 6 * i;
            double s = 0.0;
            for (int d = 0; d < 3; d++) {             // column-major walk of the 6 x 3 block, like Eigen's .sum()
                const double negvel = -(C(r + 1, d) + 2.0 * t1[i] * C(r + 2, d) + 3.0 * t2[i] * C(r + 3, d) + 4.0 * t3[i] * C(r + 4, d) + 5.0 * t4[i] * C(r + 5, d));
                const double negacc = -(2.0 * C(r + 2, d) + 6.0 * t1[i] * C(r + 3, d) + 12.0 * t2[i] * C(r + 4, d) + 20.0 * t3[i] * C(r + 5, d));
                const double negjer = -(6.0 * C(r + 3, d) + 24.0 * t1[i] * C(r + 4, d) + 60.0 * t2[i] * C(r + 5, d));
                const double negsna = -(24.0 * C(r + 4, d) + 120.0 * t1[i] * C(r + 5, d));
                const double negcra = -120.0 * C(r + 5, d);
                const double *a = adj.data() + (size_t)d * ld + r + 3;
                s += negsna * a[0]; s += negcra * a[1]; s += negvel * a[2]; s += negvel * a[3]; s += negacc * a[4]; s += negjer * a[5];
            }
            grad_t[i] = s;
        }
        {
            const int e = 6 * N, l = N - 1;
            double s = 0.0;
            for (int d = 0; d < 3; d++) {
                const double negvel = -(C(e - 5, d) + 2.0 * t1[l] * C(e - 4, d) + 3.0 * t2[l] * C(e - 3, d) + 4.0 * t3[l] * C(e - 2, d) + 5.0 * t4[l] * C(e - 1, d));
                const double negacc = -(2.0 * C(e - 4, d) + 6.0 * t1[l] * C(e - 3, d) + 12.0 * t2[l] * C(e - 2, d) + 20.0 * t3[l] * C(e - 1, d));
                const double negjer = -(6.0 * C(e - 3, d) + 24.0 * t1[l] * C(e - 2, d) + 60.0 * t2[l] * C(e - 1, d));
                const double *a = adj.data() + (size_t)d * ld + e - 3;
                s += negvel * a[0]; s += negacc * a[1]; s += negjer * a[2];
            }
            grad_t[l] = s;
        }
        for (int i = 0; i < N; i++) grad_t[i] += grad_t_partial[i];
    }

private:
    double head[9] = {0}, tail[9] = {0};
    BandLU A;
    std::vector<double> t1, t2, t3, t4, t5, adj;
};

// tau <-> T diffeomorphism of the optimizer (back_end_optimizer.hpp:216-300)
inline double tau_to_T(double tau) { return tau > 0.0 ? ((0.5 * tau + 1.0) * tau + 1.0) : 1.0 / ((0.5 * tau - 1.0) * tau + 1.0); }
inline double T_to_tau(double T) { return T > 1.0 ? (std::sqrt(2.0 * T - 1.0) - 1.0) : (1.0 - std::sqrt(2.0 / T - 1.0)); }
inline double grad_T_to_tau(double tau, double gT) {
    if (tau > 0) return gT * (tau + 1.0);
    const double den = (0.5 * tau - 1.0) * tau + 1.0;
    return gT * (1.0 - tau) / (den * den);
}

} // namespace isdf_host
