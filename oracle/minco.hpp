// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked into the product path.
// CPU restatement of MINCO (s = 3, non-uniform time) — the step either side of the hot path:
//   Banded               src/utils/include/utils/minco.hpp:43-198  (banded LU without pivoting, solve, adjoint solve)
//   Minco::set_parameters src/utils/include/utils/minco.hpp:433-513
//   minco_energy          src/utils/include/utils/minco.hpp:530-582 (energy, dE/dc, dE/dT)
//   Minco::propagate_grad src/utils/include/utils/minco.hpp:584-654
// Matrices are 6N x 3 column-major like Eigen::MatrixX3d.
#pragma once
#include <vector>
#include <algorithm>
#include <cmath>

namespace orc {

struct Banded {
    int N = 0, lo = 0, up = 0;
    std::vector<double> d;
    void create(int n, int p, int q) { N = n; lo = p; up = q; d.assign((size_t)N * (lo + up + 1), 0.0); }
    void reset() { std::fill(d.begin(), d.end(), 0.0); }
    double &at(int i, int j) { return d[(size_t)(i - j + up) * N + j]; }
    double at(int i, int j) const { return d[(size_t)(i - j + up) * N + j]; }
    void factorize() {
        for (int k = 0; k <= N - 2; k++) {
            const int iM = std::min(k + lo, N - 1);
            double cVl = at(k, k);
            for (int i = k + 1; i <= iM; i++)
                if (at(i, k) != 0.0) at(i, k) /= cVl;
            const int jM = std::min(k + up, N - 1);
            for (int j = k + 1; j <= jM; j++) {
                cVl = at(k, j);
                if (cVl != 0.0)
                    for (int i = k + 1; i <= iM; i++)
                        if (at(i, k) != 0.0) at(i, j) -= at(i, k) * cVl;
            }
        }
    }
    // b: N x m column-major (leading dimension N)
    void solve(double *b, int m) const {
        for (int j = 0; j <= N - 1; j++) {
            const int iM = std::min(j + lo, N - 1);
            for (int i = j + 1; i <= iM; i++)
                if (at(i, j) != 0.0)
                    for (int c = 0; c < m; c++) b[c * N + i] -= at(i, j) * b[c * N + j];
        }
        for (int j = N - 1; j >= 0; j--) {
            for (int c = 0; c < m; c++) b[c * N + j] /= at(j, j);
            const int iM = std::max(0, j - up);
            for (int i = iM; i <= j - 1; i++)
                if (at(i, j) != 0.0)
                    for (int c = 0; c < m; c++) b[c * N + i] -= at(i, j) * b[c * N + j];
        }
    }
    void solve_adj(double *b, int m) const {
        for (int j = 0; j <= N - 1; j++) {
            for (int c = 0; c < m; c++) b[c * N + j] /= at(j, j);
            const int iM = std::min(j + up, N - 1);
            for (int i = j + 1; i <= iM; i++)
                if (at(j, i) != 0.0)
                    for (int c = 0; c < m; c++) b[c * N + i] -= at(j, i) * b[c * N + j];
        }
        for (int j = N - 1; j >= 0; j--) {
            const int iM = std::max(0, j - lo);
            for (int i = iM; i <= j - 1; i++)
                if (at(j, i) != 0.0)
                    for (int c = 0; c < m; c++) b[c * N + i] -= at(j, i) * b[c * N + j];
        }
    }
};

struct Minco {
    int N = 0;
    double head[9], tail[9];          // 3x3 column-major: col0 = pos, col1 = vel, col2 = acc
    Banded A;
    std::vector<double> b;            // 6N x 3 column-major
    std::vector<double> T1, T2, T3, T4, T5;
    double &B(int r, int c) { return b[(size_t)c * 6 * N + r]; }
    double B(int r, int c) const { return b[(size_t)c * 6 * N + r]; }

    void set_conditions(const double *h, const double *t, int n) {
        N = n;
        std::copy(h, h + 9, head);
        std::copy(t, t + 9, tail);
        A.create(6 * N, 6, 6);
        b.assign((size_t)18 * N, 0.0);
        T1.resize(N); T2.resize(N); T3.resize(N); T4.resize(N); T5.resize(N);
    }
    // inPs: 3 x (N-1) column-major
    void set_parameters(const double *inPs, const double *ts) {
        for (int i = 0; i < N; i++) {
            T1[i] = ts[i]; T2[i] = T1[i] * T1[i]; T3[i] = T2[i] * T1[i]; T4[i] = T2[i] * T2[i]; T5[i] = T4[i] * T1[i];
        }
        A.reset();
        std::fill(b.begin(), b.end(), 0.0);
        A.at(0, 0) = 1.0; A.at(1, 1) = 1.0; A.at(2, 2) = 2.0;
        for (int c = 0; c < 3; c++) { B(0, c) = head[0 * 3 + c]; B(1, c) = head[1 * 3 + c]; B(2, c) = head[2 * 3 + c]; }
        for (int i = 0; i < N - 1; i++) {
            A.at(6 * i + 3, 6 * i + 3) = 6.0;
            A.at(6 * i + 3, 6 * i + 4) = 24.0 * T1[i];
            A.at(6 * i + 3, 6 * i + 5) = 60.0 * T2[i];
            A.at(6 * i + 3, 6 * i + 9) = -6.0;
            A.at(6 * i + 4, 6 * i + 4) = 24.0;
            A.at(6 * i + 4, 6 * i + 5) = 120.0 * T1[i];
            A.at(6 * i + 4, 6 * i + 10) = -24.0;
            A.at(6 * i + 5, 6 * i) = 1.0;
            A.at(6 * i + 5, 6 * i + 1) = T1[i];
            A.at(6 * i + 5, 6 * i + 2) = T2[i];
            A.at(6 * i + 5, 6 * i + 3) = T3[i];
            A.at(6 * i + 5, 6 * i + 4) = T4[i];
            A.at(6 * i + 5, 6 * i + 5) = T5[i];
            A.at(6 * i + 6, 6 * i) = 1.0;
            A.at(6 * i + 6, 6 * i + 1) = T1[i];
            A.at(6 * i + 6, 6 * i + 2) = T2[i];
            A.at(6 * i + 6, 6 * i + 3) = T3[i];
            A.at(6 * i + 6, 6 * i + 4) = T4[i];
            A.at(6 * i + 6, 6 * i + 5) = T5[i];
            A.at(6 * i + 6, 6 * i + 6) = -1.0;
            A.at(6 * i + 7, 6 * i + 1) = 1.0;
            A.at(6 * i + 7, 6 * i + 2) = 2 * T1[i];
            A.at(6 * i + 7, 6 * i + 3) = 3 * T2[i];
            A.at(6 * i + 7, 6 * i + 4) = 4 * T3[i];
            A.at(6 * i + 7, 6 * i + 5) = 5 * T4[i];
            A.at(6 * i + 7, 6 * i + 7) = -1.0;
            A.at(6 * i + 8, 6 * i + 2) = 2.0;
            A.at(6 * i + 8, 6 * i + 3) = 6 * T1[i];
            A.at(6 * i + 8, 6 * i + 4) = 12 * T2[i];
            A.at(6 * i + 8, 6 * i + 5) = 20 * T3[i];
            A.at(6 * i + 8, 6 * i + 8) = -2.0;
            for (int c = 0; c < 3; c++) B(6 * i + 5, c) = inPs[3 * i + c];
        }
        const int n = N - 1;
        A.at(6 * N - 3, 6 * N - 6) = 1.0;
        A.at(6 * N - 3, 6 * N - 5) = T1[n];
        A.at(6 * N - 3, 6 * N - 4) = T2[n];
        A.at(6 * N - 3, 6 * N - 3) = T3[n];
        A.at(6 * N - 3, 6 * N - 2) = T4[n];
        A.at(6 * N - 3, 6 * N - 1) = T5[n];
        A.at(6 * N - 2, 6 * N - 5) = 1.0;
        A.at(6 * N - 2, 6 * N - 4) = 2 * T1[n];
        A.at(6 * N - 2, 6 * N - 3) = 3 * T2[n];
        A.at(6 * N - 2, 6 * N - 2) = 4 * T3[n];
        A.at(6 * N - 2, 6 * N - 1) = 5 * T4[n];
        A.at(6 * N - 1, 6 * N - 4) = 2;
        A.at(6 * N - 1, 6 * N - 3) = 6 * T1[n];
        A.at(6 * N - 1, 6 * N - 2) = 12 * T2[n];
        A.at(6 * N - 1, 6 * N - 1) = 20 * T3[n];
        for (int c = 0; c < 3; c++) {
            B(6 * N - 3, c) = tail[0 * 3 + c]; B(6 * N - 2, c) = tail[1 * 3 + c]; B(6 * N - 1, c) = tail[2 * 3 + c];
        }
        A.factorize();
        A.solve(b.data(), 3);
    }

    // gradC: 6N x 3 col-major partials; gradT_partial[N]; outputs gradP 3 x (N-1) col-major, gradT[N]
    void propagate_grad(const double *gradC, const double *gradT_partial, double *gradP, double *gradT) const {
        std::vector<double> adj(gradC, gradC + (size_t)18 * N);
        A.solve_adj(adj.data(), 3);
        auto ADJ = [&](int r, int c) { return adj[(size_t)c * 6 * N + r]; };
        for (int i = 0; i < N - 1; i++)
            for (int c = 0; c < 3; c++) gradP[3 * i + c] = ADJ(6 * i + 5, c);
        for (int i = 0; i < N - 1; i++) {
            double s = 0.0;
            // Eigen's cwiseProduct(...).sum() over a 6x3 block walks column-major: c outer, r inner
            for (int c = 0; c < 3; c++) {
                double B1[6];
                B1[2] = -(B(i * 6 + 1, c) + 2.0 * T1[i] * B(i * 6 + 2, c) + 3.0 * T2[i] * B(i * 6 + 3, c) +
                          4.0 * T3[i] * B(i * 6 + 4, c) + 5.0 * T4[i] * B(i * 6 + 5, c));
                B1[3] = B1[2];
                B1[4] = -(2.0 * B(i * 6 + 2, c) + 6.0 * T1[i] * B(i * 6 + 3, c) + 12.0 * T2[i] * B(i * 6 + 4, c) +
                          20.0 * T3[i] * B(i * 6 + 5, c));
                B1[5] = -(6.0 * B(i * 6 + 3, c) + 24.0 * T1[i] * B(i * 6 + 4, c) + 60.0 * T2[i] * B(i * 6 + 5, c));
                B1[0] = -(24.0 * B(i * 6 + 4, c) + 120.0 * T1[i] * B(i * 6 + 5, c));
                B1[1] = -120.0 * B(i * 6 + 5, c);
                for (int r = 0; r < 6; r++) s += B1[r] * ADJ(6 * i + 3 + r, c);
            }
            gradT[i] = s;
        }
        {
            const int n = N - 1;
            double s = 0.0;
            for (int c = 0; c < 3; c++) {
                double B2[3];
                B2[0] = -(B(6 * N - 5, c) + 2.0 * T1[n] * B(6 * N - 4, c) + 3.0 * T2[n] * B(6 * N - 3, c) +
                          4.0 * T3[n] * B(6 * N - 2, c) + 5.0 * T4[n] * B(6 * N - 1, c));
                B2[1] = -(2.0 * B(6 * N - 4, c) + 6.0 * T1[n] * B(6 * N - 3, c) + 12.0 * T2[n] * B(6 * N - 2, c) +
                          20.0 * T3[n] * B(6 * N - 1, c));
                B2[2] = -(6.0 * B(6 * N - 3, c) + 24.0 * T1[n] * B(6 * N - 2, c) + 60.0 * T2[n] * B(6 * N - 1, c));
                for (int r = 0; r < 3; r++) s += B2[r] * ADJ(6 * N - 3 + r, c);
            }
            gradT[n] = s;
        }
        for (int i = 0; i < N; i++) gradT[i] += gradT_partial[i];
    }
};

// energy (minco.hpp:530-543), dE/dc (:550-566), dE/dT (:568-582).  gdC / gdT may be NULL.
inline double minco_energy(int N, const double *C, const double *T, double *gdC, double *gdT) {
    auto B = [&](int r, int c) { return C[(size_t)c * 6 * N + r]; };
    auto rdot = [&](int r1, int r2) { return B(r1, 0) * B(r2, 0) + B(r1, 1) * B(r2, 1) + B(r1, 2) * B(r2, 2); };
    double energy = 0.0;
    for (int i = 0; i < N; i++) {
        const double t1 = T[i], t2 = t1 * t1, t3 = t2 * t1, t4 = t2 * t2, t5 = t4 * t1;
        energy += 36.0 * rdot(6 * i + 3, 6 * i + 3) * t1 + 144.0 * rdot(6 * i + 4, 6 * i + 3) * t2 +
                  192.0 * rdot(6 * i + 4, 6 * i + 4) * t3 + 240.0 * rdot(6 * i + 5, 6 * i + 3) * t3 +
                  720.0 * rdot(6 * i + 5, 6 * i + 4) * t4 + 720.0 * rdot(6 * i + 5, 6 * i + 5) * t5;
        if (gdC) {
            for (int c = 0; c < 3; c++) {
                double *g = gdC + (size_t)c * 6 * N + 6 * i;
                g[5] = 240.0 * B(6 * i + 3, c) * t3 + 720.0 * B(6 * i + 4, c) * t4 + 1440.0 * B(6 * i + 5, c) * t5;
                g[4] = 144.0 * B(6 * i + 3, c) * t2 + 384.0 * B(6 * i + 4, c) * t3 + 720.0 * B(6 * i + 5, c) * t4;
                g[3] = 72.0 * B(6 * i + 3, c) * t1 + 144.0 * B(6 * i + 4, c) * t2 + 240.0 * B(6 * i + 5, c) * t3;
                g[0] = g[1] = g[2] = 0.0;
            }
        }
        if (gdT) {
            gdT[i] = 36.0 * rdot(6 * i + 3, 6 * i + 3) + 288.0 * rdot(6 * i + 4, 6 * i + 3) * t1 +
                     576.0 * rdot(6 * i + 4, 6 * i + 4) * t2 + 720.0 * rdot(6 * i + 5, 6 * i + 3) * t2 +
                     2880.0 * rdot(6 * i + 5, 6 * i + 4) * t3 + 3600.0 * rdot(6 * i + 5, 6 * i + 5) * t4;
        }
    }
    return energy;
}

} // namespace orc
