// Test shim: runs the PRODUCT's log-depth MINCO arithmetic (implicit-sdf-planner_amd/csrc/minco_pcr.hpp) on the CPU, one
// loop per parallel round - the same inline functions minco_dev.hip runs with one thread per row.  Built by
// tests/test_minco_pcr.py with g++.
#include "minco_pcr.hpp"
#include <cstddef>
#include <vector>
using std::size_t;
using namespace mpcr;

namespace {
struct Problem {
    int N;
    std::vector<double> T, h;
    std::vector<double> p;       // (N + 1) x 3
    std::vector<double> u;       // (N + 1) x 3 x 2: (v, a) per waypoint and axis
};
void setup(Problem &P, int N, const double *head9, const double *tail9, const double *way, const double *T) {
    P.N = N; P.T.assign(T, T + N); P.h.resize(N);
    for (int i = 0; i < N; i++) P.h[i] = 1.0 / T[i];
    P.p.assign((size_t)3 * (N + 1), 0.0); P.u.assign((size_t)6 * (N + 1), 0.0);
    for (int d = 0; d < 3; d++) {
        P.p[d] = head9[d]; P.p[3 * N + d] = tail9[d];
        P.u[d * 2 + 0] = head9[3 + d]; P.u[d * 2 + 1] = head9[6 + d];
        P.u[6 * N + d * 2 + 0] = tail9[3 + d]; P.u[6 * N + d * 2 + 1] = tail9[6 + d];
    }
    for (int i = 1; i < N; i++) for (int d = 0; d < 3; d++) P.p[3 * i + d] = way[3 * (i - 1) + d];
}
// solves K x = rhs for the n = N - 1 inner rows; rhs / x: [row][axis][2]
void solve(const Problem &P, std::vector<Row> &rows, std::vector<double> &x) {
    const int n = P.N - 1;
    std::vector<Norm> norm(n);
    for (int s = 1; s < n; s *= 2) {
        for (int j = 0; j < n; j++) norm[j] = pcr_normalise(rows[j]);
        for (int j = 0; j < n; j++) pcr_combine(rows[j], j - s >= 0 ? &norm[j - s] : nullptr, j + s < n ? &norm[j + s] : nullptr);
    }
    x.assign((size_t)6 * n, 0.0);
    for (int j = 0; j < n; j++) {
        double u[3][2];
        pcr_finish(rows[j], u);
        for (int d = 0; d < 3; d++) { x[6 * j + 2 * d] = u[d][0]; x[6 * j + 2 * d + 1] = u[d][1]; }
    }
}
void forward(Problem &P) {
    const int N = P.N, n = N - 1;
    if (n < 1) return;
    std::vector<Row> rows(n);
    for (int j = 1; j <= n; j++) {
        Row &w = rows[j - 1];
        const double hl = P.h[j - 1], hr = P.h[j];
        junction_blocks(hl, hr, w.L, w.D, w.U);
        double dpl[3], dpr[3];
        for (int d = 0; d < 3; d++) { dpl[d] = P.p[3 * j + d] - P.p[3 * (j - 1) + d]; dpr[d] = P.p[3 * (j + 1) + d] - P.p[3 * j + d]; }
        junction_rhs(hl, hr, dpl, dpr, w.r);
        if (j == 1) {
            double va[3][2];
            for (int d = 0; d < 3; d++) { va[d][0] = P.u[2 * d]; va[d][1] = P.u[2 * d + 1]; }
            rhs_minus(w.L, va, w.r);
            w.L = {0, 0, 0, 0};
        }
        if (j == n) {
            double va[3][2];
            for (int d = 0; d < 3; d++) { va[d][0] = P.u[6 * N + 2 * d]; va[d][1] = P.u[6 * N + 2 * d + 1]; }
            rhs_minus(w.U, va, w.r);
            w.U = {0, 0, 0, 0};
        }
    }
    std::vector<double> x;
    solve(P, rows, x);
    for (int j = 1; j <= n; j++) for (int k = 0; k < 6; k++) P.u[6 * j + k] = x[6 * (j - 1) + k];
}
}  // namespace

extern "C" {
// coeffs: 6N x 3 column-major; gdC likewise; gdT: N
void pcr_minco(int N, const double *head9, const double *tail9, const double *way, const double *T,
               double *coeffs, double *energy, double *gdC, double *gdT) {
    Problem P;
    setup(P, N, head9, tail9, way, T);
    forward(P);
    double e = 0.0;
    for (int k = 0; k < N; k++) {
        double c[3][6], gc[3][6], gT;
        for (int d = 0; d < 3; d++)
            hermite(P.T[k], P.h[k], P.p[3 * k + d], P.u[6 * k + 2 * d], P.u[6 * k + 2 * d + 1],
                    P.p[3 * (k + 1) + d], P.u[6 * (k + 1) + 2 * d], P.u[6 * (k + 1) + 2 * d + 1], c[d]);
        e += piece_energy(P.T[k], c, gc, gT);
        gdT[k] = gT;
        for (int d = 0; d < 3; d++) for (int r = 0; r < 6; r++) {
            coeffs[(size_t)d * 6 * N + 6 * k + r] = c[d][r];
            gdC[(size_t)d * 6 * N + 6 * k + r] = gc[d][r];
        }
    }
    *energy = e;
}
// gradP: 3 per inner waypoint (point-major), gradT: N
void pcr_minco_propagate(int N, const double *head9, const double *tail9, const double *way, const double *T,
                         const double *gradC, const double *gradT_partial, double *gradP, double *gradT) {
    Problem P;
    setup(P, N, head9, tail9, way, T);
    forward(P);
    const int n = N - 1;
    std::vector<double> ge((size_t)18 * N);          // [piece][axis][6]
    for (int k = 0; k < N; k++) for (int d = 0; d < 3; d++) {
        double G[6];
        for (int r = 0; r < 6; r++) G[r] = gradC[(size_t)d * 6 * N + 6 * k + r];
        piece_adjoint(P.h[k], G, &ge[(size_t)18 * k + 6 * d]);
    }
    std::vector<double> mu((size_t)6 * (N + 1), 0.0);   // [waypoint][axis][2], zero at the ends
    if (n >= 1) {
        std::vector<Row> rows(n);
        for (int j = 1; j <= n; j++) {
            Row &w = rows[j - 1];
            junction_blocks(P.h[j - 1], P.h[j], w.L, w.D, w.U);
            if (j == 1) w.L = {0, 0, 0, 0};
            if (j == n) w.U = {0, 0, 0, 0};
            for (int d = 0; d < 3; d++) {
                w.r[d][0] = ge[(size_t)18 * (j - 1) + 6 * d + 4] + ge[(size_t)18 * j + 6 * d + 1];
                w.r[d][1] = ge[(size_t)18 * (j - 1) + 6 * d + 5] + ge[(size_t)18 * j + 6 * d + 2];
            }
        }
        std::vector<double> x;
        solve(P, rows, x);
        for (int j = 1; j <= n; j++) for (int k = 0; k < 6; k++) mu[6 * j + k] = x[6 * (j - 1) + k];
    }
    std::vector<double> w((size_t)3 * N);
    for (int k = 0; k < N; k++) {
        double gt = 0.0;
        for (int d = 0; d < 3; d++) {
            double G[6];
            for (int r = 0; r < 6; r++) G[r] = gradC[(size_t)d * 6 * N + 6 * k + r];
            const double *ul = &P.u[6 * k + 2 * d], *ur = &P.u[6 * (k + 1) + 2 * d];
            const double dp = P.p[3 * (k + 1) + d] - P.p[3 * k + d];
            gt += piece_grad_T(P.h[k], G, dp, ul[0], ul[1], ur[0], ur[1], &mu[6 * k + 2 * d], &mu[6 * (k + 1) + 2 * d]);
            w[3 * k + d] = piece_w(P.h[k], &mu[6 * k + 2 * d], &mu[6 * (k + 1) + 2 * d]);
        }
        gradT[k] = gt + gradT_partial[k];
    }
    for (int j = 1; j <= n; j++) for (int d = 0; d < 3; d++)
        gradP[3 * (j - 1) + d] = ge[(size_t)18 * (j - 1) + 6 * d + 3] + ge[(size_t)18 * j + 6 * d + 0] + w[3 * (j - 1) + d] - w[3 * j + d];
}
}
