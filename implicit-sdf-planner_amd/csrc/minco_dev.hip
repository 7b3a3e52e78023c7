// Device half of the objective callback: everything TrajOptimizer::costFunctionLmbm does around the two
// sweeps (back_end_optimizer.hpp:358-430) - tau -> T, MINCO setParameters (minco.hpp:397-513), the jerk energy and its
// partials (:584-654), propogateGrad (:515-582), rho * sum(T), the chain rule back to (tau, waypoints) - as two small
// kernels either side of the sweeps, so that a callback moves n doubles down and n + 1 up:
//
//   cb_pre_kernel    x -> T, coefficients (straight into the sweeps' input arrays), energy + its partials, the junction states
//                    and the elimination's multipliers for the way back
//   (the sweeps)     accumulate [cost | gradT | gradC] per trajectory, as ever
//   cb_post_kernel   energy partials + the sweeps' sums -> adjoint solve -> (cost, g), written where the host waits for them
//
// One workgroup per trajectory, one THREAD per piece / junction.  The linear algebra is csrc/minco_pcr.hpp: the trajectory in
// its junction states (a symmetric positive definite block-tridiagonal system with 2 x 2 blocks) solved by parallel cyclic
// reduction - ceil(log2(N - 1)) rounds, against the 6N dependent pivots of the reference's band LU (which one wavefront walks in
// 140 us at N = 40: rounds 2-4, removed).  The matrix depends on the durations only: the way forward stores every round's two
// multiplier blocks per row, and the adjoint solve (same matrix: it is symmetric) replays them on its right-hand sides.
#include "isdf_ctx.hpp"
#include "minco_pcr.hpp"
#include "minco_dev.hpp"

namespace isdf {

using mpcr::M2;

namespace {

__device__ __forceinline__ void mv(const M2 &m, const double x[2], double y[2]) { y[0] = m.a * x[0] + m.b * x[1]; y[1] = m.c * x[0] + m.d * x[1]; }

// waypoint k of trajectory b (0: head position, N: tail position)
__device__ __forceinline__ void waypoint(const CbDev &P, const double *x, const double *ends, int k, double p[3]) {
    if (k == 0) { p[0] = ends[0]; p[1] = ends[1]; p[2] = ends[2]; }
    else if (k == P.N) { p[0] = ends[9]; p[1] = ends[10]; p[2] = ends[11]; }
    else { const double *w = x + P.N + 3 * (k - 1); p[0] = w[0]; p[1] = w[1]; p[2] = w[2]; }
}

// fixed-shape sum over the workgroup's threads (wavefront sums, then the wavefronts in order); result in every thread
__device__ __forceinline__ double block_sum(double v, double *s_part) {
    const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_part[wave] = v;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < nw; w++) s += s_part[w];
    return s;
}

constexpr int SH_ROW = 18;      // LDS record of a row during a round: inv(D) (4), L (4), U (4), r (3 x 2)

}  // namespace

__global__ __launch_bounds__(CB_MAX_THREADS) void cb_pre_kernel(const CbDev P) {
    extern __shared__ double s_mem[];
    const int N = P.N, n = N - 1, k = threadIdx.x, b = blockIdx.x;
    const int nvar = N + 3 * n;
    double *s_h = s_mem;                        // [N]
    double *s_row = s_mem + N;                  // [N][SH_ROW] rows 1..N-1; later the junction states [(N + 1)][6]
    __shared__ double s_part[CB_MAX_THREADS / 64];
    const double *x = P.x + (size_t)b * nvar;
    const double *ends = P.ends + (size_t)b * 18;
    const size_t ostride = (size_t)1 + 19 * (size_t)N;

    double T = 1.0, h = 1.0;
    if (k < N) {
        T = mpcr::tau_to_T(x[k]);
        h = 1.0 / T;
        s_h[k] = h;
        P.T[(size_t)b * N + k] = T;
    }
    __syncthreads();

    // ---- junction k (between pieces k - 1 and k), 1 <= k <= N - 1
    const bool row = k >= 1 && k <= n;
    M2 L{0, 0, 0, 0}, D{1, 0, 0, 1}, U{0, 0, 0, 0};
    double r[3][2] = {{0, 0}, {0, 0}, {0, 0}};
    double pk[3] = {0, 0, 0}, pn[3] = {0, 0, 0};      // waypoints k and k + 1 (this thread's piece)
    if (k < N) { waypoint(P, x, ends, k, pk); waypoint(P, x, ends, k + 1, pn); }
    if (row) {
        const double hl = s_h[k - 1], hr = h;
        mpcr::junction_blocks(hl, hr, L, D, U);
        double pl[3], dpl[3], dpr[3];
        waypoint(P, x, ends, k - 1, pl);
        for (int d = 0; d < 3; d++) { dpl[d] = pk[d] - pl[d]; dpr[d] = pn[d] - pk[d]; }
        mpcr::junction_rhs(hl, hr, dpl, dpr, r);
        if (k == 1) {
            double va[3][2];
            for (int d = 0; d < 3; d++) { va[d][0] = ends[3 + d]; va[d][1] = ends[6 + d]; }
            mpcr::rhs_minus(L, va, r);
            L = {0, 0, 0, 0};
        }
        if (k == n) {
            double va[3][2];
            for (int d = 0; d < 3; d++) { va[d][0] = ends[9 + 3 + d]; va[d][1] = ends[9 + 6 + d]; }
            mpcr::rhs_minus(U, va, r);
            U = {0, 0, 0, 0};
        }
    }
    // ---- parallel cyclic reduction: round `rd` eliminates the neighbours at distance s
    double *fac = P.fac + ((size_t)b * N + k) * CB_FAC_STRIDE;
    int rd = 0;
    for (int s = 1; s < n; s <<= 1, rd++) {
        if (row) {
            const M2 i = mpcr::inv(D);
            double *w = s_row + (size_t)k * SH_ROW;
            w[0] = i.a; w[1] = i.b; w[2] = i.c; w[3] = i.d;
            w[4] = L.a; w[5] = L.b; w[6] = L.c; w[7] = L.d;
            w[8] = U.a; w[9] = U.b; w[10] = U.c; w[11] = U.d;
            for (int d = 0; d < 3; d++) { w[12 + 2 * d] = r[d][0]; w[13 + 2 * d] = r[d][1]; }
        }
        __syncthreads();
        if (row) {
            M2 A{0, 0, 0, 0}, B{0, 0, 0, 0}, Ln{0, 0, 0, 0}, Un{0, 0, 0, 0};
            if (k - s >= 1) {
                const double *w = s_row + (size_t)(k - s) * SH_ROW;
                const M2 il{w[0], w[1], w[2], w[3]}, Ll{w[4], w[5], w[6], w[7]}, Ul{w[8], w[9], w[10], w[11]};
                A = mpcr::mul(L, il);
                const M2 t = mpcr::mul(A, Ul), l = mpcr::mul(A, Ll);
                D.a -= t.a; D.b -= t.b; D.c -= t.c; D.d -= t.d;
                Ln = {-l.a, -l.b, -l.c, -l.d};
                for (int d = 0; d < 3; d++) { double y[2]; mv(A, w + 12 + 2 * d, y); r[d][0] -= y[0]; r[d][1] -= y[1]; }
            }
            if (k + s <= n) {
                const double *w = s_row + (size_t)(k + s) * SH_ROW;
                const M2 ih{w[0], w[1], w[2], w[3]}, Lh{w[4], w[5], w[6], w[7]}, Uh{w[8], w[9], w[10], w[11]};
                B = mpcr::mul(U, ih);
                const M2 t = mpcr::mul(B, Lh), u = mpcr::mul(B, Uh);
                D.a -= t.a; D.b -= t.b; D.c -= t.c; D.d -= t.d;
                Un = {-u.a, -u.b, -u.c, -u.d};
                for (int d = 0; d < 3; d++) { double y[2]; mv(B, w + 12 + 2 * d, y); r[d][0] -= y[0]; r[d][1] -= y[1]; }
            }
            L = Ln; U = Un;
            double *f = fac + 8 * rd;
            f[0] = A.a; f[1] = A.b; f[2] = A.c; f[3] = A.d; f[4] = B.a; f[5] = B.b; f[6] = B.c; f[7] = B.d;
        }
        __syncthreads();
    }
    // ---- the rows stand alone: u = D^-1 r; junction states of every waypoint into LDS and global memory
    double *s_u = s_row;                        // [(N + 1)][6]: (v, a) per axis
    double uk[3][2] = {{0, 0}, {0, 0}, {0, 0}};
    if (row) {
        const M2 i = mpcr::inv(D);
        double *f = fac + 8 * CB_MAX_ROUNDS;
        f[0] = i.a; f[1] = i.b; f[2] = i.c; f[3] = i.d;
        for (int d = 0; d < 3; d++) mv(i, r[d], uk[d]);
    } else if (k == 0) {
        for (int d = 0; d < 3; d++) { uk[d][0] = ends[3 + d]; uk[d][1] = ends[6 + d]; }
    }
    double *gu = P.u + (size_t)b * (N + 1) * 6;
    if (k < N) for (int d = 0; d < 3; d++) { s_u[6 * k + 2 * d] = uk[d][0]; s_u[6 * k + 2 * d + 1] = uk[d][1]; gu[6 * k + 2 * d] = uk[d][0]; gu[6 * k + 2 * d + 1] = uk[d][1]; }
    if (k == 0) for (int d = 0; d < 3; d++) {
        const double v = ends[9 + 3 + d], a = ends[9 + 6 + d];
        s_u[6 * N + 2 * d] = v; s_u[6 * N + 2 * d + 1] = a; gu[6 * N + 2 * d] = v; gu[6 * N + 2 * d + 1] = a;
    }
    __syncthreads();
    // ---- piece k: coefficients into the sweeps' input array, energy and its partials into the energy block
    double e = 0.0;
    if (k < N) {
        double c[3][6], gc[3][6], gT;
        for (int d = 0; d < 3; d++)
            mpcr::hermite(T, h, pk[d], uk[d][0], uk[d][1], pn[d], s_u[6 * (k + 1) + 2 * d], s_u[6 * (k + 1) + 2 * d + 1], c[d]);
        e = mpcr::piece_energy(T, c, gc, gT);
        double *C = P.coeffs + (size_t)b * 18 * N + 6 * k;
        double *E = P.epart + (size_t)b * ostride;
        E[1 + k] = gT;
        for (int d = 0; d < 3; d++)
            for (int q = 0; q < 6; q++) { C[(size_t)d * 6 * N + q] = c[d][q]; E[1 + N + (size_t)d * 6 * N + 6 * k + q] = gc[d][q]; }
    }
    e = block_sum(e, s_part);
    if (k == 0) P.epart[(size_t)b * ostride] = e;
}

__global__ __launch_bounds__(CB_MAX_THREADS) void cb_post_kernel(const CbDev P) {
    extern __shared__ double s_mem[];
    const int N = P.N, n = N - 1, k = threadIdx.x, b = blockIdx.x;
    const int nvar = N + 3 * n;
    double *s_a = s_mem;                         // [9 N]: the pieces' end-half adjoints, then the rounds' right-hand sides, then mu [(N + 1)][6]
    double *s_w = s_mem + (size_t)9 * N + 6;     // [3 N]
    __shared__ double s_part[CB_MAX_THREADS / 64];
    const double *x = P.x + (size_t)b * nvar;
    const double *ends = P.ends + (size_t)b * 18;
    const size_t ostride = (size_t)1 + 19 * (size_t)N;
    const bool piece = k < N, row = k >= 1 && k <= n;

    // ---- dCost/dc of piece k: energy partials + the sweeps' sums, in the reference's order (energy, swept volume, integral)
    double G[3][6], gTp = 0.0, T = 0.0, h = 1.0, ge[3][6];
    const double *fac = P.fac + ((size_t)b * N + k) * CB_FAC_STRIDE;
    double fa[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (row && n > 1) for (int q = 0; q < 8; q++) fa[q] = fac[q];             // round 0's multipliers: on their way early
    if (piece) {
        T = P.T[(size_t)b * N + k];
        h = 1.0 / T;
        const double *E = P.epart + (size_t)b * ostride;
        gTp = E[1 + k];
        for (int d = 0; d < 3; d++) for (int q = 0; q < 6; q++) G[d][q] = E[1 + N + (size_t)d * 6 * N + 6 * k + q];
        for (int o = 0; o < P.n_out; o++) {
            const double *S = P.sweep + ((size_t)o * gridDim.x + b) * ostride;
            gTp += S[1 + k];
            for (int d = 0; d < 3; d++) for (int q = 0; q < 6; q++) G[d][q] += S[1 + N + (size_t)d * 6 * N + 6 * k + q];
        }
        for (int d = 0; d < 3; d++) {
            mpcr::piece_adjoint(h, G[d], ge[d]);
            s_a[9 * k + 3 * d] = ge[d][3]; s_a[9 * k + 3 * d + 1] = ge[d][4]; s_a[9 * k + 3 * d + 2] = ge[d][5];
        }
    }
    __syncthreads();
    double r[3][2] = {{0, 0}, {0, 0}, {0, 0}}, baseP[3] = {0, 0, 0};
    if (row) for (int d = 0; d < 3; d++) {
        const double *l = s_a + 9 * (k - 1) + 3 * d;
        baseP[d] = l[0] + ge[d][0];
        r[d][0] = l[1] + ge[d][1];
        r[d][1] = l[2] + ge[d][2];
    }
    __syncthreads();
    // ---- mu = K^-1 g_u: the forward elimination's rounds replayed on this right-hand side
    int rd = 0;
    for (int s = 1; s < n; s <<= 1, rd++) {
        if (row) for (int d = 0; d < 3; d++) { s_a[6 * k + 2 * d] = r[d][0]; s_a[6 * k + 2 * d + 1] = r[d][1]; }
        double fn[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (row && (s << 1) < n) for (int q = 0; q < 8; q++) fn[q] = fac[8 * (rd + 1) + q];
        __syncthreads();
        if (row) {
            const M2 A{fa[0], fa[1], fa[2], fa[3]}, B{fa[4], fa[5], fa[6], fa[7]};
            if (k - s >= 1) for (int d = 0; d < 3; d++) { double y[2]; mv(A, s_a + 6 * (k - s) + 2 * d, y); r[d][0] -= y[0]; r[d][1] -= y[1]; }
            if (k + s <= n) for (int d = 0; d < 3; d++) { double y[2]; mv(B, s_a + 6 * (k + s) + 2 * d, y); r[d][0] -= y[0]; r[d][1] -= y[1]; }
        }
        for (int q = 0; q < 8; q++) fa[q] = fn[q];
        __syncthreads();
    }
    double *s_mu = s_a;                          // [(N + 1)][6], zero at the trajectory's ends
    if (row) {
        const double *f = fac + 8 * CB_MAX_ROUNDS;
        const M2 i{f[0], f[1], f[2], f[3]};
        for (int d = 0; d < 3; d++) { double y[2]; mv(i, r[d], y); s_mu[6 * k + 2 * d] = y[0]; s_mu[6 * k + 2 * d + 1] = y[1]; }
    }
    if (k == 0) for (int q = 0; q < 6; q++) { s_mu[q] = 0.0; s_mu[6 * N + q] = 0.0; }
    __syncthreads();
    // ---- piece k: its share of dCost/dT and of the waypoint gradients
    double *res = P.res + (size_t)b * P.res_stride;
    double gT = 0.0;
    if (piece) {
        double pk[3], pn[3];
        waypoint(P, x, ends, k, pk); waypoint(P, x, ends, k + 1, pn);
        const double *u = P.u + (size_t)b * (N + 1) * 6 + 6 * k;
        for (int d = 0; d < 3; d++) {
            const double *ml = s_mu + 6 * k + 2 * d, *mr = s_mu + 6 * (k + 1) + 2 * d;
            gT += mpcr::piece_grad_T(h, G[d], pn[d] - pk[d], u[2 * d], u[2 * d + 1], u[6 + 2 * d], u[6 + 2 * d + 1], ml, mr);
            s_w[3 * k + d] = mpcr::piece_w(h, ml, mr);
        }
        gT += gTp;
        res[1 + k] = mpcr::grad_T_to_tau(x[k], gT + P.rho);
    }
    __syncthreads();
    if (row) for (int d = 0; d < 3; d++) res[1 + N + 3 * (k - 1) + d] = baseP[d] + s_w[3 * (k - 1) + d] - s_w[3 * k + d];
    // ---- cost = energy + sweeps + rho * sum(T)  (:377-420)
    const double tsum = block_sum(piece ? T : 0.0, s_part);
    if (k == 0) {
        const double e = P.epart[(size_t)b * ostride];
        double cost = e, parts[4] = {e, 0.0, 0.0, P.rho * tsum};
        for (int o = 0; o < P.n_out; o++) {
            const double v = P.sweep[((size_t)o * gridDim.x + b) * ostride];
            cost += v;
            parts[P.n_out == 2 ? 1 + o : 2] = v;
        }
        cost += P.rho * tsum;
        res[0] = cost;
        for (int q = 0; q < 4; q++) res[1 + nvar + q] = parts[q];
    }
    if (P.flag) {
        // `res` is host-mapped: every thread's stores go out (system scope) before the flag the host spins on
        __threadfence_system();
        __syncthreads();
        if (k == 0) {
            const unsigned long long ovf = P.stats ? __hip_atomic_load(&P.stats[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            __hip_atomic_store(&P.flag[b], P.seq | (ovf ? HOST_FLAG_OVERFLOW : 0ull), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

static int cb_threads(int N) { return ((N + 63) / 64) * 64; }

void launch_cb_pre(const CbDev &P, hipStream_t st) {
    const size_t lds = ((size_t)P.N + (size_t)P.N * SH_ROW + 6) * sizeof(double);
    hipLaunchKernelGGL(cb_pre_kernel, dim3(P.nb), dim3(cb_threads(P.N)), lds, st, P);
}
void launch_cb_post(const CbDev &P, hipStream_t st) {
    const size_t lds = ((size_t)12 * P.N + 12) * sizeof(double);
    hipLaunchKernelGGL(cb_post_kernel, dim3(P.nb), dim3(cb_threads(P.N)), lds, st, P);
}

}  // namespace isdf
