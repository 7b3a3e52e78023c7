// Device half of the objective callback - the two kernel BODIES (minco_dev.hip wraps them in kernels).
//
// Device half of the objective callback: everything TrajOptimizer::costFunctionLmbm does around the two
// sweeps (back_end_optimizer.hpp:358-430) - tau -> T, MINCO setParameters (minco.hpp:397-513), the jerk energy and its
// partials (:584-654), propogateGrad (:515-582), rho * sum(T), the chain rule back to (tau, waypoints) - as two small
// kernels either side of the sweeps, so that a callback moves n doubles down and n + 5 up:
//
//   cb_pre_kernel    x -> T, coefficients (straight into the sweeps' input arrays), energy + its partials, the junction states
//                    and the elimination's multipliers for the way back
//   (the sweeps)     accumulate [cost | gradT | gradC] per trajectory, as ever
//   cb_post_kernel   energy partials + the sweeps' sums -> adjoint solve -> (cost, g), written where the host waits for them
//
// One workgroup per trajectory; one thread per (piece / junction, axis) - the three axes of a row are independent right-hand
// sides of the same 2 x 2-block system, so they sit in three wavefront groups that run side by side on different SIMDs (each
// repeats the small matrix part; trajectories of more than 320 pieces keep all three axes in one thread: the workgroup size).
// The linear algebra is csrc/minco_pcr.hpp: the trajectory in its junction states (a symmetric positive definite
// block-tridiagonal system) solved by parallel cyclic reduction - ceil(log2(N - 1)) rounds, against the 6N dependent pivots of
// the reference's band LU (which one wavefront walks in 140 us at N = 40: rounds 2-4, removed).  The adjoint system of
// propogateGrad is the same matrix (it is symmetric) with another right-hand side.
//
// Tried for the callback whose sweep is ONE fused launch (C2, 16 us), measured, dropped (profiles/r5_callback_*_dropped.txt):
// the three kernels on two streams chained by device flags (44.7 us per callback against 36.8 one after the other: queue
// switches cost more than the overlapped launch latencies bring); the bodies INSIDE the fused launch, workgroup 0 solving while
// the launch's other 2 700 wavefronts poll its ready flags (46.9 us: the polling starves the solve); the same with EVERY
// workgroup repeating the solve into an LDS table (39.0 us: no polling, but three workgroups per CU share its SIMDs for the
// solve, and the adjoint still sits behind the last sample).  The host's band LU costs 10 us at N = 40: that configuration keeps
// it (isdf_host.hip cb_device_minco).
#pragma once
#include "isdf_internal.hpp"
#include "dev_math.hpp"
#include "minco_pcr.hpp"
#include "minco_dev.hpp"

namespace isdf {

namespace cbd {

using mpcr::M2;

__device__ __forceinline__ void mv(const M2 &m, const double x[2], double y[2]) { y[0] = m.a * x[0] + m.b * x[1]; y[1] = m.c * x[0] + m.d * x[1]; }

// coordinate d of waypoint k of the trajectory (0: head position, N: tail position)
__device__ __forceinline__ double waypoint(int N, const double *x, const double *ends, int k, int d) {
    if (k == 0) return ends[d];
    if (k == N) return ends[9 + d];
    return x[N + 3 * (k - 1) + d];
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

// LDS of one body (doubles): [x | ends staged: 4 N + 15, SPLIT only][1 / T: N][rows: 18 N] (+ padding)
__host__ __device__ constexpr size_t cb_lds_doubles(int N, bool split) { return (split ? (size_t)4 * N + 16 : 0) + (size_t)19 * N + 8; }

// Parallel cyclic reduction of K x = r for this thread's axes: ceil(log2(n)) rounds in which row k eliminates its neighbours at
// distance s = 1, 2, 4 ...; on return the row stands alone (x = D^-1 r).  s_row: [N][SH_ROW].  Ends on a barrier.
template <int ND> __device__ __forceinline__ void pcr_rounds(const bool row, const bool lead, const int k, const int n, const int d0,
                                                             M2 &L, M2 &D, M2 &U, double (&r)[ND][2], double *s_row) {
    for (int s = 1; s < n; s <<= 1) {
        if (row) {
            double *w = s_row + (size_t)k * SH_ROW;
            if (lead) {
                const M2 i = mpcr::inv(D);
                w[0] = i.a; w[1] = i.b; w[2] = i.c; w[3] = i.d;
                w[4] = L.a; w[5] = L.b; w[6] = L.c; w[7] = L.d;
                w[8] = U.a; w[9] = U.b; w[10] = U.c; w[11] = U.d;
            }
            _Pragma("unroll") for (int l = 0, d = d0; l < ND; l++, d++) { w[12 + 2 * d] = r[l][0]; w[13 + 2 * d] = r[l][1]; }
        }
        __syncthreads();
        if (row) {
            M2 Ln{0, 0, 0, 0}, Un{0, 0, 0, 0};
            if (k - s >= 1) {
                const double *w = s_row + (size_t)(k - s) * SH_ROW;
                const M2 il{w[0], w[1], w[2], w[3]}, Ll{w[4], w[5], w[6], w[7]}, Ul{w[8], w[9], w[10], w[11]};
                const M2 A = mpcr::mul(L, il);
                const M2 t = mpcr::mul(A, Ul), ll = mpcr::mul(A, Ll);
                D.a -= t.a; D.b -= t.b; D.c -= t.c; D.d -= t.d;
                Ln = {-ll.a, -ll.b, -ll.c, -ll.d};
                _Pragma("unroll") for (int l = 0, d = d0; l < ND; l++, d++) { double y[2]; mv(A, w + 12 + 2 * d, y); r[l][0] -= y[0]; r[l][1] -= y[1]; }
            }
            if (k + s <= n) {
                const double *w = s_row + (size_t)(k + s) * SH_ROW;
                const M2 ih{w[0], w[1], w[2], w[3]}, Lh{w[4], w[5], w[6], w[7]}, Uh{w[8], w[9], w[10], w[11]};
                const M2 B = mpcr::mul(U, ih);
                const M2 t = mpcr::mul(B, Lh), u = mpcr::mul(B, Uh);
                D.a -= t.a; D.b -= t.b; D.c -= t.c; D.d -= t.d;
                Un = {-u.a, -u.b, -u.c, -u.d};
                _Pragma("unroll") for (int l = 0, d = d0; l < ND; l++, d++) { double y[2]; mv(B, w + 12 + 2 * d, y); r[l][0] -= y[0]; r[l][1] -= y[1]; }
            }
            L = Ln; U = Un;
        }
        __syncthreads();
    }
}

// the launch's inputs into LDS in ONE round of loads (SPLIT: trajectories of up to 320 pieces); returns where x / ends now live
template <bool SPLIT> __device__ __forceinline__ void cb_stage_inputs(const CbDev &P, const int b, double *&s_mem, const double *&x, const double *&ends) {
    const int nvar = P.N + 3 * (P.N - 1);
    x = P.x + (size_t)b * nvar;
    ends = P.ends + (size_t)b * 18;
    if constexpr (SPLIT) {
        for (int i = threadIdx.x; i < nvar + 18; i += blockDim.x) s_mem[i] = i < nvar ? x[i] : ends[i - nvar];
        x = s_mem; ends = s_mem + nvar;
        s_mem += (size_t)4 * P.N + 16;
        __syncthreads();
    }
}

// SPLIT: three wavefront groups, one per axis (threads [0, 3 NT), NT >= pieces, a multiple of 64); the workgroup's further threads
// only keep the barriers.  b: trajectory of the launch; s_mem: cb_lds_doubles(N, SPLIT) doubles of LDS, s_part: 16.
// s_tab (LDS, nullable): additionally [piece][19] = T, c(6 x 3); glob = false: LDS outputs only.
template <bool SPLIT> __device__ __forceinline__ void cb_pre_body(const CbDev &P, const int b, const int NT, double *s_mem, double *s_part, double *s_tab = nullptr, const bool glob = true) {
    const int N = P.N, n = N - 1;
    const int ax = (int)threadIdx.x / NT;
    const int k = ax < (SPLIT ? 3 : 1) ? (int)threadIdx.x % NT : N;
    constexpr int ND = SPLIT ? 1 : 3;           // axes of this thread: d0 .. d0 + ND - 1 (local arrays are indexed by l = d - d0)
    const int d0 = SPLIT ? ax : 0;
    const bool lead = ax == 0;                  // the thread of a row that writes what the axes share
    const double *x, *ends;
    cb_stage_inputs<SPLIT>(P, b, s_mem, x, ends);
    double *s_h = s_mem;                        // [N]
    double *s_row = s_mem + N;                  // [N][SH_ROW] rows 1..N-1; later the junction states [(N + 1)][6] and the pieces' dE/dT shares [N][3]
    const size_t ostride = (size_t)1 + 19 * (size_t)N;

    double T = 1.0, h = 1.0;
    if (k < N) {
        mpcr::tau_to_T_h(x[k], T, h);
        if (lead) { s_h[k] = h; if (glob) P.T[(size_t)b * N + k] = T; }
    }
    __syncthreads();

    // ---- junction k (between pieces k - 1 and k), 1 <= k <= N - 1
    const bool row = k >= 1 && k <= n;
    M2 L{0, 0, 0, 0}, D{1, 0, 0, 1}, U{0, 0, 0, 0};
    double r[ND][2] = {}, pk[ND] = {}, pn[ND] = {};      // waypoints k and k + 1 (this thread's piece)
    if (k < N) _Pragma("unroll") for (int l = 0, d = d0; l < ND; l++, d++) { pk[l] = waypoint(N, x, ends, k, d); pn[l] = waypoint(N, x, ends, k + 1, d); }
    if (row) {
        const double hl = s_h[k - 1], hr = h;
        mpcr::junction_blocks(hl, hr, L, D, U);
        _Pragma("unroll") for (int l = 0, d = d0; l < ND; l++, d++) {
            const double pl = waypoint(N, x, ends, k - 1, d);
            mpcr::junction_rhs1(hl, hr, pk[l] - pl, pn[l] - pk[l], r[l]);
            if (k == 1) { const double va[2] = {ends[3 + d], ends[6 + d]}; double y[2]; mv(L, va, y); r[l][0] -= y[0]; r[l][1] -= y[1]; }
            if (k == n) { const double va[2] = {ends[9 + 3 + d], ends[9 + 6 + d]}; double y[2]; mv(U, va, y); r[l][0] -= y[0]; r[l][1] -= y[1]; }
        }
        if (k == 1) L = {0, 0, 0, 0};
        if (k == n) U = {0, 0, 0, 0};
    }
    pcr_rounds<ND>(row, lead, k, n, d0, L, D, U, r, s_row);
    // ---- the rows stand alone: u = D^-1 r; junction states of every waypoint into LDS and global memory
    double *s_u = s_row;                        // [(N + 1)][6]: (v, a) per axis
    double *s_gt = s_row + (size_t)6 * (N + 1); // [N][3]
    double uk[ND][2] = {};
    if (row) {
        const M2 i = mpcr::inv(D);
        _Pragma("unroll") for (int l = 0, d = d0; l < ND; l++, d++) mv(i, r[l], uk[l]);
    } else if (k == 0) {
        _Pragma("unroll") for (int l = 0, d = d0; l < ND; l++, d++) { uk[l][0] = ends[3 + d]; uk[l][1] = ends[6 + d]; }
    }
    double *gu = P.u + (size_t)b * (N + 1) * 6;
    if (k < N) _Pragma("unroll") for (int l = 0, d = d0; l < ND; l++, d++) { s_u[6 * k + 2 * d] = uk[l][0]; s_u[6 * k + 2 * d + 1] = uk[l][1]; if (glob) { gu[6 * k + 2 * d] = uk[l][0]; gu[6 * k + 2 * d + 1] = uk[l][1]; } }
    if (k == 0) _Pragma("unroll") for (int l = 0, d = d0; l < ND; l++, d++) {
        const double v = ends[9 + 3 + d], a = ends[9 + 6 + d];
        s_u[6 * N + 2 * d] = v; s_u[6 * N + 2 * d + 1] = a; if (glob) { gu[6 * N + 2 * d] = v; gu[6 * N + 2 * d + 1] = a; }
    }
    __syncthreads();
    // ---- piece k: coefficients into the sweeps' input array / the fused launch's LDS table, energy and its partials into the
    // energy block
    double e = 0.0;
    if (k < N) {
        double *C = P.coeffs + (size_t)b * 18 * N + 6 * k;
        double *E = P.epart + (size_t)b * ostride;
        double *S = s_tab ? s_tab + (size_t)k * 19 : nullptr;
        if (S && lead) S[0] = T;
        _Pragma("unroll") for (int l = 0, d = d0; l < ND; l++, d++) {
            double c[6], gc[6], gT;
            mpcr::hermite(T, h, pk[l], uk[l][0], uk[l][1], pn[l], s_u[6 * (k + 1) + 2 * d], s_u[6 * (k + 1) + 2 * d + 1], c);
            if (S) for (int q = 0; q < 6; q++) S[1 + 6 * d + q] = c[q];
            if (glob) {
                e += mpcr::piece_energy1(T, c, gc, gT);
                s_gt[3 * k + d] = gT;
                for (int q = 0; q < 6; q++) { C[(size_t)d * 6 * N + q] = c[q]; E[1 + N + (size_t)d * 6 * N + 6 * k + q] = gc[q]; }
            }
        }
    }
    if (glob) {                                     // (workgroup-uniform)
        e = block_sum(e, s_part);                   // (its barriers also order s_gt)
        if (k < N && lead) P.epart[(size_t)b * ostride + 1 + k] = s_gt[3 * k] + s_gt[3 * k + 1] + s_gt[3 * k + 2];
        if (threadIdx.x == 0) P.epart[(size_t)b * ostride] = e;
    }
}

// The way back.  The junction matrix depends on the durations only and is symmetric: the adjoint system is the forward system
// with another right-hand side - its elimination is simply run again (a round is ~70 instructions; replaying multipliers stored by
// the forward pass was measured slower: one dependent L2 round trip per round).
// s_mem: cb_lds_doubles(N, SPLIT) doubles of LDS.  n_traj: trajectories of the launch (the sweeps' blocks are [sweep][trajectory]).
template <bool SPLIT> __device__ __forceinline__ void cb_post_body(const CbDev &P, const int b, const int n_traj, const int NT, double *s_mem, double *s_part) {
    const int N = P.N, n = N - 1;
    const int ax = (int)threadIdx.x / NT;
    const int k = ax < (SPLIT ? 3 : 1) ? (int)threadIdx.x % NT : N;
    constexpr int ND = SPLIT ? 1 : 3;
    const int d0 = SPLIT ? ax : 0;
    const bool lead = ax == 0;
    const int nvar = N + 3 * n;
    const size_t ostride = (size_t)1 + 19 * (size_t)N;
    const bool piece = k < N, row = k >= 1 && k <= n;
    // ---- everything this thread needs from global memory, requested in one go: dCost/dc of piece k = energy partials + the
    // sweeps' sums, in the reference's order (energy, swept volume, integral); the junction states of its two ends
    double G[ND][6], gTp = 0.0, uu[ND][4] = {};
    if (piece) {
        const double *E = P.epart + (size_t)b * ostride;
        const double *u = P.u + (size_t)b * (N + 1) * 6 + 6 * k;
        if (lead) gTp = E[1 + k];
        _Pragma("unroll") for (int l = 0, d = d0; l < ND; l++, d++) {
            for (int q = 0; q < 6; q++) G[l][q] = E[1 + N + (size_t)d * 6 * N + 6 * k + q];
            uu[l][0] = u[2 * d]; uu[l][1] = u[2 * d + 1]; uu[l][2] = u[6 + 2 * d]; uu[l][3] = u[6 + 2 * d + 1];
        }
        for (int o = 0; o < P.n_out; o++) {
            const double *S = P.sweep + ((size_t)o * n_traj + b) * ostride;
            if (lead) gTp += S[1 + k];
            _Pragma("unroll") for (int l = 0, d = d0; l < ND; l++, d++) for (int q = 0; q < 6; q++) G[l][q] += S[1 + N + (size_t)d * 6 * N + 6 * k + q];
        }
    }
    const double *x, *ends;
    cb_stage_inputs<SPLIT>(P, b, s_mem, x, ends);
    double *s_h = s_mem;                         // [N]
    double *s_row = s_mem + N;                   // [N][SH_ROW]; before the rounds the pieces' end-half adjoints [N][9]; after them mu [(N + 1)][6], w [N][3], dT shares [N][3]
    double *s_a = s_row;

    double T = 0.0, h = 1.0;
    if (piece) {
        mpcr::tau_to_T_h(x[k], T, h);            // the same bits as on the way forward
        if (lead) s_h[k] = h;
    }
    double ge[ND][6];
    if (piece) _Pragma("unroll") for (int l = 0, d = d0; l < ND; l++, d++) {
        mpcr::piece_adjoint(h, G[l], ge[l]);
        s_a[9 * k + 3 * d] = ge[l][3]; s_a[9 * k + 3 * d + 1] = ge[l][4]; s_a[9 * k + 3 * d + 2] = ge[l][5];
    }
    __syncthreads();
    M2 L{0, 0, 0, 0}, D{1, 0, 0, 1}, U{0, 0, 0, 0};
    double r[ND][2] = {}, baseP[ND] = {};
    if (row) {
        mpcr::junction_blocks(s_h[k - 1], h, L, D, U);
        if (k == 1) L = {0, 0, 0, 0};
        if (k == n) U = {0, 0, 0, 0};
        _Pragma("unroll") for (int l = 0, d = d0; l < ND; l++, d++) {
            const double *lf = s_a + 9 * (k - 1) + 3 * d;       // end half of the piece on the left
            baseP[l] = lf[0] + ge[l][0];
            r[l][0] = lf[1] + ge[l][1];
            r[l][1] = lf[2] + ge[l][2];
        }
    }
    __syncthreads();
    // ---- mu = K^-1 g_u
    pcr_rounds<ND>(row, lead, k, n, d0, L, D, U, r, s_row);
    double *s_mu = s_row;                        // [(N + 1)][6], zero at the trajectory's ends
    double *s_w = s_row + (size_t)6 * (N + 1);   // [N][3]
    double *s_gt = s_w + (size_t)3 * N;          // [N][3]
    if (row) {
        const M2 i = mpcr::inv(D);
        _Pragma("unroll") for (int l = 0, d = d0; l < ND; l++, d++) { double y[2]; mv(i, r[l], y); s_mu[6 * k + 2 * d] = y[0]; s_mu[6 * k + 2 * d + 1] = y[1]; }
    }
    if (k == 0) _Pragma("unroll") for (int l = 0, d = d0; l < ND; l++, d++) { s_mu[2 * d] = 0.0; s_mu[2 * d + 1] = 0.0; s_mu[6 * N + 2 * d] = 0.0; s_mu[6 * N + 2 * d + 1] = 0.0; }
    __syncthreads();
    // ---- piece k: its share of dCost/dT and of the waypoint gradients
    double *res = P.res + (size_t)b * P.res_stride;
    if (piece) {
        _Pragma("unroll") for (int l = 0, d = d0; l < ND; l++, d++) {
            const double *ml = s_mu + 6 * k + 2 * d, *mr = s_mu + 6 * (k + 1) + 2 * d;
            const double dp = waypoint(N, x, ends, k + 1, d) - waypoint(N, x, ends, k, d);
            s_gt[3 * k + d] = mpcr::piece_grad_T(h, G[l], dp, uu[l][0], uu[l][1], uu[l][2], uu[l][3], ml, mr);
            s_w[3 * k + d] = mpcr::piece_w(h, ml, mr);
        }
    }
    __syncthreads();
    if (piece && lead) res[1 + k] = mpcr::grad_T_to_tau(x[k], (s_gt[3 * k] + s_gt[3 * k + 1] + s_gt[3 * k + 2]) + gTp + P.rho);
    if (row) _Pragma("unroll") for (int l = 0, d = d0; l < ND; l++, d++) res[1 + N + 3 * (k - 1) + d] = baseP[l] + s_w[3 * (k - 1) + d] - s_w[3 * k + d];
    // ---- cost = energy + sweeps + rho * sum(T)  (:377-420)
    const double tsum = block_sum(piece && lead ? T : 0.0, s_part);
    if (threadIdx.x == 0) {
        const double e = P.epart[(size_t)b * ostride];
        double cost = e, parts[4] = {e, 0.0, 0.0, P.rho * tsum};
        for (int o = 0; o < P.n_out; o++) {
            const double v = P.sweep[((size_t)o * n_traj + b) * ostride];
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
        if (threadIdx.x == 0) {
            const unsigned long long ovf = P.stats ? __hip_atomic_load(&P.stats[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            __hip_atomic_store(&P.flag[b], P.seq | (ovf ? HOST_FLAG_OVERFLOW : 0ull), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

}  // namespace cbd
}  // namespace isdf
