// MINCO (s = 3, non-uniform time) on the device, for BATCHES of trajectories: the two halves of
// TrajOptimizer::costFunctionLmbm (back_end_optimizer.hpp:358-430) either side of the sweeps -
//   minco_forward_kernel   x = [tau | waypoints] -> T (forwardT :228-241), minco.setParameters (minco.hpp:433-513: the banded
//                          system, LU without pivoting :93-131, solve :137-165) -> coefficients, in the layout the sweep reads;
//   minco_backward_kernel  energy and its partials (:530-582), + the sweep's [cost | gradT | gradC], propogateGrad (:584-654:
//                          adjoint solve :169-197), rho * sum(T), chain rule to (tau, xi) (backwardGradT :283-305)  -> cost, g.
// One WAVEFRONT per trajectory; the band matrix (13 diagonals x 6N) and the right-hand sides live in LDS.  A factorisation is
// a chain of 6N dependent pivots - a single trajectory is faster on a CPU core (csrc/minco_host.hpp, which the
// single-trajectory callback keeps using) - but a batch runs its 128 chains side by side under another round's sweep, and
// the optimizer's host threads are left with nothing but the L-BFGS update.
// Every element sees EXACTLY the floating-point operations of minco_host.hpp / the reference's loops (same elimination order,
// same skipped zeros, no FMA contraction): the batch's iterates are bitwise those of the single-trajectory driver
// (tests/test_gpu_callback.py::test_optimize_lbfgs_batch_equals_individual).
#include "isdf_internal.hpp"
#include "minco_dev.hpp"

#pragma clang fp contract(off)

namespace isdf {

namespace {
constexpr int BW = 6;                         // sub- and super-diagonals of MINCO_S3NU's system
constexpr int ND = 2 * BW + 1;

__device__ __forceinline__ void wsync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// band storage like BandLU: element (i, j) at d[(i - j + BW) * n + j]
__device__ __forceinline__ double &band(double *d, int n, int i, int j) { return d[(size_t)(i - j + BW) * n + j]; }

__device__ __forceinline__ double tau_to_T(double tau) { return tau > 0.0 ? ((0.5 * tau + 1.0) * tau + 1.0) : 1.0 / ((0.5 * tau - 1.0) * tau + 1.0); }
__device__ __forceinline__ double grad_T_to_tau(double tau, double gT) {
    if (tau > 0) return gT * (tau + 1.0);
    const double den = (0.5 * tau - 1.0) * tau + 1.0;
    return gT * (1.0 - tau) / (den * den);
}

// BandLU::factorize: column k scaled by the pivot (lanes = the <= 6 rows below it), rank-1 update of the trailing 6 x 6
// block (lanes = its elements); the chain over k is the algorithm's
__device__ void band_factorize(double *A, int n, int lane) {
    const int li = lane % BW, lj = lane / BW;
    for (int k = 0; k + 1 < n; k++) {
        const int i_end = min(k + BW, n - 1), j_end = min(k + BW, n - 1);
        const double piv = band(A, n, k, k);
        if (lane < BW) {
            const int i = k + 1 + lane;
            if (i <= i_end) { double &a = band(A, n, i, k); if (a != 0.0) a = a / piv; }
        }
        wsync();
        if (lane < BW * BW) {
            const int i = k + 1 + li, j = k + 1 + lj;
            if (i <= i_end && j <= j_end) {
                const double ukj = band(A, n, k, j);
                if (ukj != 0.0) {
                    const double lik = band(A, n, i, k);
                    if (lik != 0.0) { double &a = band(A, n, i, j); a = a - lik * ukj; }
                }
            }
        }
        wsync();
    }
}
// BandLU::solve, 3 right-hand sides (column-major, leading dimension n): lanes = (row offset, column)
__device__ void band_solve(const double *A, int n, double *x, int lane) {
    const int li = lane % BW, c = lane / BW;
    for (int j = 0; j < n; j++) {
        const int i = j + 1 + li;
        if (lane < BW * 3 && i <= min(j + BW, n - 1)) {
            const double l = A[(size_t)(i - j + BW) * n + j];
            if (l != 0.0) x[(size_t)c * n + i] = x[(size_t)c * n + i] - l * x[(size_t)c * n + j];
        }
        wsync();
    }
    for (int j = n - 1; j >= 0; j--) {
        const double piv = A[(size_t)BW * n + j];
        if (lane < 3) x[(size_t)lane * n + j] = x[(size_t)lane * n + j] / piv;
        wsync();
        const int i = max(0, j - BW) + li;
        if (lane < BW * 3 && i < j) {
            const double u = A[(size_t)(i - j + BW) * n + j];
            if (u != 0.0) x[(size_t)c * n + i] = x[(size_t)c * n + i] - u * x[(size_t)c * n + j];
        }
        wsync();
    }
}
// BandLU::solve_transposed
__device__ void band_solve_transposed(const double *A, int n, double *x, int lane) {
    const int li = lane % BW, c = lane / BW;
    for (int j = 0; j < n; j++) {
        const double piv = A[(size_t)BW * n + j];
        if (lane < 3) x[(size_t)lane * n + j] = x[(size_t)lane * n + j] / piv;
        wsync();
        const int i = j + 1 + li;
        if (lane < BW * 3 && i <= min(j + BW, n - 1)) {
            const double u = A[(size_t)(j - i + BW) * n + i];            // A(j, i)
            if (u != 0.0) x[(size_t)c * n + i] = x[(size_t)c * n + i] - u * x[(size_t)c * n + j];
        }
        wsync();
    }
    for (int j = n - 1; j >= 0; j--) {
        const int i = max(0, j - BW) + li;
        if (lane < BW * 3 && i < j) {
            const double l = A[(size_t)(j - i + BW) * n + i];            // A(j, i)
            if (l != 0.0) x[(size_t)c * n + i] = x[(size_t)c * n + i] - l * x[(size_t)c * n + j];
        }
        wsync();
    }
}
}  // namespace

// LDS: [band 13 x 6N | c 6N x 3 | t1..t5 5N]
__global__ __launch_bounds__(64) void minco_forward_kernel(const MincoDevParams P) {
    extern __shared__ double s_dyn[];
    const int lane = threadIdx.x, k = blockIdx.x, N = P.N, n = 6 * N, nv = N + 3 * (N - 1);
    double *A = s_dyn, *c = A + (size_t)ND * n, *t1 = c + (size_t)3 * n, *t2 = t1 + N, *t3 = t2 + N, *t4 = t3 + N, *t5 = t4 + N;
    const double *x = P.x + (size_t)k * nv;
    const double *head = P.cond + (size_t)P.ids[k] * 18, *tail = head + 9;
    for (int e = lane; e < ND * n; e += 64) A[e] = 0.0;
    for (int e = lane; e < 3 * n; e += 64) c[e] = 0.0;
    for (int i = lane; i < N; i += 64) {
        const double T = tau_to_T(x[i]);
        t1[i] = T; t2[i] = T * T; t3[i] = t2[i] * T; t4[i] = t2[i] * t2[i]; t5[i] = t4[i] * T;
        P.T[(size_t)k * N + i] = T;
    }
    wsync();
#define AA(i, j) band(A, n, (i), (j))
#define CC(r, d) c[(size_t)(d) * n + (r)]
    if (lane == 0) {
        AA(0, 0) = 1.0; AA(1, 1) = 1.0; AA(2, 2) = 2.0;
        for (int d = 0; d < 3; d++) { CC(0, d) = head[d]; CC(1, d) = head[3 + d]; CC(2, d) = head[6 + d]; }
        const int e = n, l = N - 1;
        AA(e - 3, e - 6) = 1.0; AA(e - 3, e - 5) = t1[l]; AA(e - 3, e - 4) = t2[l]; AA(e - 3, e - 3) = t3[l]; AA(e - 3, e - 2) = t4[l]; AA(e - 3, e - 1) = t5[l];
        AA(e - 2, e - 5) = 1.0; AA(e - 2, e - 4) = 2 * t1[l]; AA(e - 2, e - 3) = 3 * t2[l]; AA(e - 2, e - 2) = 4 * t3[l]; AA(e - 2, e - 1) = 5 * t4[l];
        AA(e - 1, e - 4) = 2; AA(e - 1, e - 3) = 6 * t1[l]; AA(e - 1, e - 2) = 12 * t2[l]; AA(e - 1, e - 1) = 20 * t3[l];
        for (int d = 0; d < 3; d++) { CC(e - 3, d) = tail[d]; CC(e - 2, d) = tail[3 + d]; CC(e - 1, d) = tail[6 + d]; }
    }
    for (int i = lane; i + 1 < N; i += 64) {
        const int r = 6 * i, q = 6 * i;
        const double T1 = t1[i], T2 = t2[i], T3 = t3[i], T4 = t4[i], T5 = t5[i];
        AA(r + 3, q + 3) = 6.0;  AA(r + 3, q + 4) = 24.0 * T1;  AA(r + 3, q + 5) = 60.0 * T2;  AA(r + 3, q + 9) = -6.0;
        AA(r + 4, q + 4) = 24.0; AA(r + 4, q + 5) = 120.0 * T1; AA(r + 4, q + 10) = -24.0;
        AA(r + 5, q) = 1.0; AA(r + 5, q + 1) = T1; AA(r + 5, q + 2) = T2; AA(r + 5, q + 3) = T3; AA(r + 5, q + 4) = T4; AA(r + 5, q + 5) = T5;
        AA(r + 6, q) = 1.0; AA(r + 6, q + 1) = T1; AA(r + 6, q + 2) = T2; AA(r + 6, q + 3) = T3; AA(r + 6, q + 4) = T4; AA(r + 6, q + 5) = T5;
        AA(r + 6, q + 6) = -1.0;
        AA(r + 7, q + 1) = 1.0; AA(r + 7, q + 2) = 2 * T1; AA(r + 7, q + 3) = 3 * T2; AA(r + 7, q + 4) = 4 * T3; AA(r + 7, q + 5) = 5 * T4;
        AA(r + 7, q + 7) = -1.0;
        AA(r + 8, q + 2) = 2.0; AA(r + 8, q + 3) = 6 * T1; AA(r + 8, q + 4) = 12 * T2; AA(r + 8, q + 5) = 20 * T3;
        AA(r + 8, q + 8) = -2.0;
        for (int d = 0; d < 3; d++) CC(r + 5, d) = x[N + 3 * i + d];
    }
#undef AA
#undef CC
    wsync();
    band_factorize(A, n, lane);
    band_solve(A, n, c, lane);
    // coefficients in the sweep's layout (6N x 3 column-major per trajectory), the factors for the adjoint solve
    double *co = P.coeffs + (size_t)k * 3 * n, *lu = P.lu + (size_t)k * ND * n;
    for (int e = lane; e < 3 * n; e += 64) co[e] = c[e];
    for (int e = lane; e < ND * n; e += 64) lu[e] = A[e];
}

// LDS: [band 13 x 6N | c 6N x 3 | adj 6N x 3 | t1..t5 5N | gdT N | gradT N | term N]
__global__ __launch_bounds__(64) void minco_backward_kernel(const MincoDevParams P) {
    extern __shared__ double s_dyn[];
    const int lane = threadIdx.x, k = blockIdx.x, N = P.N, n = 6 * N, nv = N + 3 * (N - 1);
    double *A = s_dyn, *c = A + (size_t)ND * n, *adj = c + (size_t)3 * n, *t1 = adj + (size_t)3 * n, *t2 = t1 + N, *t3 = t2 + N, *t4 = t3 + N,
           *t5 = t4 + N, *gdT = t5 + N, *gradT = gdT + N, *term = gradT + N;
    const double *x = P.x + (size_t)k * nv;
    const double *o = P.out + (size_t)k * (1 + 19 * (size_t)N);              // the sweep's [cost | gradT | gradC] of this trajectory
    const double *lu = P.lu + (size_t)k * ND * n, *co = P.coeffs + (size_t)k * 3 * n;
    for (int e = lane; e < ND * n; e += 64) A[e] = lu[e];
    for (int e = lane; e < 3 * n; e += 64) c[e] = co[e];
    for (int i = lane; i < N; i += 64) {
        const double T = P.T[(size_t)k * N + i];
        t1[i] = T; t2[i] = T * T; t3[i] = t2[i] * T; t4[i] = t2[i] * t2[i]; t5[i] = t4[i] * T;
    }
    wsync();
#define CC(r, d) c[(size_t)(d) * n + (r)]
#define RD(ra, rb) (CC(ra, 0) * CC(rb, 0) + CC(ra, 1) * CC(rb, 1) + CC(ra, 2) * CC(rb, 2))
    // energy and its partials (MincoS3::energy), + the sweep's sums: adj = dCost/dc, gdT = dCost/dT (partial)
    for (int i = lane; i < N; i += 64) {
        const int r = 6 * i;
        term[i] = 36.0 * RD(r + 3, r + 3) * t1[i] + 144.0 * RD(r + 4, r + 3) * t2[i] + 192.0 * RD(r + 4, r + 4) * t3[i] +
                  240.0 * RD(r + 5, r + 3) * t3[i] + 720.0 * RD(r + 5, r + 4) * t4[i] + 720.0 * RD(r + 5, r + 5) * t5[i];
        for (int d = 0; d < 3; d++) {
            const double g5 = 240.0 * CC(r + 3, d) * t3[i] + 720.0 * CC(r + 4, d) * t4[i] + 1440.0 * CC(r + 5, d) * t5[i];
            const double g4 = 144.0 * CC(r + 3, d) * t2[i] + 384.0 * CC(r + 4, d) * t3[i] + 720.0 * CC(r + 5, d) * t4[i];
            const double g3 = 72.0 * CC(r + 3, d) * t1[i] + 144.0 * CC(r + 4, d) * t2[i] + 240.0 * CC(r + 5, d) * t3[i];
            const double *og = o + 1 + N + (size_t)d * n + r;
            double *a = adj + (size_t)d * n + r;
            a[0] = 0.0 + og[0]; a[1] = 0.0 + og[1]; a[2] = 0.0 + og[2]; a[3] = g3 + og[3]; a[4] = g4 + og[4]; a[5] = g5 + og[5];
        }
        const double gt = 36.0 * RD(r + 3, r + 3) + 288.0 * RD(r + 4, r + 3) * t1[i] + 576.0 * RD(r + 4, r + 4) * t2[i] +
                          720.0 * RD(r + 5, r + 3) * t2[i] + 2880.0 * RD(r + 5, r + 4) * t3[i] + 3600.0 * RD(r + 5, r + 5) * t4[i];
        gdT[i] = gt + o[1 + i];
    }
    wsync();
    band_solve_transposed(A, n, adj, lane);
    // propogateGrad's duration gradient (one lane per piece, the reference's summation order inside)
    for (int i = lane; i < N; i += 64) {
        double s = 0.0;
        if (i + 1 < N) {
            const int r = 6 * i;
            for (int d = 0; d < 3; d++) {
                const double negvel = -(CC(r + 1, d) + 2.0 * t1[i] * CC(r + 2, d) + 3.0 * t2[i] * CC(r + 3, d) + 4.0 * t3[i] * CC(r + 4, d) + 5.0 * t4[i] * CC(r + 5, d));
                const double negacc = -(2.0 * CC(r + 2, d) + 6.0 * t1[i] * CC(r + 3, d) + 12.0 * t2[i] * CC(r + 4, d) + 20.0 * t3[i] * CC(r + 5, d));
                const double negjer = -(6.0 * CC(r + 3, d) + 24.0 * t1[i] * CC(r + 4, d) + 60.0 * t2[i] * CC(r + 5, d));
                const double negsna = -(24.0 * CC(r + 4, d) + 120.0 * t1[i] * CC(r + 5, d));
                const double negcra = -120.0 * CC(r + 5, d);
                const double *a = adj + (size_t)d * n + r + 3;
                s += negsna * a[0]; s += negcra * a[1]; s += negvel * a[2]; s += negvel * a[3]; s += negacc * a[4]; s += negjer * a[5];
            }
        } else {
            const int e = n, l = N - 1;
            for (int d = 0; d < 3; d++) {
                const double negvel = -(CC(e - 5, d) + 2.0 * t1[l] * CC(e - 4, d) + 3.0 * t2[l] * CC(e - 3, d) + 4.0 * t3[l] * CC(e - 2, d) + 5.0 * t4[l] * CC(e - 1, d));
                const double negacc = -(2.0 * CC(e - 4, d) + 6.0 * t1[l] * CC(e - 3, d) + 12.0 * t2[l] * CC(e - 2, d) + 20.0 * t3[l] * CC(e - 1, d));
                const double negjer = -(6.0 * CC(e - 3, d) + 24.0 * t1[l] * CC(e - 2, d) + 60.0 * t2[l] * CC(e - 1, d));
                const double *a = adj + (size_t)d * n + e - 3;
                s += negvel * a[0]; s += negacc * a[1]; s += negjer * a[2];
            }
        }
        gradT[i] = s + gdT[i];
    }
#undef RD
#undef CC
    wsync();
    double *res = P.res + (size_t)k * (1 + nv);
    // chain rule to (tau, xi) (backwardGradT / backwardGradP); cost = energy + sweeps + rho * sum(T), sums in piece order
    for (int i = lane; i < N; i += 64) res[1 + i] = grad_T_to_tau(x[i], gradT[i] + P.rho);
    for (int e = lane; e < 3 * (N - 1); e += 64) { const int i = e / 3, d = e - 3 * i; res[1 + N + e] = adj[(size_t)d * n + 6 * i + 5]; }
    if (lane == 0) {
        double energy = 0.0, tsum = 0.0;
        for (int i = 0; i < N; i++) energy += term[i];
        for (int i = 0; i < N; i++) tsum += t1[i];
        double cost = energy + o[0];
        cost += P.rho * tsum;
        res[0] = cost;
    }
}

size_t minco_forward_lds(int N) { return ((size_t)ND * 6 * N + (size_t)18 * N + (size_t)5 * N) * sizeof(double); }
size_t minco_backward_lds(int N) { return ((size_t)ND * 6 * N + (size_t)36 * N + (size_t)8 * N) * sizeof(double); }

void launch_minco_forward(const MincoDevParams &P, int n_active, hipStream_t stream) {
    hipLaunchKernelGGL(minco_forward_kernel, dim3(n_active), dim3(64), minco_forward_lds(P.N), stream, P);
}
void launch_minco_backward(const MincoDevParams &P, int n_active, hipStream_t stream) {
    hipLaunchKernelGGL(minco_backward_kernel, dim3(n_active), dim3(64), minco_backward_lds(P.N), stream, P);
}

}  // namespace isdf

// The two kernels by themselves, behind the C ABI (include/isdf_accel.h): what the batch optimizer runs around every sweep,
// callable on its own so that the device MINCO can be checked against the reference's minco.hpp piece by piece.
#include "isdf_ctx.hpp"
#include <vector>
extern "C" int isdf_minco_device(isdf_ctx *c, int n_traj, int N, const double *heads_pva, const double *tails_pva, double rho,
                                 const double *x, const double *sweep_out, double *T_out, double *coeffs_out, double *cost_g_out) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    if (n_traj < 1 || N < 1 || !heads_pva || !tails_pva || !x) return isdf_fail(c, ISDF_ERR_INVALID_ARG, "bad arguments");
    if (isdf::minco_backward_lds(N) > 64 * 1024) return isdf_fail(c, ISDF_ERR_UNSUPPORTED, "a trajectory's band system must fit in 64 KB of LDS (N <= 46)");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t nv = (size_t)N + 3 * (size_t)(N - 1), n6 = (size_t)6 * N, ostride = isdf_out_stride(N);
    const size_t x_all = (size_t)n_traj * nv, T_all = (size_t)n_traj * N, C_all = (size_t)n_traj * 3 * n6, lu_all = (size_t)n_traj * 13 * n6,
                 out_all = (size_t)n_traj * ostride, res_all = (size_t)n_traj * (1 + nv), cond_all = (size_t)n_traj * 18;
    double *d = nullptr; int *d_ids = nullptr;
    HIPCHK(c, hipMalloc(&d, (x_all + T_all + C_all + lu_all + out_all + res_all + cond_all) * sizeof(double)));
    if (hipMalloc(&d_ids, (size_t)n_traj * sizeof(int)) != hipSuccess) { (void)hipFree(d); return isdf_fail(c, ISDF_ERR_HIP, "hipMalloc"); }
    double *d_x = d, *d_T = d_x + x_all, *d_C = d_T + T_all, *d_lu = d_C + C_all, *d_out = d_lu + lu_all, *d_res = d_out + out_all, *d_cond = d_res + res_all;
    std::vector<double> cond(cond_all);
    std::vector<int> ids(n_traj);
    for (int t = 0; t < n_traj; t++) {
        for (int k = 0; k < 9; k++) { cond[(size_t)18 * t + k] = heads_pva[9 * t + k]; cond[(size_t)18 * t + 9 + k] = tails_pva[9 * t + k]; }
        ids[t] = t;
    }
    int rc = ISDF_OK;
    auto chk = [&](hipError_t e) { if (e != hipSuccess && rc == ISDF_OK) { c->err = hipGetErrorString(e); rc = ISDF_ERR_HIP; } };
    chk(hipMemcpy(d_x, x, x_all * sizeof(double), hipMemcpyHostToDevice));
    chk(hipMemcpy(d_cond, cond.data(), cond_all * sizeof(double), hipMemcpyHostToDevice));
    chk(hipMemcpy(d_ids, ids.data(), (size_t)n_traj * sizeof(int), hipMemcpyHostToDevice));
    if (sweep_out) chk(hipMemcpy(d_out, sweep_out, out_all * sizeof(double), hipMemcpyHostToDevice));
    else chk(hipMemset(d_out, 0, out_all * sizeof(double)));
    isdf::MincoDevParams P{};
    P.N = N; P.rho = rho; P.x = d_x; P.ids = d_ids; P.cond = d_cond; P.T = d_T; P.coeffs = d_C; P.lu = d_lu; P.out = d_out; P.res = d_res;
    if (rc == ISDF_OK) {
        isdf::launch_minco_forward(P, n_traj, c->stream);
        isdf::launch_minco_backward(P, n_traj, c->stream);
        chk(hipGetLastError());
        chk(hipStreamSynchronize(c->stream));
    }
    if (rc == ISDF_OK && T_out) chk(hipMemcpy(T_out, d_T, T_all * sizeof(double), hipMemcpyDeviceToHost));
    if (rc == ISDF_OK && coeffs_out) chk(hipMemcpy(coeffs_out, d_C, C_all * sizeof(double), hipMemcpyDeviceToHost));
    if (rc == ISDF_OK && cost_g_out) chk(hipMemcpy(cost_g_out, d_res, res_all * sizeof(double), hipMemcpyDeviceToHost));
    (void)hipFree(d); (void)hipFree(d_ids);
    return rc;
}
