// The reference-side binding of INTEGRATION.md, compiled: the two `static inline` sweeps of TrajOptimizer with the
// reference's EXACT signatures
//   addTimeIntPenaltyParallel          src/planner_algorithm/include/planner_algorithm/back_end_optimizer.hpp:432-438
//   addSaftyPenaOnSweptVolumeParallel  back_end_optimizer.hpp:557-562
// re-bodied as calls into the C ABI (include/isdf_accel.h), over Eigen types (here the Eigen stand-in the oracle's
// reference build uses, oracle/ref_build/eigen_shim: column-major storage, .data() like Eigen's).  What this proves rather
// than asserts: Eigen::MatrixX3d::data() of the 6N x 3 coefficient matrix IS the column-major layout isdf_eval reads and
// writes, cost / gradT / gradC are accumulated (+=) into the caller's objects, and lastTstar round-trips through the
// optimizer's std::vector<double>.  tests/test_gpu_binding.py drives it against the reference's own sweeps
// (oracle/_ref/libref_sweeps.so) on the same Eigen objects.  Also hosts the NATIVE timing loops of tools/boundary_bench.py
// (a reference caller is C++: Python's ctypes marshalling must not be in the measured call).
#include <Eigen/Eigen>
#include <chrono>
#include <vector>
#include <algorithm>
#include "isdf_accel.h"

class TrajOptimizer {
public:
    // what the reference keeps in TrajOptimizer around the two sweeps (:59-62), reduced to what the binding touches
    isdf_ctx *integral_ctx = nullptr;      // V2 / V3 context (or the dynamics-only sweep of the live configuration)
    isdf_ctx *swept_ctx = nullptr;         // V1 context
    std::vector<double> lastTstar;
    int last_status = 0;

    static inline void addTimeIntPenaltyParallel(void *ptr,
                                                 const Eigen::VectorXd &T,
                                                 const Eigen::MatrixX3d &coeffs,
                                                 double &cost,
                                                 double &pos_cost,
                                                 Eigen::VectorXd &gradT,
                                                 Eigen::MatrixX3d &gradC)
    {
        TrajOptimizer &obj = *(TrajOptimizer *)ptr;
        pos_cost = 0.0;                                        // the reference sets it to 0 and never adds to it (:445)
        const int N = (int)T.size();
        const double *Tp = T.data(), *Cp = coeffs.data();
        double *gT = gradT.data(), *gC = gradC.data();
        obj.last_status = isdf_eval(obj.integral_ctx, 1, &N, &Tp, &Cp, &cost, &gT, &gC, nullptr);
    }

    static inline void addSaftyPenaOnSweptVolumeParallel(void *ptr,
                                                         const Eigen::VectorXd &T,
                                                         const Eigen::MatrixX3d &coeffs,
                                                         double &cost,
                                                         Eigen::VectorXd &gradT,
                                                         Eigen::MatrixX3d &gradC)
    {
        TrajOptimizer &obj = *(TrajOptimizer *)ptr;
        const int N = (int)T.size();
        const double *Tp = T.data(), *Cp = coeffs.data();
        double *gT = gradT.data(), *gC = gradC.data();
        obj.last_status = isdf_eval(obj.swept_ctx, 1, &N, &Tp, &Cp, &cost, &gT, &gC, obj.lastTstar.empty() ? nullptr : obj.lastTstar.data());
    }
};

extern "C" {
// Builds the Eigen objects a TrajOptimizer would hold, pre-loads cost / gradT / gradC with the given start values (to show
// the += contract) and calls the stub.  coeffs_rm: 6N x 3 ROW-major on purpose - it is copied element by element into the
// Eigen::MatrixX3d, so the column-major layout the ABI sees is Eigen's own, not the test's.
// which: 0 = addTimeIntPenaltyParallel, 1 = addSaftyPenaOnSweptVolumeParallel.  Returns the isdf status.
int stub_call(void *ctx, int which, int N, const double *T, const double *coeffs_rm, double *cost_io, double *gradT_io, double *gradC_rm_io,
              double *tstar_io, int M) {
    TrajOptimizer opt;
    if (which == 0) opt.integral_ctx = (isdf_ctx *)ctx; else opt.swept_ctx = (isdf_ctx *)ctx;
    Eigen::VectorXd Tv(N), gT(N);
    Eigen::MatrixX3d C(6 * N, 3), gC(6 * N, 3);
    for (int i = 0; i < N; i++) { Tv(i) = T[i]; gT(i) = gradT_io[i]; }
    for (int r = 0; r < 6 * N; r++) for (int d = 0; d < 3; d++) { C(r, d) = coeffs_rm[3 * r + d]; gC(r, d) = gradC_rm_io[3 * r + d]; }
    if (which == 1 && tstar_io) opt.lastTstar.assign(tstar_io, tstar_io + M);
    double cost = *cost_io, pos_cost = -1.0;
    if (which == 0) TrajOptimizer::addTimeIntPenaltyParallel(&opt, Tv, C, cost, pos_cost, gT, gC);
    else TrajOptimizer::addSaftyPenaOnSweptVolumeParallel(&opt, Tv, C, cost, gT, gC);
    *cost_io = cost;
    for (int i = 0; i < N; i++) gradT_io[i] = gT(i);
    for (int r = 0; r < 6 * N; r++) for (int d = 0; d < 3; d++) gradC_rm_io[3 * r + d] = gC(r, d);
    if (which == 1 && tstar_io) for (int k = 0; k < M; k++) tstar_io[k] = opt.lastTstar[k];
    return opt.last_status;
}

// Native latency of the drop-in call: `iters` calls of addTimeIntPenaltyParallel (-> isdf_eval) on the same Eigen objects,
// per-call wall time in microseconds into lat_us (sorted ascending on return).
int stub_time_eval(void *ctx, int N, const double *T, const double *coeffs_cm, int iters, double *lat_us) {
    TrajOptimizer opt;
    opt.integral_ctx = (isdf_ctx *)ctx;
    Eigen::VectorXd Tv(N), gT(N);
    Eigen::MatrixX3d C(6 * N, 3), gC(6 * N, 3);
    for (int i = 0; i < N; i++) Tv(i) = T[i];
    for (int i = 0; i < 18 * N; i++) C.data()[i] = coeffs_cm[i];
    int rc = 0;
    for (int it = 0; it < iters; it++) {
        double cost = 0.0, pos_cost = 0.0;
        gT.setZero(); gC.setZero();
        const auto t0 = std::chrono::steady_clock::now();
        TrajOptimizer::addTimeIntPenaltyParallel(&opt, Tv, C, cost, pos_cost, gT, gC);
        const auto t1 = std::chrono::steady_clock::now();
        lat_us[it] = std::chrono::duration<double, std::micro>(t1 - t0).count();
        if (opt.last_status) rc = opt.last_status;
    }
    std::sort(lat_us, lat_us + iters);
    return rc;
}
// the same for the full objective callback through its lmbm_evaluate_t twin (what lmbm::lmbm_optimize would call)
int stub_time_callback(void *ctx, int n, const double *x, int iters, double *lat_us, double *cost_out) {
    std::vector<double> g(n);
    for (int it = 0; it < iters; it++) {
        const auto t0 = std::chrono::steady_clock::now();
        *cost_out = isdf_cost_function_lmbm(ctx, x, g.data(), n);
        const auto t1 = std::chrono::steady_clock::now();
        lat_us[it] = std::chrono::duration<double, std::micro>(t1 - t0).count();
    }
    std::sort(lat_us, lat_us + iters);
    return 0;
}
}
