// C entry point over the reference's LMBM shim (src/utils/include/lmbm/lmbm.cpp:27-129, lmbm::lmbm_optimize), exactly as
// TrajOptimizer::optimize_traj_lmbm calls it (src/planner_algorithm/src/back_end_optimizer.cpp:125-132): a default-constructed
// lmbm_parameter_t (bundle 2, corrections 7 / 15, <= 10 000 iterations, <= 20 000 evaluations, 300 s) unless overridden.
// TEST INFRASTRUCTURE ONLY (the optimizer is the CALLER of the drop-in boundary, not part of the product).
#include "lmbm.h"
extern "C" {
// max_iterations / max_evaluations <= 0 and timeout <= 0 keep the reference's defaults.  Returns lmbm_optimize's code
// (LMBM_CONVERGENCE = 0 ... ; negative = LMBMERR_*).
int ref_lmbm_optimize(int n, double *x, double *fx, lmbm::lmbm_evaluate_t evaluate, void *instance, lmbm::lmbm_progress_t progress,
                      int max_iterations, int max_evaluations, double timeout) {
    lmbm::lmbm_parameter_t param;
    if (max_iterations > 0) param.max_iterations = max_iterations;
    if (max_evaluations > 0) param.max_evaluations = max_evaluations;
    if (timeout > 0) param.timeout = (float)timeout;
    return lmbm::lmbm_optimize(n, x, fx, evaluate, instance, progress, &param);
}
}
