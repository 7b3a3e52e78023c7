// Builds the reference's own L-BFGS driver (src/utils/include/utils/lbfgs.hpp, included from /root/reference, not
// copied) behind a C ABI so that tests can pin the oracle's restatement (oracle/lbfgs_ref.py) and the product driver
// (csrc/lbfgs_host.hpp) against the real code, iterate by iterate.  Eigen is the stand-in of eigen_shim/.
// TEST INFRASTRUCTURE ONLY.
#include <utils/lbfgs.hpp>
namespace {
typedef double (*eval_fn)(void *, const double *, double *, int);
struct Ctx { eval_fn f; void *inst; };
double tramp(void *instance, const Eigen::VectorXd &x, Eigen::VectorXd &g, double &p_cost) {
    Ctx *c = (Ctx *)instance;
    p_cost = 0.0;
    return c->f(c->inst, x.data(), g.data(), (int)x.size());
}
}
extern "C" int ref_lbfgs_optimize(eval_fn f, void *inst, double *x_inout, int n, int mem_size, double g_epsilon, int past,
                                  double delta, int max_iterations, double *f_out) {
    Ctx c{f, inst};
    lbfgs::lbfgs_parameter_t p;
    p.mem_size = mem_size; p.g_epsilon = g_epsilon; p.past = past; p.delta = delta; p.max_iterations = max_iterations;
    Eigen::VectorXd x(n);
    for (int i = 0; i < n; i++) x(i) = x_inout[i];
    double fx = 0.0;
    const int ret = lbfgs::lbfgs_optimize(x, fx, tramp, nullptr, nullptr, &c, p);
    for (int i = 0; i < n; i++) x_inout[i] = x(i);
    *f_out = fx;
    return ret;
}
