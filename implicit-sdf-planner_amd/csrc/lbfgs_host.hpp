// Host-side L-BFGS driver behind the objective callback.
// Follows lbfgs::lbfgs_optimize of src/utils/include/utils/lbfgs.hpp:480-835 - the ZJU-FAST-Lab LBFGS-Lite fork the
// reference ships - with its line search (line_search_lewisoverton, :290-440):
//   * the fork accepts a step as soon as the Armijo condition holds (the weak-Wolfe test is commented out, :373-386);
//     `weak_wolfe = 1` restores the Lewis-Overton test of the unpatched algorithm;
//   * the fork's two "patches" (:788-819) - fall back to the normalised steepest-descent direction, scaled to the
//     previous direction's norm, when the two-loop direction is long (>= dir_norm_cap) or not a descent direction, or
//     when the cautious update rejects the pair - are kept behind `reference_patches`.
// The driver only sequences evaluations of the callback; all objective arithmetic stays in the callback.
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

namespace isdf_host {

struct LbfgsParams {
    int mem_size = 8;
    double g_epsilon = 1.0e-5;
    int past = 3;
    double delta = 1.0e-6;
    int max_iterations = 0;
    int max_linesearch = 64;
    double min_step = 1.0e-20, max_step = 1.0e+20;
    double f_dec_coeff = 1.0e-4, s_curv_coeff = 0.9;
    double cautious_factor = 1.0e-6, machine_prec = 1.0e-16;
    int weak_wolfe = 0;            // 0: the reference fork (Armijo only)   1: Lewis-Overton weak Wolfe
    int reference_patches = 1;     // the fork's steepest-descent fallbacks
    double dir_norm_cap = 0.04;    // lbfgs.hpp:788
};

enum {   // the reference's return codes (lbfgs.hpp:133-160)
    LBFGS_CONVERGENCE = 0, LBFGS_STOP = 1, LBFGS_CANCELED = 2,
    LBFGSERR_UNKNOWNERROR = -1024, LBFGSERR_INVALID_N, LBFGSERR_INVALID_MEMSIZE, LBFGSERR_INVALID_GEPSILON,
    LBFGSERR_INVALID_TESTPERIOD, LBFGSERR_INVALID_DELTA, LBFGSERR_INVALID_MINSTEP, LBFGSERR_INVALID_MAXSTEP,
    LBFGSERR_INVALID_FDECCOEFF, LBFGSERR_INVALID_SCURVCOEFF, LBFGSERR_INVALID_MACHINEPREC, LBFGSERR_INVALID_MAXLINESEARCH,
    LBFGSERR_INVALID_FUNCVAL, LBFGSERR_MINIMUMSTEP, LBFGSERR_MAXIMUMSTEP, LBFGSERR_MAXIMUMLINESEARCH,
    LBFGSERR_MAXIMUMITERATION, LBFGSERR_WIDTHTOOSMALL, LBFGSERR_INVALIDPARAMETERS, LBFGSERR_INCREASEGRADIENT
};

typedef double (*lbfgs_eval_fn)(void *instance, const double *x, double *g, const int n);   // == lmbm_evaluate_t
typedef int (*lbfgs_progress_fn)(void *instance, const double *x, const double *g, double fx, double step, int k, int ls);

struct LbfgsResult { double f = 0.0; int status = 0, iterations = 0, evaluations = 0; };

namespace detail {
inline double dot(const std::vector<double> &a, const std::vector<double> &b) {
    double s = 0.0;
    for (size_t i = 0; i < a.size(); i++) s += a[i] * b[i];
    return s;
}
inline double dot(const double *a, const double *b, int n) {
    double s = 0.0;
    for (int i = 0; i < n; i++) s += a[i] * b[i];
    return s;
}
inline double inf_norm(const std::vector<double> &a) {
    double m = 0.0;
    for (double v : a) m = std::max(m, std::fabs(v));
    return m;
}
}   // namespace detail

class Lbfgs {
public:
    LbfgsParams param;
    lbfgs_eval_fn evaluate = nullptr;
    lbfgs_progress_fn progress = nullptr;
    void *instance = nullptr;
    void *progress_instance = nullptr;      // what `progress` gets (null: `instance`, as in the reference's callback_data_t)
    int evaluations = 0;

    // x (n doubles) holds the initial guess on entry and the last accepted iterate on return
    LbfgsResult minimize(double *x_io, int n) {
        using namespace detail;
        LbfgsResult res;
        evaluations = 0;
        const int m = param.mem_size;
        if (n <= 0) return fail(res, LBFGSERR_INVALID_N);
        if (m <= 0) return fail(res, LBFGSERR_INVALID_MEMSIZE);
        if (param.g_epsilon < 0.0) return fail(res, LBFGSERR_INVALID_GEPSILON);
        if (param.past < 0) return fail(res, LBFGSERR_INVALID_TESTPERIOD);
        if (param.delta < 0.0) return fail(res, LBFGSERR_INVALID_DELTA);
        if (param.min_step < 0.0) return fail(res, LBFGSERR_INVALID_MINSTEP);
        if (param.max_step < param.min_step) return fail(res, LBFGSERR_INVALID_MAXSTEP);
        if (!(param.f_dec_coeff > 0.0 && param.f_dec_coeff < 1.0)) return fail(res, LBFGSERR_INVALID_FDECCOEFF);
        if (!(param.s_curv_coeff < 1.0 && param.s_curv_coeff > param.f_dec_coeff)) return fail(res, LBFGSERR_INVALID_SCURVCOEFF);
        if (!(param.machine_prec > 0.0)) return fail(res, LBFGSERR_INVALID_MACHINEPREC);
        if (param.max_linesearch <= 0) return fail(res, LBFGSERR_INVALID_MAXLINESEARCH);

        std::vector<double> x(x_io, x_io + n), xp(n), g(n), gp(n), d(n), g_again(n);
        std::vector<double> pf(std::max(1, param.past), 0.0), alpha(m, 0.0), ys_hist(m, 0.0);
        std::vector<std::vector<double>> S(m, std::vector<double>(n, 0.0)), Y(m, std::vector<double>(n, 0.0));
        double old_dnorm = 1.0;

        double fx = call(x, g);
        pf[0] = fx;
        for (int i = 0; i < n; i++) d[i] = -g[i];
        int ret;
        int k = 1;
        if (inf_norm(g) / std::max(1.0, inf_norm(x)) < param.g_epsilon) {
            ret = LBFGS_CONVERGENCE;
        } else {
            double step = 1.0 / std::sqrt(dot(d, d));
            int end = 0, bound = 0;
            for (;;) {
                xp = x; gp = g;
                const int ls = line_search(x, fx, g, step, d, xp, gp, param.min_step, param.max_step);
                if (ls < 0) { x = xp; g = gp; ret = ls; break; }
                if (progress && progress(progress_instance ? progress_instance : instance, x.data(), g.data(), fx, step, k, ls)) { ret = LBFGS_CANCELED; break; }
                if (inf_norm(g) / std::max(1.0, inf_norm(x)) < param.g_epsilon) { ret = LBFGS_CONVERGENCE; break; }
                if (0 < param.past) {
                    if (param.past <= k) {
                        const double rate = std::fabs(pf[k % param.past] - fx) / std::max(1.0, std::fabs(fx));
                        if (rate < param.delta) { ret = LBFGS_STOP; break; }
                    }
                    pf[k % param.past] = fx;
                }
                if (param.max_iterations != 0 && param.max_iterations <= k) { ret = LBFGSERR_MAXIMUMITERATION; break; }
                ++k;
                std::vector<double> &s = S[end], &y = Y[end];
                for (int i = 0; i < n; i++) { s[i] = x[i] - xp[i]; y[i] = g[i] - gp[i]; }
                const double ys = dot(y, s), yy = dot(y, y);
                ys_hist[end] = ys;
                for (int i = 0; i < n; i++) d[i] = -g[i];
                const double cau = dot(s, s) * std::sqrt(dot(gp, gp)) * param.cautious_factor;
                bool steepest = false;
                if (ys > cau) {
                    // two-loop recursion (Nocedal 1980) over the last `bound` pairs
                    ++bound;
                    bound = m < bound ? m : bound;
                    end = (end + 1) % m;
                    int j = end;
                    for (int i = 0; i < bound; ++i) {
                        j = (j + m - 1) % m;
                        alpha[j] = dot(S[j], d) / ys_hist[j];
                        for (int q = 0; q < n; q++) d[q] += (-alpha[j]) * Y[j][q];
                    }
                    const double scale = ys / yy;
                    for (int q = 0; q < n; q++) d[q] *= scale;
                    for (int i = 0; i < bound; ++i) {
                        const double beta = dot(Y[j], d) / ys_hist[j];
                        for (int q = 0; q < n; q++) d[q] += (alpha[j] - beta) * S[j][q];
                        j = (j + 1) % m;
                    }
                    if (param.reference_patches && (std::sqrt(dot(d, d)) >= param.dir_norm_cap || dot(d, g) >= 0)) steepest = true;
                } else if (param.reference_patches) {
                    steepest = true;
                    end = (end + 1) % m;
                }
                if (steepest) {     // the fork re-evaluates the callback here (:790, :803)
                    call(x, g_again);
                    g = g_again;
                    const double gn = std::sqrt(dot(g, g));
                    for (int q = 0; q < n; q++) d[q] = gn > 0.0 ? -g[q] / gn * old_dnorm : -g[q] * old_dnorm;
                }
                const double dn = std::sqrt(dot(d, d));
                if (std::isinf(dn) || std::isnan(dn)) { ret = LBFGSERR_INVALID_FUNCVAL; break; }
                old_dnorm = dn;
                step = 1.0;
            }
        }
        std::copy(x.begin(), x.end(), x_io);
        res.f = fx; res.status = ret; res.iterations = k; res.evaluations = evaluations;
        return res;
    }

private:
    LbfgsResult fail(LbfgsResult &r, int code) { r.status = code; return r; }
    double call(std::vector<double> &x, std::vector<double> &g) {
        ++evaluations;
        return evaluate(instance, x.data(), g.data(), (int)x.size());
    }
    // line_search_lewisoverton (lbfgs.hpp:290-440)
    int line_search(std::vector<double> &x, double &f, std::vector<double> &g, double &stp, const std::vector<double> &s,
                    const std::vector<double> &xp, const std::vector<double> &gp, double stpmin, double stpmax) {
        using namespace detail;
        const int n = (int)x.size();
        int count = 0;
        bool brackt = false, touched = false;
        double mu = 0.0, nu = stpmax;
        if (!(stp > 0.0)) return LBFGSERR_INVALIDPARAMETERS;
        const double dginit = dot(gp, s);
        if (0.0 < dginit) return LBFGSERR_INCREASEGRADIENT;
        const double finit = f, dgtest = param.f_dec_coeff * dginit, dstest = param.s_curv_coeff * dginit;
        for (;;) {
            for (int i = 0; i < n; i++) x[i] = xp[i] + stp * s[i];
            f = call(x, g);
            ++count;
            if (std::isinf(f) || std::isnan(f)) return LBFGSERR_INVALID_FUNCVAL;
            if (f > finit + stp * dgtest) { nu = stp; brackt = true; }
            else if (param.weak_wolfe && dot(g, s) < dstest) mu = stp;
            else return count;
            if (param.max_linesearch <= count) return LBFGSERR_MAXIMUMLINESEARCH;
            if (brackt && (nu - mu) < param.machine_prec * nu) return LBFGSERR_WIDTHTOOSMALL;
            stp = brackt ? 0.5 * (mu + nu) : stp * 2.0;
            if (stp < stpmin) return LBFGSERR_MINIMUMSTEP;
            if (stp > stpmax) {
                if (touched) return LBFGSERR_MAXIMUMSTEP;
                touched = true;
                stp = stpmax;
            }
        }
    }
};

} // namespace isdf_host
