"""ORACLE - TEST INFRASTRUCTURE ONLY.  Restatement of the reference's L-BFGS driver
(src/utils/include/utils/lbfgs.hpp: line_search_lewisoverton :290-440, lbfgs_optimize :480-835 - the LBFGS-Lite fork with
the Armijo-only acceptance :373-386 and the steepest-descent "patches" :788-819), written as plain sequential Python so
that the product driver (csrc/lbfgs_host.hpp) can be compared iterate by iterate on small problems.
Pinned against the reference's own lbfgs.hpp (oracle/_ref/libref_lbfgs.so, tests/test_oracle_ref.py); real Eigen's vectorised dot/norm reductions
may order their sums differently, which only matters at rounding level."""
import math

CONVERGENCE, STOP, CANCELED = 0, 1, 2
(ERR_UNKNOWN, ERR_N, ERR_MEM, ERR_GEPS, ERR_PAST, ERR_DELTA, ERR_MINSTEP, ERR_MAXSTEP, ERR_FDEC, ERR_SCURV, ERR_MPREC,
 ERR_MAXLS_PARAM, ERR_FUNCVAL, ERR_MINIMUMSTEP, ERR_MAXIMUMSTEP, ERR_MAXIMUMLINESEARCH, ERR_MAXIMUMITERATION,
 ERR_WIDTHTOOSMALL, ERR_INVALIDPARAMETERS, ERR_INCREASEGRADIENT) = range(-1024, -1024 + 20)


def _dot(a, b):
    s = 0.0
    for u, v in zip(a, b):
        s += u * v
    return s


def _inf(a):
    m = 0.0
    for v in a:
        m = max(m, abs(v))
    return m


def lbfgs_optimize(fun, x0, mem_size=8, g_epsilon=1e-5, past=3, delta=1e-6, max_iterations=0, max_linesearch=64,
                   min_step=1e-20, max_step=1e20, f_dec_coeff=1e-4, s_curv_coeff=0.9, cautious_factor=1e-6,
                   machine_prec=1e-16, trace=None):
    """fun(x list) -> (f, g list).  Returns (x, f, status, k, evaluations).  trace: list collecting x after each iteration."""
    n, m = len(x0), mem_size
    evals = [0]

    def call(x):
        evals[0] += 1
        f, g = fun(list(x))
        return float(f), [float(v) for v in g]

    def line_search(fx, stp, s, xp, gp):       # :290-440 (Armijo only, :373-386)
        count = 0
        brackt = touched = False
        mu, nu = 0.0, max_step
        if not stp > 0.0:
            return ERR_INVALIDPARAMETERS, None
        dginit = _dot(gp, s)
        if 0.0 < dginit:
            return ERR_INCREASEGRADIENT, None
        finit = fx
        dgtest = f_dec_coeff * dginit
        while True:
            x = [a + stp * b for a, b in zip(xp, s)]
            f, g = call(x)
            count += 1
            if math.isinf(f) or math.isnan(f):
                return ERR_FUNCVAL, None
            if f > finit + stp * dgtest:
                nu = stp
                brackt = True
            else:
                return count, (x, f, g, stp)
            if max_linesearch <= count:
                return ERR_MAXIMUMLINESEARCH, None
            if brackt and (nu - mu) < machine_prec * nu:
                return ERR_WIDTHTOOSMALL, None
            stp = 0.5 * (mu + nu) if brackt else stp * 2.0
            if stp < min_step:
                return ERR_MINIMUMSTEP, None
            if stp > max_step:
                if touched:
                    return ERR_MAXIMUMSTEP, None
                touched = True
                stp = max_step

    x = list(map(float, x0))
    pf = [0.0] * max(1, past)
    alpha = [0.0] * m
    lm_ys = [0.0] * m
    S = [[0.0] * n for _ in range(m)]
    Y = [[0.0] * n for _ in range(m)]
    olddnorm = 1.0
    fx, g = call(x)
    pf[0] = fx
    d = [-v for v in g]
    k = 1
    if _inf(g) / max(1.0, _inf(x)) < g_epsilon:
        return x, fx, CONVERGENCE, k, evals[0]
    step = 1.0 / math.sqrt(_dot(d, d))
    end = bound = 0
    while True:
        xp, gp = list(x), list(g)
        ls, out = line_search(fx, step, d, xp, gp)
        if ls < 0:
            x, g, ret = xp, gp, ls
            break
        x, fx, g, step = out
        if trace is not None:
            trace.append(list(x))
        if _inf(g) / max(1.0, _inf(x)) < g_epsilon:
            ret = CONVERGENCE
            break
        if 0 < past:
            if past <= k:
                rate = abs(pf[k % past] - fx) / max(1.0, abs(fx))
                if rate < delta:
                    ret = STOP
                    break
            pf[k % past] = fx
        if max_iterations != 0 and max_iterations <= k:
            ret = ERR_MAXIMUMITERATION
            break
        k += 1
        S[end] = [a - b for a, b in zip(x, xp)]
        Y[end] = [a - b for a, b in zip(g, gp)]
        ys = _dot(Y[end], S[end])
        yy = _dot(Y[end], Y[end])
        lm_ys[end] = ys
        d = [-v for v in g]
        cau = _dot(S[end], S[end]) * math.sqrt(_dot(gp, gp)) * cautious_factor
        steepest = False
        if ys > cau:
            bound = min(m, bound + 1)
            end = (end + 1) % m
            j = end
            for _ in range(bound):
                j = (j + m - 1) % m
                alpha[j] = _dot(S[j], d) / lm_ys[j]
                d = [a + (-alpha[j]) * b for a, b in zip(d, Y[j])]
            sc = ys / yy
            d = [a * sc for a in d]
            for _ in range(bound):
                beta = _dot(Y[j], d) / lm_ys[j]
                d = [a + (alpha[j] - beta) * b for a, b in zip(d, S[j])]
                j = (j + 1) % m
            if math.sqrt(_dot(d, d)) >= 0.04 or _dot(d, g) >= 0:      # :788
                steepest = True
        else:                                                            # :799-812
            steepest = True
            end = (end + 1) % m
        if steepest:
            _, g = call(x)
            gn = math.sqrt(_dot(g, g))
            d = [(-v / gn * olddnorm) if gn > 0.0 else (-v * olddnorm) for v in g]
        dn = math.sqrt(_dot(d, d))
        if math.isinf(dn) or math.isnan(dn):
            return x, fx, ERR_FUNCVAL, k, evals[0]
        olddnorm = dn
        step = 1.0
    return x, fx, ret, k, evals[0]
