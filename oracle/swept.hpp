// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked into the product path.
// CPU restatement of the swept-volume SDF query (argmin over trajectory time) of
// src/swept_volume/include/swept_volume/sw_manager.hpp:
//   update_traj        :287-296   (traj_duration refreshed only when the total is < 300 s)
//   state_on_traj      :332-341   (getStateOnTrajStamp, position + rotation)
//   sdf_at_time        :550-556   (getSDFAtTimeStamp)
//   grad_prel_at_time  :566-572   (getGradPrelAtTimeStamp)
//   sdf_dot_at_time    :593-662   (getSDF_DOTAtTimeStampOptimized)
//   choice_t_init      :367-445   (coarse 0.2 s scan -> intervals, fine 0.02 s scan -> one seed per interval;
//                                  quirks q1/q2/q3 of SURVEY.md §8(a) A5 reproduced: shared running minimum and
//                                  seed across intervals, trailing open interval dropped, repeated t += dt)
//   gradient_descent   :1000-1062 (sign descent with up to 9 step halvings; g recomputed, momentum unused)
//   sdf_of_swept_volume:710-747   (the 4-argument overload the hot loop binds to, set_ts == false)
#pragma once
#include "vecmath.hpp"
#include "flatness.hpp"
#include "trajectory.hpp"
#include "shapes.hpp"
#include <vector>

namespace orc {

struct SweptCtx {
    Traj traj;
    double traj_duration = 0.0;     // the reference leaves this uninitialised; 0 until a < 300 s trajectory arrives
    FlatParams flat;
    isdf_shape shape;
    const MeshSdf *mesh = nullptr;
    double safety_hor = 0.866;

    void update_traj(const Traj &t) {
        traj = t;
        const double td = total_duration(t);
        if (td < 3 * 1e2) traj_duration = td;
    }
    void state_on_traj(double t, V3 &xt, M3 &Rt) const {
        V3 vt, at, jt;
        eval_pvaj(traj, t, xt, vt, at, jt);
        V4 q;
        flat_forward(flat, vt, at, jt, q);
        Rt = quat_to_rot(q);
    }
    double sdf_at_time(const V3 &p, double t) const {
        V3 xt; M3 Rt;
        state_on_traj(t, xt, Rt);
        return shape_sdf(shape, mesh, mulT(Rt, p - xt));
    }
    V3 grad_prel_at_time(const V3 &p, double t) const {
        V3 xt; M3 Rt;
        state_on_traj(t, xt, Rt);
        return shape_grad(shape, mesh, mulT(Rt, p - xt));
    }
    double sdf_dot_at_time(const V3 &p, double t) const {
        V3 xt, vt, at, jt;
        eval_pvaj(traj, t, xt, vt, at, jt);
        V4 q; V3 omg;
        flat_forward(flat, vt, at, jt, q, omg);
        const M3 Rt = quat_to_rot(q);
        const V3 temp = mulT(Rt, p - xt);
        const V3 g = shape_grad(shape, mesh, temp);
        // w_hat * temp, with w_hat = [0 -wz wy; wz 0 -wx; -wy wx 0]  (matrix-vector product, row sums in k order)
        const V3 wx(0 * temp.x + (-omg.z) * temp.y + omg.y * temp.z,
                    omg.z * temp.x + 0 * temp.y + (-omg.x) * temp.z,
                    (-omg.y) * temp.x + omg.x * temp.y + 0 * temp.z);
        const V3 pv = -(mulT(Rt, vt) + wx);
        return dot(g, pv);
    }

    double choice_t_init(const V3 &p, double dt, std::vector<double> &rl, std::vector<double> &rr,
                         std::vector<double> &rts) const {
        const double rough_dt = 0.2;
        double mindis = 1e9, range_mindis = 1e9, dis = 1e9;
        double time_seed = 0.0, range_time_seed = 0.0;
        double tou_lb = 0, tou_ub = 0;
        const double inf = 2 * safety_hor + 0.1;
        rl.clear(); rr.clear(); rts.clear();
        bool in_range = false;
        for (double t = 0; t < traj_duration; t += rough_dt) {
            dis = sdf_at_time(p, t);
            if (dis < mindis) { time_seed = t; mindis = dis; }
            if (dis < inf) {
                if (!in_range) {
                    in_range = true;
                    range_mindis = dis;
                    tou_lb = std::max(0.0, t - rough_dt);
                    tou_ub = t;
                } else {
                    tou_ub = std::min(traj_duration, t + rough_dt);
                }
            } else if (in_range) {
                in_range = false;
                tou_ub = std::min(traj_duration, t + rough_dt);
                rl.push_back(tou_lb);
                rr.push_back(tou_ub);
            }
        }
        const size_t n = rl.size();
        for (size_t i = 0; i < n; i++) {
            tou_lb = rl[i]; tou_ub = rr[i];
            for (double t = tou_lb; t < tou_ub; t += dt) {
                dis = sdf_at_time(p, t);
                if (dis < range_mindis) { range_time_seed = t; range_mindis = dis; }
            }
            rts.push_back(range_time_seed);
        }
        return time_seed;
    }

    void gradient_descent(double t_min, double t_max, double x0, double &fx, double &x, const V3 &p,
                          int *iters_out = nullptr) const {
        const int max_iter = 300;
        const double alpha = 0.02, tol = 1e-5;
        double tau = alpha, g = 100.0;
        x = x0;
        double prev_x = 10000000.0;
        int iter = 0;
        bool stop = false;
        while (iter < max_iter && !stop && std::fabs(x - prev_x) > tol) {
            if (iter == 0) fx = sdf_at_time(p, x);
            g = sdf_dot_at_time(p, x);
            tau = alpha;
            prev_x = x;
            for (int div = 1; div < 10; div++) {
                iter = iter + 1;
                g = sdf_dot_at_time(p, x);
                double xc = x - tau * ((int)(g > 0) - (int)(g < 0));
                xc = std::max(std::min(xc, t_max), t_min);
                const double fc = sdf_at_time(p, xc);
                if ((fc - fx) < 0) { x = xc; fx = fc; break; }
                tau = 0.5 * tau;
                if (div == 9) stop = true;
            }
        }
        if (iters_out) *iters_out = iter;
    }

    // returns min sdf (10.0 when no interval qualifies: outputs untouched)
    double sdf_of_swept_volume(const V3 &p, double &time_seed_f, V3 &grad_prel, int *n_ranges = nullptr) const {
        double t_star = 0, sdf_star = 1e1, min_sdf_star = 1e1;
        const double dtime = 0.02;
        std::vector<double> rl, rr, rts;
        choice_t_init(p, dtime, rl, rr, rts);
        const int range_count = (int)rl.size();
        if (n_ranges) *n_ranges = range_count;
        for (int i = 0; i < range_count; i++) {
            const double tmin_ = std::max(0.0, rl[i]);
            const double tmax_ = std::min(rr[i], traj_duration);
            gradient_descent(tmin_, tmax_, rts[i], sdf_star, t_star, p);
            if (sdf_star < min_sdf_star) {
                min_sdf_star = sdf_star;
                time_seed_f = t_star;
                grad_prel = grad_prel_at_time(p, t_star);
            }
        }
        return min_sdf_star;
    }
};

} // namespace orc
