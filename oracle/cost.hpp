// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked into the product path.
// CPU restatement of the two per-step sweeps of
// src/planner_algorithm/include/planner_algorithm/back_end_optimizer.hpp:
//   smoothed_l1            :332-356
//   quat_trans_d{w,x,y,z}  :127-173
//   grad_cost_p_sw         :827-866   (V1 per-point penalty, mu hard-coded 0.01)
//   grad_cost_p            :766-824   (V2 per-pose tile penalty; V3 = same with the ESDF qualifier)
//   sweep_swept (V1)       :557-649   addSaftyPenaOnSweptVolumeParallel
//   sweep_integral         :432-554   addTimeIntPenaltyParallel, with grad_cost_p plugged in before
//                                     backwardthreadsafe (:536) when enable_pos is set
// OpenMP mirrors the reference: parallel for, schedule(dynamic), one unnamed critical section for the merge
// (so the summation order — and the last bits — are nondeterministic, like the reference).
#pragma once
#include "vecmath.hpp"
#include "flatness.hpp"
#include "trajectory.hpp"
#include "shapes.hpp"
#include "grid.hpp"
#include "swept.hpp"
#include <omp.h>

namespace orc {

inline bool smoothed_l1(double x, double mu, double &f, double &df) {
    if (x < 0.0) return false;
    if (x > mu) { f = x - 0.5 * mu; df = 1.0; return true; }
    const double xdmu = x / mu;
    const double sqrxdmu = xdmu * xdmu;
    const double mumxd2 = mu - 0.5 * x;
    f = mumxd2 * sqrxdmu * xdmu;
    df = sqrxdmu * ((-0.5) * xdmu + 3.0 * mumxd2 / mu);
    return true;
}

// gradp_rel^T * D{W,X,Y,Z}(quat) * p_minus_x   (:859-862 / :816-819)
inline V4 quat_step_grad(const V4 &q, const V3 &g, const V3 &d) {
    const double w = q.w, x = q.x, y = q.y, z = q.z;
    M3 DW, DX, DY, DZ;
    const double dw[3][3] = {{0, 2 * z, -2 * y}, {-2 * z, 0, 2 * x}, {2 * y, -2 * x, 0}};
    const double dx[3][3] = {{0, 2 * y, 2 * z}, {2 * y, -4 * x, 2 * w}, {2 * z, -2 * w, -4 * x}};
    const double dy[3][3] = {{-4 * y, 2 * x, -2 * w}, {2 * x, 0, 2 * z}, {2 * w, 2 * z, -4 * y}};
    const double dz[3][3] = {{-4 * z, 2 * w, 2 * x}, {-2 * w, -4 * z, 2 * y}, {2 * x, 2 * y, 0}};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) { DW.m[i][j] = dw[i][j]; DX.m[i][j] = dx[i][j]; DY.m[i][j] = dy[i][j]; DZ.m[i][j] = dz[i][j]; }
    // Eigen evaluates (row * matrix) * vector left to right
    V4 r;
    r.w = dot(mulT(DW, g), d);
    r.x = dot(mulT(DX, g), d);
    r.y = dot(mulT(DY, g), d);
    r.z = dot(mulT(DZ, g), d);
    return r;
}

struct CostParams {
    FlatParams flat;
    isdf_shape shape;
    const MeshSdf *mesh = nullptr;
    const Grid *grid = nullptr;
    int variant = ISDF_V3_ESDF_TILE;
    int kernel_size = 13;
    int K = 64;
    int enable_dyn = 1, enable_pos = 1, enable_cull = 0;
    double safety_hor = 0.866, weight_p = 4000, weight_v = 1000, weight_omg = 1000, weight_theta = 1000;
    double vmax = 10, omgmax = 10, thetamax = 100, smoothing_eps = 1e-2, occ_thresh = 0.0;
    int threads = 1;
    int shard_rank = 0, shard_world = 1;   // isdf_set_shard mirror: contiguous range of pieces (integral) / points (V1)
    void shard_range(long long total, long long &b, long long &e) const {
        const long long q = total / shard_world, r = total % shard_world;
        b = shard_rank * q + (shard_rank < r ? shard_rank : r);
        e = b + q + (shard_rank < r ? 1 : 0);
    }
    double bd() const { return kernel_size * grid->res; }
    // whole-tile cull threshold of V3 (DESIGN.md "cull"): every voxel centre within reach of a non-zero penalty
    // lies closer than bound_radius + safety_hor to the pose; the trilinear sample can over-estimate the distance
    // to the nearest OCCUPIED voxel centre by at most sqrt(3)*res, and a voxel that qualifies (esdf <= occ_thresh) lies within
    // occ_thresh of an occupied one.
    double cull_threshold() const { return shape.bound_radius + safety_hor + std::sqrt(3.0) * grid->res + (occ_thresh > 0 ? occ_thresh : 0.0); }
};

struct SweepStats { long long units = 0, culled = 0, pairs = 0, grad_pairs = 0; };

// V2/V3 per-pose tile term.  Returns costp > 0.
inline bool grad_cost_p(const CostParams &P, const V3 &pos, const M3 &R, const V4 &quat, V3 &gradp, V4 &grad_quat,
                        double &costp, SweepStats *st) {
    costp = 0.0; gradp = V3(); grad_quat = V4();
    const Grid &G = *P.grid;
    const double bd = P.bd();
    if (P.variant == ISDF_V3_ESDF_TILE && P.enable_cull && P.shape.bound_radius > 0) {
        V3 gtmp;
        if (G.esdf_trilinear(pos, gtmp) > P.cull_threshold()) { if (st) st->culled++; return false; }
    }
    int lo[3], hi[3];
    G.aabb_index_box(pos, bd / 2, lo, hi);
    const double thresh = P.occ_thresh;
    for (int i = lo[0]; i <= hi[0]; i++)
        for (int j = lo[1]; j <= hi[1]; j++)
            for (int k = lo[2]; k <= hi[2]; k++) {
                const bool q = (P.variant == ISDF_V3_ESDF_TILE) ? G.index_qualifies_esdf(i, j, k, thresh)
                                                                : G.index_occupied(i, j, k);
                if (!q) continue;
                const V3 pos_eva = G.cube_center(i, j, k);
                const V3 pmx = pos_eva - pos;
                const V3 p_rel = mulT(R, pmx);
                if (std::fabs(p_rel.x) > bd / 2 || std::fabs(p_rel.y) > bd / 2 || std::fabs(p_rel.z) > bd / 2) continue;
                V3 gradp_rel;
                const double sdf_value = shape_sdf_grad(P.shape, P.mesh, p_rel, gradp_rel);
                if (st) st->pairs++;
                double sdf_cost = 0, grad_out = 0.0;
                smoothed_l1(P.safety_hor - sdf_value, P.smoothing_eps, sdf_cost, grad_out);
                if (sdf_cost > 0) {
                    if (st) st->grad_pairs++;
                    costp += sdf_cost;
                    gradp += grad_out * mul(R, gradp_rel);
                    const V4 sq = quat_step_grad(quat, gradp_rel, pmx);
                    grad_quat.w += -grad_out * sq.w;
                    grad_quat.x += -grad_out * sq.x;
                    grad_quat.y += -grad_out * sq.y;
                    grad_quat.z += -grad_out * sq.z;
                }
            }
    return costp > 0;
}

inline void betas(double s1, double b0[6], double b1[6], double b2[6], double b3[6], double b4[6]) {
    const double s2 = s1 * s1, s3 = s2 * s1, s4 = s2 * s2, s5 = s4 * s1;
    b0[0] = 1.0; b0[1] = s1; b0[2] = s2; b0[3] = s3; b0[4] = s4; b0[5] = s5;
    b1[0] = 0.0; b1[1] = 1.0; b1[2] = 2.0 * s1; b1[3] = 3.0 * s2; b1[4] = 4.0 * s3; b1[5] = 5.0 * s4;
    b2[0] = 0.0; b2[1] = 0.0; b2[2] = 2.0; b2[3] = 6.0 * s1; b2[4] = 12.0 * s2; b2[5] = 20.0 * s3;
    b3[0] = 0.0; b3[1] = 0.0; b3[2] = 0.0; b3[3] = 6.0; b3[4] = 24.0 * s1; b3[5] = 60.0 * s2;
    b4[0] = 0.0; b4[1] = 0.0; b4[2] = 0.0; b4[3] = 0.0; b4[4] = 24.0; b4[5] = 120.0 * s1;
}
// c^T * beta  (c = coeffs.block<6,3>(6i,0))
inline V3 ct_beta(const Traj &tr, int i, const double b[6]) {
    V3 r;
    for (int d = 0; d < 3; d++) {
        double s = 0.0;
        for (int k = 0; k < 6; k++) s += tr.coef(i, k, d) * b[k];
        r[d] = s;
    }
    return r;
}

// addTimeIntPenaltyParallel (+ grad_cost_p when enable_pos).  Accumulates into cost, gradT[N], gradC[6N x 3 colmajor].
inline void sweep_integral(const CostParams &P, const Traj &tr, double &cost, double *gradT, double *gradC,
                           SweepStats *stats) {
    const double velSqrMax = P.vmax * P.vmax, omgSqrMax = P.omgmax * P.omgmax, thetaMax = P.thetamax;
    const int N = tr.N, K = P.K;
    const double integralFrac = 1.0 / K;
    SweepStats tot;
    long long pb, pe;
    P.shard_range(N, pb, pe);
#pragma omp parallel for num_threads(P.threads) schedule(dynamic)
    for (int count = (int)pb * (K + 1); count < (int)pe * (K + 1); count++) {
        const int j = count % (K + 1), i = count / (K + 1);
        const double step = tr.T[i] * integralFrac;
        const double s1 = j * step;
        double b0[6], b1[6], b2[6], b3[6], b4[6];
        betas(s1, b0, b1, b2, b3, b4);
        const V3 pos = ct_beta(tr, i, b0), vel = ct_beta(tr, i, b1), acc = ct_beta(tr, i, b2),
                 jer = ct_beta(tr, i, b3), sna = ct_beta(tr, i, b4);
        V4 quat; V3 omg;
        flat_forward(P.flat, vel, acc, jer, quat, omg);
        const M3 rotate = quat_to_rot(quat);
        double pena = 0.0;
        V3 gradVel, gradPos, gradOmg; V4 gradQuat;
        SweepStats st;
        st.units = 1;
        if (P.enable_dyn) {
            const double cos_theta = 1.0 - 2.0 * (quat.x * quat.x + quat.y * quat.y);
            const double violaVel = sqnorm(vel) - velSqrMax;
            const double violaOmg = sqnorm(omg) - omgSqrMax;
            const double violaTheta = std::acos(cos_theta) - thetaMax;
            double pe, pd;
            if (smoothed_l1(violaVel, P.smoothing_eps, pe, pd)) {
                gradVel += (P.weight_v * pd * 2.0) * vel;
                pena += P.weight_v * pe;
            }
            if (smoothed_l1(violaOmg, P.smoothing_eps, pe, pd)) {
                gradOmg += (P.weight_omg * pd * 2.0) * omg;
                pena += P.weight_omg * pe;
            }
            if (smoothed_l1(violaTheta, P.smoothing_eps, pe, pd)) {
                const double f = P.weight_theta * pd / std::sqrt(1.0 - cos_theta * cos_theta) * 4.0;
                gradQuat.x += f * quat.x;
                gradQuat.y += f * quat.y;
                pena += P.weight_theta * pe;
            }
        }
        if (P.enable_pos) {
            V3 gp; V4 gq; double cp;
            if (grad_cost_p(P, pos, rotate, quat, gp, gq, cp, &st)) {
                gradPos += P.weight_p * gp;
                gradQuat.w += P.weight_p * gq.w; gradQuat.x += P.weight_p * gq.x;
                gradQuat.y += P.weight_p * gq.y; gradQuat.z += P.weight_p * gq.z;
                pena += P.weight_p * cp;
            }
        }
        V3 gP, gV, gA, gJ;
        flat_backward(P.flat, vel, acc, jer, gradPos, gradVel, gradQuat, gradOmg, gP, gV, gA, gJ);
        const double node = (j == 0 || j == K) ? 0.5 : 1.0;
        const double alpha = j * integralFrac;
#pragma omp critical
        {
            for (int r = 0; r < 6; r++)
                for (int d = 0; d < 3; d++)
                    gradC[d * 6 * N + 6 * i + r] += (b0[r] * gP[d] + b1[r] * gV[d] + b2[r] * gA[d] + b3[r] * gJ[d]) * node * step;
            gradT[i] += (dot(gP, vel) + dot(gV, acc) + dot(gA, jer) + dot(gJ, sna)) * alpha * node * step +
                        node * integralFrac * pena;
            cost += node * step * pena;
            tot.units += st.units; tot.culled += st.culled; tot.pairs += st.pairs; tot.grad_pairs += st.grad_pairs;
        }
    }
    if (stats) *stats = tot;
}

// addSaftyPenaOnSweptVolumeParallel.  tstar = lastTstar[M] (in/out).  Accumulates like the reference.
inline void sweep_swept(const CostParams &P, SweptCtx &sv, const Traj &tr, const double *points, int M,
                        double *tstar, double &cost, double *gradT, double *gradC, SweepStats *stats) {
    const int N = tr.N;
    sv.update_traj(tr);   // costFunctionLmbm :381-382
    SweepStats tot;
    long long kb, ke;
    P.shard_range(M, kb, ke);
#pragma omp parallel for num_threads(P.threads) schedule(dynamic)
    for (int k = (int)kb; k < (int)ke; k++) {
        V3 gradp_rel;   // reference leaves it uninitialised; it is only used when an interval was found
        const V3 pos_eva(points[3 * k], points[3 * k + 1], points[3 * k + 2]);
        int nr = 0;
        const double sdf_value = sv.sdf_of_swept_volume(pos_eva, tstar[k], gradp_rel, &nr);
        double time_local = tstar[k];
        const int i = locate_piece(tr, time_local);
        double b0[6], b1[6], b2[6], b3[6], b4[6];
        betas(time_local, b0, b1, b2, b3, b4);
        const V3 pos = ct_beta(tr, i, b0), vel = ct_beta(tr, i, b1), acc = ct_beta(tr, i, b2),
                 jer = ct_beta(tr, i, b3), sna = ct_beta(tr, i, b4);
        V4 quat;
        flat_forward(P.flat, vel, acc, jer, quat);
        const M3 rotate = quat_to_rot(quat);
        double pena = 0.0;
        V3 gradPos, gradVel, gradOmg; V4 gradQuat;
        // grad_cost_p_sw :827-866
        double sdf_cost = -1.0, sdf_out_grad = 0.0;
        smoothed_l1(P.safety_hor - sdf_value, 0.01, sdf_cost, sdf_out_grad);
        bool active = false;
        if (sdf_cost > 0) {
            const V3 sdf_grad = sdf_out_grad * mul(rotate, gradp_rel);
            const V3 pmx = pos_eva - pos;
            const V4 sq = quat_step_grad(quat, gradp_rel, pmx);
            if (sdf_cost > DBL_EPSILON) {
                active = true;
                gradPos += P.weight_p * sdf_grad;
                gradQuat.w += P.weight_p * (-sdf_out_grad * sq.w);
                gradQuat.x += P.weight_p * (-sdf_out_grad * sq.x);
                gradQuat.y += P.weight_p * (-sdf_out_grad * sq.y);
                gradQuat.z += P.weight_p * (-sdf_out_grad * sq.z);
                pena += P.weight_p * sdf_cost;
            }
        }
        V3 gP, gV, gA, gJ;
        flat_backward(P.flat, vel, acc, jer, gradPos, gradVel, gradQuat, gradOmg, gP, gV, gA, gJ);
        const double gdT = (-dot(gP, vel) + -dot(gV, acc) + -dot(gA, jer) + -dot(gJ, sna));
#pragma omp critical
        {
            cost += pena;
            for (int r = 0; r < 6; r++)
                for (int d = 0; d < 3; d++)
                    gradC[d * 6 * N + 6 * i + r] += (b0[r] * gP[d] + b1[r] * gV[d] + b2[r] * gA[d] + b3[r] * gJ[d]);
            for (int j = 0; j < i; j++) gradT[j] += gdT;
            tot.units += 1; tot.pairs += nr; tot.grad_pairs += active ? 1 : 0;
        }
    }
    if (stats) *stats = tot;
}

} // namespace orc
