// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked into the product path.
// CPU restatement of the quadrotor differential-flatness map used on the hot path.
//   forward  (quat)        follows src/utils/include/utils/flatness.hpp:53-86
//   forward  (quat, omega) follows src/utils/include/utils/flatness.hpp:88-148
//   backward (adjoint)     follows src/utils/include/utils/flatness.hpp:230-406 (backwardthreadsafe)
// Yaw psi == 0 on this path.  Checked bit-for-bit against the reference header itself
// (compiled through a minimal Eigen stand-in) by tests/test_oracle_ref.py -> oracle/_ref.
#pragma once
#include "vecmath.hpp"

namespace orc {

struct FlatParams {
    double mass = 0.61, grav = 9.8, dh = 0.10, dv = 0.10, cp = 0.01, veps = 1e-4;
    double dh_over_m() const { return dh / mass; }
};

// Every intermediate of the forward pass, shared by forward and adjoint.
struct FlatState {
    double v[3], a[3];
    double cp_term, w_term, w[3];
    double zu[3], zu_sq[3], zu_sqr_norm, zu_norm, z[3];
    double tilt_den;
    // second half (needed for omega and for the adjoint)
    double ng_den, ng00, ng01, ng02, ng11, ng12, ng22;
    double zu01, zu12, zu02;
    double v_dot_a, dw_term, dw[3], dz_term[3], dz[3];
    double omg_den, omg_term;
};

inline void flat_state(const FlatParams &P, const V3 &vel, const V3 &acc, const V3 &jer, FlatState &s) {
    const double dhm = P.dh_over_m();
    for (int i = 0; i < 3; i++) { s.v[i] = vel[i]; s.a[i] = acc[i]; }
    s.cp_term = std::sqrt(s.v[0] * s.v[0] + s.v[1] * s.v[1] + s.v[2] * s.v[2] + P.veps);
    s.w_term = 1.0 + P.cp * s.cp_term;
    for (int i = 0; i < 3; i++) s.w[i] = s.w_term * s.v[i];
    s.zu[0] = s.a[0] + dhm * s.w[0];
    s.zu[1] = s.a[1] + dhm * s.w[1];
    s.zu[2] = s.a[2] + dhm * s.w[2] + P.grav;
    for (int i = 0; i < 3; i++) s.zu_sq[i] = s.zu[i] * s.zu[i];
    s.zu_sqr_norm = s.zu_sq[0] + s.zu_sq[1] + s.zu_sq[2];
    s.zu_norm = std::sqrt(s.zu_sqr_norm);
    for (int i = 0; i < 3; i++) s.z[i] = s.zu[i] / s.zu_norm;
    s.tilt_den = std::sqrt(2.0 * (1.0 + s.z[2]));
    s.ng_den = s.zu_sqr_norm * s.zu_norm;
    s.ng00 = (s.zu_sq[1] + s.zu_sq[2]) / s.ng_den;
    s.zu01 = s.zu[0] * s.zu[1];
    s.zu12 = s.zu[1] * s.zu[2];
    s.zu02 = s.zu[0] * s.zu[2];
    s.ng01 = -s.zu01 / s.ng_den;
    s.ng02 = -s.zu02 / s.ng_den;
    s.ng11 = (s.zu_sq[0] + s.zu_sq[2]) / s.ng_den;
    s.ng12 = -s.zu12 / s.ng_den;
    s.ng22 = (s.zu_sq[0] + s.zu_sq[1]) / s.ng_den;
    s.v_dot_a = s.v[0] * s.a[0] + s.v[1] * s.a[1] + s.v[2] * s.a[2];
    s.dw_term = P.cp * s.v_dot_a / s.cp_term;
    for (int i = 0; i < 3; i++) s.dw[i] = s.w_term * s.a[i] + s.dw_term * s.v[i];
    for (int i = 0; i < 3; i++) s.dz_term[i] = jer[i] + dhm * s.dw[i];
    s.dz[0] = s.ng00 * s.dz_term[0] + s.ng01 * s.dz_term[1] + s.ng02 * s.dz_term[2];
    s.dz[1] = s.ng01 * s.dz_term[0] + s.ng11 * s.dz_term[1] + s.ng12 * s.dz_term[2];
    s.dz[2] = s.ng02 * s.dz_term[0] + s.ng12 * s.dz_term[1] + s.ng22 * s.dz_term[2];
    s.omg_den = s.z[2] + 1.0;
    s.omg_term = s.dz[2] / s.omg_den;
}

inline V4 flat_quat(const FlatState &s) {
    V4 q;
    q.w = 0.5 * s.tilt_den;
    q.x = -s.z[1] / s.tilt_den;
    q.y = s.z[0] / s.tilt_den;
    q.z = 0.0;
    return q;
}
inline V3 flat_omg(const FlatState &s) {
    return {-s.dz[1] + s.z[1] * s.omg_term,
            s.dz[0] - s.z[0] * s.omg_term,
            (s.z[1] * s.dz[0] - s.z[0] * s.dz[1]) / s.omg_den};
}

// flatness.hpp:53-86
inline void flat_forward(const FlatParams &P, const V3 &vel, const V3 &acc, const V3 &jer, V4 &quat) {
    FlatState s; flat_state(P, vel, acc, jer, s); quat = flat_quat(s);
}
// flatness.hpp:88-148
inline void flat_forward(const FlatParams &P, const V3 &vel, const V3 &acc, const V3 &jer, V4 &quat, V3 &omg) {
    FlatState s; flat_state(P, vel, acc, jer, s); quat = flat_quat(s); omg = flat_omg(s);
}

// flatness.hpp:230-406.  Inputs are the partials w.r.t. (pos, vel, quat, omega); outputs are the total
// partials w.r.t. (pos, vel, acc, jerk).  NB: a partial w.r.t. acc is NOT an input (the caller's gradAcc
// is never propagated by the reference) and psi == 0 so the psi-related adjoints are dropped exactly as the
// reference computes-and-discards them.
inline void flat_backward(const FlatParams &P, const V3 &vel, const V3 &acc, const V3 &jer,
                          const V3 &pos_grad, const V3 &vel_grad, const V4 &quat_grad, const V3 &omg_grad,
                          V3 &pos_tot, V3 &vel_tot, V3 &acc_tot, V3 &jer_tot) {
    FlatState s; flat_state(P, vel, acc, jer, s);
    const double dhm = P.dh_over_m();
    const double *v = s.v, *a = s.a, *z = s.z, *dz = s.dz, *zu = s.zu, *dzt = s.dz_term;
    const double tilt_den = s.tilt_den, omg_den = s.omg_den, omg_term = s.omg_term;
    const double ng_den = s.ng_den, zu_norm = s.zu_norm, zu_sqr_norm = s.zu_sqr_norm;

    // adjoint of the tilt quaternion
    const double tilt0b = quat_grad.w, tilt1b = quat_grad.x, tilt2b = quat_grad.y;
    const double tilt_den_sqr = tilt_den * tilt_den;
    const double tilt_denb = (z[1] * tilt1b - z[0] * tilt2b) / tilt_den_sqr + 0.5 * tilt0b;
    // adjoint of omega
    const double omg_termb = -z[0] * omg_grad.y + z[1] * omg_grad.x;
    double tempb = omg_grad.z / omg_den;
    double z1b = dz[0] * tempb;
    const double dz0b = z[1] * tempb + omg_grad.y;
    double z0b = -(dz[1] * tempb);
    const double dz1b = -z[0] * tempb - omg_grad.x;
    const double omg_denb = -((z[1] * dz[0] - z[0] * dz[1]) * tempb / omg_den) -
                            dz[2] * omg_termb / (omg_den * omg_den);
    tempb = -(omg_term * omg_grad.y);
    z0b += tempb;
    tempb = -(omg_term * omg_grad.x);
    z0b += tilt2b / tilt_den;
    z1b += -tempb - tilt1b / tilt_den;
    const double dz2b = omg_termb / omg_den;
    const double z2b = omg_denb + tilt_denb / tilt_den;

    // adjoint of dz = NG * dz_term
    const double ng02b = dzt[0] * dz2b + dzt[2] * dz0b;
    const double dzt0b = s.ng02 * dz2b + s.ng01 * dz1b + s.ng00 * dz0b;
    const double ng12b = dzt[1] * dz2b + dzt[2] * dz1b;
    const double dzt1b = s.ng12 * dz2b + s.ng11 * dz1b + s.ng01 * dz0b;
    const double ng22b = dzt[2] * dz2b;
    const double dzt2b = s.ng22 * dz2b + s.ng12 * dz1b + s.ng02 * dz0b;
    const double ng01b = dzt[0] * dz1b + dzt[1] * dz0b;
    const double ng11b = dzt[1] * dz1b;
    const double ng00b = dzt[0] * dz0b;
    jer_tot.z = dzt2b; const double dw2b = dhm * dzt2b;
    jer_tot.y = dzt1b; const double dw1b = dhm * dzt1b;
    jer_tot.x = dzt0b; const double dw0b = dhm * dzt0b;
    tempb = P.cp * (v[2] * dw2b + v[1] * dw1b + v[0] * dw0b) / s.cp_term;
    acc_tot.z = s.w_term * dw2b + v[2] * tempb;
    acc_tot.y = s.w_term * dw1b + v[1] * tempb;
    acc_tot.x = s.w_term * dw0b + v[0] * tempb;
    vel_tot.z = s.dw_term * dw2b + a[2] * tempb;
    vel_tot.y = s.dw_term * dw1b + a[1] * tempb;
    vel_tot.x = s.dw_term * dw0b + a[0] * tempb;
    double cp_termb = -(s.v_dot_a * tempb / s.cp_term);
    // adjoint of the NG entries
    tempb = ng22b / ng_den;
    double zu_sqr0b = tempb;
    double zu_sqr1b = tempb;
    double ng_denb = -((s.zu_sq[0] + s.zu_sq[1]) * tempb / ng_den);
    const double zu12b = -(ng12b / ng_den);
    tempb = ng11b / ng_den;
    ng_denb += s.zu12 * ng12b / (ng_den * ng_den) - (s.zu_sq[0] + s.zu_sq[2]) * tempb / ng_den;
    zu_sqr0b += tempb;
    double zu_sqr2b = tempb;
    const double zu02b = -(ng02b / ng_den);
    const double zu01b = -(ng01b / ng_den);
    tempb = ng00b / ng_den;
    ng_denb += s.zu02 * ng02b / (ng_den * ng_den) + s.zu01 * ng01b / (ng_den * ng_den) -
               (s.zu_sq[1] + s.zu_sq[2]) * tempb / ng_den;
    const double zu_normb = zu_sqr_norm * ng_denb - (zu[2] * z2b + zu[1] * z1b + zu[0] * z0b) / zu_sqr_norm;
    const double zu_sqr_normb = zu_norm * ng_denb + zu_normb / (2.0 * zu_norm);
    tempb += zu_sqr_normb;
    zu_sqr1b += tempb;
    zu_sqr2b += tempb;
    const double zu2b = z2b / zu_norm + zu[0] * zu02b + zu[1] * zu12b + 2 * zu[2] * zu_sqr2b;
    const double w2b = dhm * zu2b;
    const double zu1b = z1b / zu_norm + zu[2] * zu12b + zu[0] * zu01b + 2 * zu[1] * zu_sqr1b;
    const double w1b = dhm * zu1b;
    zu_sqr0b += zu_sqr_normb;
    const double zu0b = z0b / zu_norm + zu[2] * zu02b + zu[1] * zu01b + 2 * zu[0] * zu_sqr0b;
    const double w0b = dhm * zu0b;
    const double w_termb = a[2] * dw2b + a[1] * dw1b + a[0] * dw0b + v[2] * w2b + v[1] * w1b + v[0] * w0b;
    acc_tot.z += zu2b;
    acc_tot.y += zu1b;
    acc_tot.x += zu0b;
    cp_termb += P.cp * w_termb;
    const double v_sqr_normb = cp_termb / (2.0 * s.cp_term);
    vel_tot.z += s.w_term * w2b + 2 * v[2] * v_sqr_normb + vel_grad.z;
    vel_tot.y += s.w_term * w1b + 2 * v[1] * v_sqr_normb + vel_grad.y;
    vel_tot.x += s.w_term * w0b + 2 * v[0] * v_sqr_normb + vel_grad.x;
    pos_tot = pos_grad;
}

} // namespace orc
