// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked into the product path.
// CPU restatement of the piecewise-quintic container used by the swept-volume path.
//   locate_piece      follows src/utils/include/utils/trajectory.hpp:545-563 (Trajectory::locatePieceIdx)
//   eval_pvaj         follows src/utils/include/utils/trajectory.hpp:105-149 (Piece::getPos_Vel_Acc_Jerk)
//   total_duration    follows src/utils/include/utils/trajectory.hpp:457-466
// Coefficient storage: the optimizer's 6N x 3 column-major matrix (ascending powers, minco.hpp:402,545).
// Trajectory<5> keeps the same numbers reversed (highest power first, minco.hpp:521-525); eval_pvaj walks
// them in that reversed order so the floating-point sequence is the reference's.
#pragma once
#include "vecmath.hpp"
#include <vector>

namespace orc {

struct Traj {
    int N = 0;
    const double *T = nullptr;        // N durations
    const double *C = nullptr;        // 6N x 3 column-major, rows 6i..6i+5 = c0..c5 of piece i
    double coef(int piece, int power, int dim) const { return C[dim * 6 * N + 6 * piece + power]; }
};

inline double total_duration(const Traj &tr) {
    double s = 0.0;
    for (int i = 0; i < tr.N; i++) s += tr.T[i];
    return s;
}

// t is replaced by the piece-local time.  Junction times belong to the EARLIER piece (strict '>').
inline int locate_piece(const Traj &tr, double &t) {
    int idx = 0;
    double dur = 0.0;
    for (idx = 0; idx < tr.N && t > (dur = tr.T[idx]); idx++) t -= dur;
    if (idx == tr.N) {
        idx--;
        t += tr.T[idx];
    }
    return idx;
}

// Fused evaluation at piece-local time t; col(i) of the reference's 3x6 matrix is the coefficient of
// power (5 - i), so the loop "for i = D..0" visits powers 0,1,...,5.
inline void eval_pvaj_local(const Traj &tr, int piece, double t, V3 &pos, V3 &vel, V3 &acc, V3 &jerk) {
    pos = V3(); vel = V3(); acc = V3(); jerk = V3();
    double pos_tn = 1.0, vel_tn = 1.0, acc_tn = 1.0, jerk_tn = 1.0;
    int vel_n = 1, acc_m = 1, acc_n = 2, jerk_l = 1, jerk_m = 2, jerk_n = 3;
    const int D = 5;
    for (int i = D; i >= 0; i--) {
        const int power = D - i;
        const V3 c(tr.coef(piece, power, 0), tr.coef(piece, power, 1), tr.coef(piece, power, 2));
        pos += pos_tn * c;
        pos_tn *= t;
        if (i <= D - 1) {
            vel += (vel_n * vel_tn) * c;
            vel_tn *= t;
            vel_n++;
        }
        if (i <= D - 2) {
            acc += (acc_m * acc_n * acc_tn) * c;
            acc_tn *= t;
            acc_m++; acc_n++;
        }
        if (i <= D - 3) {
            jerk += (jerk_l * jerk_m * jerk_n * jerk_tn) * c;
            jerk_tn *= t;
            jerk_l++; jerk_m++; jerk_n++;
        }
    }
}

inline void eval_pvaj(const Traj &tr, double t, V3 &pos, V3 &vel, V3 &acc, V3 &jerk) {
    int piece = locate_piece(tr, t);
    eval_pvaj_local(tr, piece, t, pos, vel, acc, jerk);
}

} // namespace orc
