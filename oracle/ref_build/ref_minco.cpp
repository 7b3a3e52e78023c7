// Builds the reference's own MINCO_S3NU and Trajectory<5> (src/utils/include/utils/minco.hpp, trajectory.hpp,
// root_finder.hpp, included from /root/reference, not copied) behind a C ABI so that tests can pin the oracle's
// restatements (oracle/minco.hpp, oracle/trajectory.hpp) and the product's host MINCO (csrc/minco_host.hpp) against the
// real code.  Eigen is the stand-in of eigen_shim/.  TEST INFRASTRUCTURE ONLY.
#include <utils/minco.hpp>
extern "C" {
// head/tail: 3x3 column-major (pos | vel | acc); waypoints 3 x (N-1) column-major; T: N
void ref_minco(int N, const double *head9, const double *tail9, const double *waypoints, const double *T,
               double *coeffs, double *energy, double *gdC, double *gdT) {
    Eigen::Matrix3d h, t;
    for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) { h(r, c) = head9[3 * c + r]; t(r, c) = tail9[3 * c + r]; }
    Eigen::Matrix3Xd P(3, N - 1);
    for (int i = 0; i < N - 1; i++) for (int r = 0; r < 3; r++) P(r, i) = waypoints[3 * i + r];
    Eigen::VectorXd ts(N);
    for (int i = 0; i < N; i++) ts(i) = T[i];
    minco::MINCO_S3NU m;
    m.setConditions(h, t, N);
    m.setParameters(P, ts);
    const Eigen::MatrixX3d &b = m.getCoeffs();
    for (int c = 0; c < 3; c++) for (int r = 0; r < 6 * N; r++) coeffs[(size_t)c * 6 * N + r] = b(r, c);
    double e = 0.0;
    m.getEnergy(e);
    *energy = e;
    Eigen::MatrixX3d g; Eigen::VectorXd gt;
    m.getEnergyPartialGradByCoeffs(g);
    m.getEnergyPartialGradByTimes(gt);
    for (int c = 0; c < 3; c++) for (int r = 0; r < 6 * N; r++) gdC[(size_t)c * 6 * N + r] = g(r, c);
    for (int i = 0; i < N; i++) gdT[i] = gt(i);
}
void ref_minco_propagate(int N, const double *head9, const double *tail9, const double *waypoints, const double *T,
                         const double *gradC, const double *gradT_partial, double *gradP, double *gradT) {
    Eigen::Matrix3d h, t;
    for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) { h(r, c) = head9[3 * c + r]; t(r, c) = tail9[3 * c + r]; }
    Eigen::Matrix3Xd P(3, N - 1);
    for (int i = 0; i < N - 1; i++) for (int r = 0; r < 3; r++) P(r, i) = waypoints[3 * i + r];
    Eigen::VectorXd ts(N), gtp(N);
    for (int i = 0; i < N; i++) { ts(i) = T[i]; gtp(i) = gradT_partial[i]; }
    minco::MINCO_S3NU m;
    m.setConditions(h, t, N);
    m.setParameters(P, ts);
    Eigen::MatrixX3d gc(6 * N, 3);
    for (int c = 0; c < 3; c++) for (int r = 0; r < 6 * N; r++) gc(r, c) = gradC[(size_t)c * 6 * N + r];
    Eigen::Matrix3Xd gp; Eigen::VectorXd gt;
    m.propogateGrad(gc, gtp, gp, gt);
    for (int i = 0; i < N - 1; i++) for (int r = 0; r < 3; r++) gradP[3 * i + r] = gp(r, i);
    for (int i = 0; i < N; i++) gradT[i] = gt(i);
}
// Trajectory<5> built by MINCO: piece lookup + state at time t (locatePieceIdx :545-563, getPos_Vel_Acc_Jerk :105-149)
void ref_traj_eval(int N, const double *head9, const double *tail9, const double *waypoints, const double *T, double t,
                   double out12[12], double *total_duration) {
    Eigen::Matrix3d h, tl;
    for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) { h(r, c) = head9[3 * c + r]; tl(r, c) = tail9[3 * c + r]; }
    Eigen::Matrix3Xd P(3, N - 1);
    for (int i = 0; i < N - 1; i++) for (int r = 0; r < 3; r++) P(r, i) = waypoints[3 * i + r];
    Eigen::VectorXd ts(N);
    for (int i = 0; i < N; i++) ts(i) = T[i];
    minco::MINCO_S3NU m;
    m.setConditions(h, tl, N);
    m.setParameters(P, ts);
    Trajectory<5> traj;
    m.getTrajectory(traj);
    Eigen::Vector3d p, v, a, j;
    traj.getPos_Vel_Acc_Jerk(t, p, v, a, j);
    for (int r = 0; r < 3; r++) { out12[r] = p(r); out12[3 + r] = v(r); out12[6 + r] = a(r); out12[9 + r] = j(r); }
    *total_duration = traj.getTotalDuration();
}
}
