// Builds the reference's own flatness map (src/utils/include/utils/flatness.hpp, included from
// /root/reference, not copied) behind a C ABI so tests can pin oracle/flatness.hpp against it bit-for-bit.
// TEST INFRASTRUCTURE ONLY.
#include <utils/flatness.hpp>
extern "C" {
void ref_flat_forward(const double fp[6], const double vel[3], const double acc[3], const double jer[3],
                      double quat[4], double omg[3], double quat_only[4]) {
    flatness::FlatnessMap m;
    m.reset(fp[0], fp[1], fp[2], fp[3], fp[4], fp[5]);
    Eigen::Vector3d v, a, j, w; Eigen::Vector4d q, q2;
    for (int i = 0; i < 3; i++) { v(i) = vel[i]; a(i) = acc[i]; j(i) = jer[i]; }
    m.optimizated_forward(v, a, j, q, w);
    m.optimizated_forward(v, a, j, q2);
    for (int i = 0; i < 4; i++) { quat[i] = q(i); quat_only[i] = q2(i); }
    for (int i = 0; i < 3; i++) omg[i] = w(i);
}
void ref_flat_backward(const double fp[6], const double vel[3], const double acc[3], const double jer[3],
                       const double pos_grad[3], const double vel_grad[3], const double quat_grad[4],
                       const double omg_grad[3], double out12[12]) {
    flatness::FlatnessMap m;
    m.reset(fp[0], fp[1], fp[2], fp[3], fp[4], fp[5]);
    Eigen::Vector3d v, a, j, pg, vg, og, o0, o1, o2, o3; Eigen::Vector4d qg;
    for (int i = 0; i < 3; i++) { v(i) = vel[i]; a(i) = acc[i]; j(i) = jer[i]; pg(i) = pos_grad[i]; vg(i) = vel_grad[i]; og(i) = omg_grad[i]; }
    for (int i = 0; i < 4; i++) qg(i) = quat_grad[i];
    m.backwardthreadsafe(v, a, j, pg, vg, qg, og, o0, o1, o2, o3);
    for (int i = 0; i < 3; i++) { out12[i] = o0(i); out12[3 + i] = o1(i); out12[6 + i] = o2(i); out12[9 + i] = o3(i); }
}
}
