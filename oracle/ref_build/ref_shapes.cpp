// Builds the reference's own analytic robot shapes - the 16 registry classes + Box of src/utils/include/utils/Shape.hpp
// (:824-2390) with their DEFINE_USEFUL_FUNCTION gradient macro (:32-88) - behind a C ABI, so that tests can pin the
// oracle's restatement (oracle/shapes.hpp) against the real formulas.  Shape.hpp as a whole needs ROS, libigl and the
// yaml Config, so the Makefile EXTRACTS those two line ranges from /root/reference into oracle/_ref/gen/ at build time
// (generated, git-ignored, never committed) and this file supplies the minimal surroundings: a Config that carries the
// numbers Box reads, and a Generalshape base whose getTransform hands back the body offset the test passes in.
// Eigen is the stand-in of eigen_shim/.  TEST INFRASTRUCTURE ONLY.
#include "ref_shapes_common.hpp"
extern "C" {
// trans[3], rotate[9] row-major (the body offset of poly_params), box[3] (Box only)
void *ref_shape_create(const char *name, const double *trans, const double *rotate, const double *box) {
    Config c;
    c.trans = Eigen::RowVector3d(trans[0], trans[1], trans[2]);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) c.rotate(i, j) = rotate[3 * i + j];
    if (box) { c.box_x = box[0]; c.box_y = box[1]; c.box_z = box[2]; }
    return make(name, c);
}
void ref_shape_destroy(void *h) { delete (shape::Generalshape *)h; }
void ref_shape_eval(void *h, const double *pts, int n, double *sdf, double *grad) {
    shape::Generalshape *s = (shape::Generalshape *)h;
    for (int i = 0; i < n; i++) {
        Eigen::RowVector3d p(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
        sdf[i] = s->getonlySDF(p);
        if (grad) {
            Eigen::Vector3d g = s->getonlyGrad1(p);
            grad[3 * i] = g(0); grad[3 * i + 1] = g(1); grad[3 * i + 2] = g(2);
        }
    }
}
}
