// Builds the reference's own analytic robot shapes - the 16 registry classes + Box of src/utils/include/utils/Shape.hpp
// (:824-2390) with their DEFINE_USEFUL_FUNCTION gradient macro (:32-88) - behind a C ABI, so that tests can pin the
// oracle's restatement (oracle/shapes.hpp) against the real formulas.  Shape.hpp as a whole needs ROS, libigl and the
// yaml Config, so the Makefile EXTRACTS those two line ranges from /root/reference into oracle/_ref/gen/ at build time
// (generated, git-ignored, never committed) and this file supplies the minimal surroundings: a Config that carries the
// numbers Box reads, and a Generalshape base whose getTransform hands back the body offset the test passes in.
// Eigen is the stand-in of eigen_shim/.  TEST INFRASTRUCTURE ONLY.
#include <Eigen/Eigen>
#include <algorithm>
#include <cmath>
#include <functional>
#include <iostream>
#include <memory>
#include <string>
#include <vector>
#define PI 3.14159265358979323846
using namespace std;
struct Config {
    double box_x = 0, box_y = 0, box_z = 0;
    Eigen::RowVector3d trans;
    Eigen::Matrix3d rotate;
};
#include "../_ref/gen/shape_macro.inc"
namespace shape {
using Eigen::Vector3d; using Eigen::Vector2d; using Eigen::Matrix3d; using Eigen::RowVector3d;   // Shape.hpp gets these through `using namespace vis`
typedef std::function<double(const Eigen::Vector3d &)> SDFFunction;
class Generalshape {
public:
    Config config;
    Generalshape(const Config &conf, bool /*analytic*/) : config(conf) {}
    virtual ~Generalshape() {}
    void getTransform(Eigen::Matrix3d &R, Eigen::RowVector3d &t) const { R = config.rotate; t = config.trans; }
    virtual double getonlySDF(const Eigen::RowVector3d &pos_rel) = 0;
    virtual double getonlySDF(const Eigen::RowVector3d &pos_rel, const Eigen::Matrix3d &R_obj) = 0;
    virtual Eigen::Vector3d getonlyGrad1(const Eigen::RowVector3d &pos_rel) = 0;
    virtual double getSDFwithGrad1(const Eigen::RowVector3d &pos_rel, Eigen::Vector3d &grad) = 0;
};
#include "../_ref/gen/shape_classes.inc"
}  // namespace shape

namespace {
shape::Generalshape *make(const std::string &name, const Config &c) {
    using namespace shape;
#define MK(n) if (name == #n) return new n(c);
    MK(CSG) MK(Torus) MK(Torus_big) MK(Cappedtorus) MK(Trefoil) MK(Table) MK(CappedCone) MK(RoundedCone) MK(WireframeBox)
    MK(BendLinear) MK(BendLinear_big) MK(TwistBox) MK(BendBox) MK(SmoothDifference) MK(SmoothIntersection)
    MK(SmoothIntersection_big) MK(Box)
#undef MK
    return nullptr;
}
}
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
