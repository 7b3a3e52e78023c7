// Shared surroundings for the pieces of the reference that are EXTRACTED from /root/reference at build time (see
// Makefile): the Config fields they read, the Generalshape base (getTransform hands back the body offset the test
// passes in) and the analytic shape classes of Shape.hpp:824-2391 with their gradient macro (:32-88).
// TEST INFRASTRUCTURE ONLY.
#pragma once
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
    double safety_hor = 0, occupancy_resolution = 0;
    int kernel_size = 0;                                                       // front end (kernel convolution)
    double kernel_max_roll = 0, kernel_max_pitch = 0, kernel_ang_res = 1, front_end_safeh = 0;
    Eigen::RowVector3d trans;
    Eigen::Matrix3d rotate;
};
#include "../_ref/gen/shape_macro.inc"
namespace shape {
using Eigen::Vector3d; using Eigen::Vector2d; using Eigen::Matrix3d; using Eigen::RowVector3d;   // Shape.hpp gets these through `using namespace vis`
typedef std::function<double(const Eigen::Vector3d &)> SDFFunction;
const uint8_t or_mask[8] = {0x80, 0x40, 0x20, 0x10, 0x08, 0x04, 0x02, 0x01};     // Shape.hpp:92
class Generalshape {
public:
    Config config;
#include "../_ref/gen/shape_kernel_types.inc"     // Shape.hpp:103-256: Shapekernel, ByteShapeKernel
    // the members BasicShape's kernel generation reads (Shape.hpp:258-306)
    int xkernel_size = -1, ykernel_size = -1, kernelsize = -1;
    double kernelresu = -1, kernel_max_roll = 0, kernel_max_pitch = 0, kernel_ang_res = 1;
    bool initselfkerneldone = false;
    Shapekernel *shape_kernels = nullptr;
    ByteShapeKernel *byte_shape_kernels = nullptr;
    Generalshape(const Config &conf, bool /*analytic*/) : config(conf) {
        xkernel_size = floor(2 * conf.kernel_max_roll / conf.kernel_ang_res) + 1;          // Shape.hpp:297-305
        ykernel_size = floor(2 * conf.kernel_max_pitch / conf.kernel_ang_res) + 1;
        kernelsize = config.kernel_size;
        kernelresu = config.occupancy_resolution;
        kernel_max_roll = config.kernel_max_roll; kernel_max_pitch = config.kernel_max_pitch; kernel_ang_res = config.kernel_ang_res;
    }
    virtual ~Generalshape() { delete[] shape_kernels; delete[] byte_shape_kernels; }
    // the `if (enablekernel)` block of BasicShape::initShape (Shape.hpp:406-462), verbatim
    void initShapeKernels() {
        const bool enablekernel = true;
#include "../_ref/gen/shape_kernel_gen.inc"
    }
    void getTransform(Eigen::Matrix3d &R, Eigen::RowVector3d &t) const { R = config.rotate; t = config.trans; }
    virtual double getonlySDF(const Eigen::RowVector3d &pos_rel) = 0;
    virtual double getonlySDF(const Eigen::RowVector3d &pos_rel, const Eigen::Matrix3d &R_obj) = 0;
    virtual Eigen::Vector3d getonlyGrad1(const Eigen::RowVector3d &pos_rel) = 0;
    virtual double getSDFwithGrad1(const Eigen::RowVector3d &pos_rel, Eigen::Vector3d &grad) = 0;
};
#include "../_ref/gen/shape_classes.inc"
}  // namespace shape

namespace {
inline shape::Generalshape *make(const std::string &name, const Config &c) {
    using namespace shape;
#define MK(n) if (name == #n) return new n(c);
    MK(CSG) MK(Torus) MK(Torus_big) MK(Cappedtorus) MK(Trefoil) MK(Table) MK(CappedCone) MK(RoundedCone) MK(WireframeBox)
    MK(BendLinear) MK(BendLinear_big) MK(TwistBox) MK(BendBox) MK(SmoothDifference) MK(SmoothIntersection)
    MK(SmoothIntersection_big) MK(Box)
#undef MK
    return nullptr;
}
}
