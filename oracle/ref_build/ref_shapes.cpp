// Builds the reference's own analytic robot shapes - the 16 registry classes + Box of src/utils/include/utils/Shape.hpp
// (:824-2390) with their DEFINE_USEFUL_FUNCTION gradient macro (:32-88) - behind a C ABI, so that tests can pin the
// oracle's restatement (oracle/shapes.hpp) against the real formulas.  Shape.hpp as a whole needs ROS, libigl and the
// yaml Config, so the Makefile EXTRACTS those two line ranges from /root/reference into oracle/_ref/gen/ at build time
// (generated, git-ignored, never committed) and this file supplies the minimal surroundings: a Config that carries the
// numbers Box reads, and a Generalshape base whose getTransform hands back the body offset the test passes in.
// Eigen is the stand-in of eigen_shim/.  TEST INFRASTRUCTURE ONLY.
#include "ref_shapes_common.hpp"
#include <sstream>
// BasicShape's cached numeric SDF grid - num_sdf_map, filled by initShape (Shape.hpp:361-404) from the shape's OWN getSDFwithGrad1,
// sampled by getonlySDFNum / getonlyGrad1Num / getSDFwithGrad1Num (:481-600) - extracted the same way: the members (:267-288),
// the fill loop (:366-404), toAddr (:466) and the three samplers, inside a shell that forwards getSDFwithGrad1 to any of the
// reference's shape classes.  Pins the product's ISDF_SHAPE_GRID kind (and the oracle's restatement of it).
struct RefNumGrid {
    shape::Generalshape *src = nullptr;
    double getSDFwithGrad1(const Eigen::Vector3d &p, Eigen::Vector3d &g) { Eigen::RowVector3d r(p(0), p(1), p(2)); return src->getSDFwithGrad1(r, g); }
#include "../_ref/gen/num_grid_members.inc"
#include "../_ref/gen/num_grid_addr.inc"
    void initShape(const double ndx, const double ndy, const double ndz, const double nres) {
        const bool enableselfmap = true;
#include "../_ref/gen/num_grid_init.inc"
    }
#include "../_ref/gen/num_grid_samplers.inc"
};
extern "C" {
// the lattice of `shape_handle` (ref_shape_create) exactly as BasicShape::initShape<false, true>(ndx, ndy, ndz, nres) builds it
void *ref_numgrid_create(void *shape_handle, double ndx, double ndy, double ndz, double nres, int *dims_out) {
    RefNumGrid *g = new RefNumGrid();
    g->src = (shape::Generalshape *)shape_handle;
    std::ostringstream sink;
    std::streambuf *old = std::cout.rdbuf(sink.rdbuf());           // (initShape reports its progress on stdout)
    g->initShape(ndx, ndy, ndz, nres);
    std::cout.rdbuf(old);
    dims_out[0] = g->num_sdf_map_X_size; dims_out[1] = g->num_sdf_map_Y_size; dims_out[2] = g->num_sdf_map_Z_size;
    return g;
}
void ref_numgrid_destroy(void *h) { RefNumGrid *g = (RefNumGrid *)h; delete[] g->num_sdf_map; delete g; }
// cells: X*Y*Z x 4 (gradient xyz, distance), address toAddr(i, j, k)
void ref_numgrid_cells(void *h, double *cells, double *min_out) {
    RefNumGrid *g = (RefNumGrid *)h;
    const int n = g->num_sdf_map_X_size * g->num_sdf_map_Y_size * g->num_sdf_map_Z_size;
    for (int a = 0; a < n; a++) { for (int c = 0; c < 3; c++) cells[4 * a + c] = g->num_sdf_map[a].gradient(c); cells[4 * a + 3] = g->num_sdf_map[a].distance; }
    min_out[0] = g->x_min; min_out[1] = g->y_min; min_out[2] = g->z_min;
}
// getonlySDFNum, getonlyGrad1Num and getSDFwithGrad1Num at n points: sdf_a / grad_a from the separate functions, sdf_b / grad_b
// from the combined one (grad_b pre-filled with 7.0: it stays untouched outside the lattice)
void ref_numgrid_eval(void *h, const double *pts, int n, double *sdf_a, double *grad_a, double *sdf_b, double *grad_b) {
    RefNumGrid *g = (RefNumGrid *)h;
    for (int i = 0; i < n; i++) {
        Eigen::Vector3d p(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
        sdf_a[i] = g->getonlySDFNum(p);
        Eigen::Vector3d ga = g->getonlyGrad1Num(p), gb(7.0, 7.0, 7.0);
        sdf_b[i] = g->getSDFwithGrad1Num(p, gb);
        for (int c = 0; c < 3; c++) { grad_a[3 * i + c] = ga(c); grad_b[3 * i + c] = gb(c); }
    }
}
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
