// Builds the reference's own hot loop behind a C ABI so that tests can pin the oracle against the REAL code:
//   SweptVolumeManager  updateTraj / getStateOnTrajStamp / posEva2Rel   sw_manager.hpp:287-358
//                       choiceTInit :367-445, getSDF...AtTimeStamp :537-572, getSDF_DOTAtTimeStampOptimized :593-662,
//                       getSDFofSweptVolume<> :710-747, gradientDescent :1000-1062
//   TrajOptimizer       getQuatTransD* back_end_optimizer.hpp:127-173, smoothedL1 :332-356,
//                       addTimeIntPenaltyParallel :432-554, addSaftyPenaOnSweptVolumeParallel :557-649,
//                       grad_cost_p :766-824, grad_cost_p_sw :827-866
//   PCSmapManager       projInMap + getPointsInAABB PCSmap_manager.h:130-170; unifiedID :101-108, getPointsInAABBOutOfLastOne :182-216,
//                       aabb_points :238 (the obstacle-point set of the swept-volume sweep, driven like plan_manager.cpp:232-251)
//   AstarPathSearcher   GridNode front_end_Astar.hpp:12-57, initGridMap :123-158, reset :160-170, getHeu / getCustomCost / AstarGetSucc /
//                       AstarPathSearch :172-357, getPath :359-376 (the SE(3) A* loop of the front end)
//   GridMap3D           Gridmap3D.cpp:41-258 (isInMap .. isIndexOccupied), generateESDF3d + fillESDF :361-533,
//                       getGridSDFValue :535-553, trilinear samplers GridMap3D.h:114-195
// sw_manager.hpp / back_end_optimizer.hpp / PCSmap_manager.h as a whole need ROS, PCL, libigl and yaml, so the Makefile
// EXTRACTS exactly those line ranges from /root/reference into oracle/_ref/gen/ at build time (generated, git-ignored,
// never committed); this file supplies the class shells (the data members those functions read) around them.
// flatness.hpp, trajectory.hpp, minco.hpp are included whole.  Eigen is the stand-in of eigen_shim/.
// TEST INFRASTRUCTURE ONLY.
#include "ref_shapes_common.hpp"
#include <cfloat>
#include <chrono>
#include <queue>
#include <unordered_map>
#include <map>
#include <cstring>
#include <ros/ros.h>
#include <cassert>
#include <limits>
#include <utils/flatness.hpp>
#include <utils/minco.hpp>          // brings trajectory.hpp (Trajectory<5>, Piece)
#define TRAJ_ORDER 5
#define ROS_ERROR(...) do { } while (0)
#define ROS_WARN(...) do { } while (0)
using Eigen::Vector3i; using Eigen::Vector3d; using Eigen::Matrix3d;

// ByteShapeKernel::generateByteKernel (Shape.hpp:238) allocates its bit array with `new uint8_t[n]` - NOT value-initialised -
// and then only ORs bits into it, i.e. the reference relies on fresh heap memory being zero.  Inside this library (linked
// -Bsymbolic) array-new hands out zeroed memory, which is the one behaviour under which that code does what it means.
#include <cstdlib>
#include <new>
void *operator new[](std::size_t n) { void *p = std::calloc(n ? n : 1, 1); if (!p) throw std::bad_alloc(); return p; }
void operator delete[](void *p) noexcept { std::free(p); }
void operator delete[](void *p, std::size_t) noexcept { std::free(p); }

// ---------------------------------------------------------------- the mesh Generalshape (utils/src/Shape.cpp:105-151)
// Generalshape's four SDF members are EXTRACTED from Shape.cpp into gen/shape_mesh.inc (the Makefile renames the class
// qualifier to MeshGeneralshape, nothing else) and compiled against the two libigl calls they make:
//   igl::fast_winding_number(fwn_bvh, 2.0, p, w)   the vendored HDK core (igl/FastWindingNumberForSoups.h, included from
//                                                  /root/reference) behind the thin wrapper libigl 2.4.0 puts around it
//                                                  (cast the query to float, w = computeSolidAngle(q, accuracy) / (4 pi));
//   tree.squared_distance(V, F, p, i, c)           igl::AABB's exact closest point - exact geometry, here by exhaustive
//                                                  search over the triangles (the library's .cpp is absent).
// g_fwn_ulps != 0 (ref_mesh_perturb) moves the float solid angle by that many units of 2^-23 (one ulp of a float in [1, 2):
// what ONE of the per-triangle atan2f terms, which are of order one, can differ by between two libm implementations), with a
// pseudo-random sign per query - for the sensitivity test of the argmin-over-t chain.
#include <igl/FastWindingNumberForSoups.h>
namespace igl {
unsigned int default_num_threads(unsigned int) { return 1; }
static int g_fwn_ulps = 0;
struct FastWindingNumberBVH {
    FastWindingNumber::HDK_Sample::UT_SolidAngle<float, float> ut_solid_angle;
    std::vector<FastWindingNumber::HDK_Sample::UT_Vector3T<float>> U;
    std::vector<int> F;
};
inline void fast_winding_number(const FastWindingNumberBVH &bvh, const float accuracy_scale, const Eigen::RowVector3d &Q, Eigen::VectorXd &WN) {
    FastWindingNumber::HDK_Sample::UT_Vector3T<float> Pt;
    Pt[0] = Q(0); Pt[1] = Q(1); Pt[2] = Q(2);
    float sa = bvh.ut_solid_angle.computeSolidAngle(Pt, accuracy_scale);
    if (g_fwn_ulps) {
        uint32_t hsh; float f0 = Pt[0] * 3.1f + Pt[1] * 5.3f + Pt[2] * 7.7f; std::memcpy(&hsh, &f0, 4);
        hsh ^= hsh >> 13; hsh *= 0x9E3779B1u; hsh ^= hsh >> 15;
        sa += ((hsh & 1u) ? 1.0f : -1.0f) * (float)g_fwn_ulps * 1.1920929e-07f;
    }
    WN(0) = sa / (4.0 * 3.1415926535897932384626433832795);
}
struct MeshTree {
    // exact closest point on the triangle soup (igl::AABB::squared_distance returns the same point and distance)
    double squared_distance(const Eigen::MatrixXd &V, const std::vector<int> &F, const Eigen::RowVector3d &p, int &idx, Eigen::RowVector3d &c) const {
        double best = 1.7976931348623157e308;
        const double P[3] = {p(0), p(1), p(2)};
        for (size_t f = 0; f < F.size() / 3; f++) {
            double a[3], b[3], cc[3];
            for (int k = 0; k < 3; k++) { a[k] = V(F[3 * f], k); b[k] = V(F[3 * f + 1], k); cc[k] = V(F[3 * f + 2], k); }
            // Ericson, Real-Time Collision Detection 5.1.5
            double ab[3], ac[3], ap[3], q[3];
            for (int k = 0; k < 3; k++) { ab[k] = b[k] - a[k]; ac[k] = cc[k] - a[k]; ap[k] = P[k] - a[k]; }
            auto dot = [](const double *x, const double *y) { return x[0] * y[0] + x[1] * y[1] + x[2] * y[2]; };
            const double d1 = dot(ab, ap), d2 = dot(ac, ap);
            bool done = false;
            if (d1 <= 0 && d2 <= 0) { for (int k = 0; k < 3; k++) q[k] = a[k]; done = true; }
            double bp[3], cp[3], d3 = 0, d4 = 0, d5 = 0, d6 = 0;
            if (!done) {
                for (int k = 0; k < 3; k++) bp[k] = P[k] - b[k];
                d3 = dot(ab, bp); d4 = dot(ac, bp);
                if (d3 >= 0 && d4 <= d3) { for (int k = 0; k < 3; k++) q[k] = b[k]; done = true; }
            }
            if (!done) {
                const double vc = d1 * d4 - d3 * d2;
                if (vc <= 0 && d1 >= 0 && d3 <= 0) { const double v = d1 / (d1 - d3); for (int k = 0; k < 3; k++) q[k] = a[k] + v * ab[k]; done = true; }
            }
            if (!done) {
                for (int k = 0; k < 3; k++) cp[k] = P[k] - cc[k];
                d5 = dot(ab, cp); d6 = dot(ac, cp);
                if (d6 >= 0 && d5 <= d6) { for (int k = 0; k < 3; k++) q[k] = cc[k]; done = true; }
            }
            if (!done) {
                const double vb = d5 * d2 - d1 * d6;
                if (vb <= 0 && d2 >= 0 && d6 <= 0) { const double w = d2 / (d2 - d6); for (int k = 0; k < 3; k++) q[k] = a[k] + w * ac[k]; done = true; }
            }
            if (!done) {
                const double va = d3 * d6 - d5 * d4;
                if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) {
                    const double w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
                    for (int k = 0; k < 3; k++) q[k] = b[k] + w * (cc[k] - b[k]);
                    done = true;
                }
            }
            if (!done) {
                const double va = d3 * d6 - d5 * d4, vb = d5 * d2 - d1 * d6, vc = d1 * d4 - d3 * d2;
                const double den = 1.0 / (va + vb + vc), v = vb * den, w = vc * den;
                for (int k = 0; k < 3; k++) q[k] = a[k] + ab[k] * v + ac[k] * w;
            }
            const double d2q = (P[0] - q[0]) * (P[0] - q[0]) + (P[1] - q[1]) * (P[1] - q[1]) + (P[2] - q[2]) * (P[2] - q[2]);
            if (d2q < best) { best = d2q; idx = (int)f; c = Eigen::RowVector3d(q[0], q[1], q[2]); }
        }
        return best;
    }
};
}  // namespace igl
namespace shape {
class MeshGeneralshape : public Generalshape {
public:
    Eigen::MatrixXd V;
    std::vector<int> F;
    igl::MeshTree tree;
    igl::FastWindingNumberBVH fwn_bvh;
    MeshGeneralshape(const Config &c, const double *Vin, int nV, const int *Fin, int nF) : Generalshape(c, false) {
        V.resize(nV, 3);
        for (int i = 0; i < nV; i++) for (int k = 0; k < 3; k++) V(i, k) = Vin[3 * i + k];
        F.assign(Fin, Fin + 3 * (size_t)nF);
        // igl::fast_winding_number(V, F, 2, fwn_bvh) (Shape.cpp:86): vertices cast to float, order-2 expansion
        fwn_bvh.U.resize(nV);
        for (int i = 0; i < nV; i++) for (int k = 0; k < 3; k++) fwn_bvh.U[i][k] = (float)Vin[3 * i + k];
        fwn_bvh.F = F;
        fwn_bvh.ut_solid_angle.init(nF, fwn_bvh.F.data(), nV, fwn_bvh.U.data(), 2);
    }
    double getonlySDF(const Eigen::RowVector3d &pos_rel) override;
    double getonlySDF(const Eigen::RowVector3d &pos_rel, const Eigen::Matrix3d &R_obj) override;
    Eigen::Vector3d getonlyGrad1(const Eigen::RowVector3d &pos_rel) override;
    double getSDFwithGrad1(const Eigen::RowVector3d &pos_rel, Eigen::Vector3d &grad) override;
};
#include "../_ref/gen/shape_mesh.inc"         // Shape.cpp:105-151
}  // namespace shape

// ---------------------------------------------------------------- GridMap3D (map_manager)
class GridMap3D {
public:
    int X_size = 0, Y_size = 0, Z_size = 0, totalsize = 0;
    double grid_resolution = 1.0;
    Vector3d boundary_xyzmin, boundary_xyzmax;
    double *grid_map = nullptr, *grid_map_buffer_neg = nullptr, *grid_map_buffer_all = nullptr;
    bool *grid_map_flags = nullptr;
    double *grid_esdf = nullptr, *grid_esdf_buffer1 = nullptr, *grid_esdf_buffer2 = nullptr;
    ~GridMap3D() { delete[] grid_map; delete[] grid_map_buffer_neg; delete[] grid_map_buffer_all; delete[] grid_map_flags;
                   delete[] grid_esdf; delete[] grid_esdf_buffer1; delete[] grid_esdf_buffer2; }
    void createGridMap(const Vector3d &boundary_xyzmin, const Vector3d &boundary_xyzmax);
    bool isInMap(const Vector3d &pos_w) const;
    bool isIndexValid(const Vector3i &index) const;
    bool isIndexValid(const int ix, const int iy, const int iz) const;
    Vector3i getGridIndex(const Vector3d &pos_w);
    Vector3d getGridCubeCenter(int ix, int iy, int iz);
    Vector3d getGridCubeCenter(const Vector3i &index);
    bool isIndexOccupied(const Vector3i &index);
    bool isIndexOccupiedFlate(const Vector3i &index, const int flate_pix);
    bool isIndexOccupied(int ix, int iy, int iz);
    int getVoxelNum(int dim);
    void generateESDF3d();
    template <typename F_get_val, typename F_set_val>
    void fillESDF(F_get_val f_get_val, F_set_val f_set_val, int start, int end, int dim);
    double getGridSDFValue(const Vector3i &index);
    double getGridSDFValue(int ix, int iy, int iz);
#include "../_ref/gen/gridmap_h.inc"          // GridMap3D.h:114-197: getSDFValue, getSDFValueWithGrad, toAddr
};
#include "../_ref/gen/gridmap_cpp_a.inc"      // Gridmap3D.cpp:25-39 createGridMap, :41-258 isInMap .. isIndexOccupied
#include "../_ref/gen/gridmap_cpp_b.inc"      // Gridmap3D.cpp:339-353 getVoxelNum, :361-553 generateESDF3d, fillESDF, getGridSDFValue

// ---------------------------------------------------------------- PCSmapManager (map_manager)
class PCSmapManager {
public:
    typedef PCSmapManager *Ptr;
    GridMap3D *occupancy_map = nullptr;
    Vector3d boundary_xyzmin, boundary_xyzmax;
#include "../_ref/gen/pcsmap.inc"             // PCSmap_manager.h:130-170 projInMap, getPointsInAABB
#include "../_ref/gen/pcsmap_gather.inc"      // PCSmap_manager.h:101-108 unifiedID, :182-216 getPointsInAABBOutOfLastOne, :238 aabb_points
    static constexpr uint8_t or_mask[8] = {0x80, 0x40, 0x20, 0x10, 0x08, 0x04, 0x02, 0x01};      // PCSmap_manager.h:32
    uint8_t *mk = nullptr;
#include "../_ref/gen/pcsmap_kernel.inc"      // PCSmap_manager.h:46-78 generateMapKernel
};
constexpr uint8_t PCSmapManager::or_mask[8];

// ---------------------------------------------------------------- SweptVolumeManager (swept_volume)
class SweptVolumeManager {
public:
    typedef SweptVolumeManager *Ptr;
    Config config;
    Trajectory<TRAJ_ORDER> traj;
    double traj_duration = 0.0, t_min = 0.0, t_max = 0.0, momentum = 0.0;
    double veps = 0, cp = 0, dh_over_m = 0, grav = 0;            // set like the constructor does (sw_manager.hpp:205-216)
    flatness::FlatnessMap flatness;
    shape::Generalshape *current_robot_shape = nullptr;
#include "../_ref/gen/swm_a.inc"              // :287-358
#include "../_ref/gen/swm_b.inc"              // :367-445
#include "../_ref/gen/swm_c.inc"              // :537-572
#include "../_ref/gen/swm_d.inc"              // :593-662
#include "../_ref/gen/swm_e.inc"              // :710-747
#include "../_ref/gen/swm_f.inc"              // :1000-1062
    // front end: kernelConv / visit_kernels_by_distance / checkKernelValue (sw_manager.hpp:775-942) and what they read
    uint8_t *map_kernel = nullptr;
    int map_Xsize = 0, map_Ysize = 0, map_Zsize = 0, xkernel_size = 0, ykernel_size = 0, kernelsize = 0;
    bool isIndexValid(const int, const int, const int) const { return false; }      // empty bodies in the reference (:768-773);
    bool isIndexOccupied(int, int, int) { return false; }                             // only the never-used bool-kernel path calls them
#include "../_ref/gen/swm_g.inc"              // :775-942
};

// ---------------------------------------------------------------- TrajOptimizer (planner_algorithm)
class TrajOptimizer {
public:
    Config conf;
    double vmax = 0, omgmax = 0, thetamax = 0, weight_v = 0, weight_p = 0, weight_omg = 0, weight_theta = 0;
    double smooth_fac = 0, safety_hor = 0, bdx = 0, bdy = 0, bdz = 0;
    int integralRes = 1, threads_num = 1, parallel_points_num = 0;
    std::vector<Eigen::Vector3d> parallel_points;
    std::vector<double> lastTstar;
    SweptVolumeManager *sv_manager = nullptr;
    PCSmapManager *pcsmap_manager = nullptr;
    Trajectory<TRAJ_ORDER> step_traj;
    flatness::FlatnessMap flatmap;
#include "../_ref/gen/beo_a.inc"              // :127-173
#include "../_ref/gen/beo_b.inc"              // :332-356
#include "../_ref/gen/beo_c.inc"              // :432-554
#include "../_ref/gen/beo_d.inc"              // :557-649
#include "../_ref/gen/beo_e.inc"              // :766-824
#include "../_ref/gen/beo_f.inc"              // :827-866
};

// ---------------------------------------------------------------- AstarPathSearcher (planner_algorithm)
using namespace Eigen;
using namespace std;
#include "../_ref/gen/astar_node.inc"         // front_end_Astar.hpp:12-57 GridNode
class AstarPathSearcher {
public:
    GridNode ****GridNodeMap = nullptr;
    GridNode *terminatePtr = nullptr;
    int GLX_SIZE = 0, GLY_SIZE = 0, GLZ_SIZE = 0, GLYZ_SIZE = 0, GLXYZ_SIZE = 0;
    uint8_t *data = nullptr;
    double total_time = 0.0, total_kernel_time = 0.0;
    int total_kernel = 0;
    inline vector<Vector3d> getPath();
    inline void reset();
    inline void initGridMap(PCSmapManager::Ptr env, SweptVolumeManager::Ptr sv);
    inline void AstarGetSucc(GridNode *currentPtr, vector<GridNode *> &neighborPtrSets, vector<double> &edgeCostSets);
    inline void AstarPathSearch(Vector3d start, Vector3d end);
    inline double getHeu(GridNode *node1, GridNode *node2);
    inline double getCustomCost(GridNode *node_neighbor, GridNode *node_current);
    bool success_flag = false;
    Vector3i goalIdx;
    Vector3d start_pt, end_pt;
    SweptVolumeManager::Ptr sv_manager = nullptr;
    std::multimap<double, GridNode *> openSet;
    PCSmapManager::Ptr environment = nullptr;
};
#include "../_ref/gen/astar_init.inc"         // :123-158 initGridMap
#include "../_ref/gen/astar_search.inc"       // :160-357 reset, getHeu, getCustomCost, AstarGetSucc, AstarPathSearch
#include "../_ref/gen/astar_path.inc"         // :359-376 getPath

namespace {
struct World {
    AstarPathSearcher *astar = nullptr;
    Config conf;
    GridMap3D grid;
    PCSmapManager pcs;
    SweptVolumeManager sv;
    TrajOptimizer opt;
    shape::Generalshape *shape = nullptr;
    ~World() { delete shape; }
};
// Trajectory<5> from the optimizer's coefficient matrix, exactly like MINCO_S3NU::getTrajectory (minco.hpp:515-528)
void make_traj(int N, const double *T, const double *coeffs, Trajectory<5> &traj, Eigen::VectorXd &Tv, Eigen::MatrixX3d &b) {
    Tv.resize(N);
    b.resize(6 * N, 3);
    for (int i = 0; i < N; i++) Tv(i) = T[i];
    for (int c = 0; c < 3; c++) for (int r = 0; r < 6 * N; r++) b(r, c) = coeffs[(size_t)c * 6 * N + r];
    traj.clear();
    traj.reserve(N);
    for (int i = 0; i < N; i++) traj.emplace_back(Tv(i), b.block<6, 3>(6 * i, 0).transpose().rowwise().reverse());
}
}

extern "C" {
// cfg doubles: [safety_hor, weight_p, weight_v, weight_omg, weight_theta, vmax, omgmax, thetamax, smooth_fac, bd (= kernel_size*res),
//               mass, grav, horiz_drag, vert_drag, paras_drag, speed_eps, occupancy_resolution];  K = integralIntervs
void *ref_world_create(const double *cfg, int K, int threads, const char *shape_name, const double *trans, const double *rotate,
                       const double *box) {
    World *w = new World();
    Config &c = w->conf;
    c.safety_hor = cfg[0]; c.occupancy_resolution = cfg[16];
    c.trans = Eigen::RowVector3d(trans[0], trans[1], trans[2]);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) c.rotate(i, j) = rotate[3 * i + j];
    if (box) { c.box_x = box[0]; c.box_y = box[1]; c.box_z = box[2]; }
    w->shape = make(shape_name, c);
    if (!w->shape) { delete w; return nullptr; }
    SweptVolumeManager &sv = w->sv;
    sv.config = c;
    sv.current_robot_shape = w->shape;
    sv.flatness.reset(cfg[10], cfg[11], cfg[12], cfg[13], cfg[14], cfg[15]);
    // the members getSDF_DOTAtTimeStampOptimized reads (sw_manager.hpp constructor)
    sv.veps = cfg[15]; sv.cp = cfg[14]; sv.dh_over_m = cfg[12] / cfg[10]; sv.grav = cfg[11];
    TrajOptimizer &o = w->opt;
    o.conf = c;
    o.safety_hor = cfg[0]; o.weight_p = cfg[1]; o.weight_v = cfg[2]; o.weight_omg = cfg[3]; o.weight_theta = cfg[4];
    o.vmax = cfg[5]; o.omgmax = cfg[6]; o.thetamax = cfg[7]; o.smooth_fac = cfg[8];
    o.bdx = o.bdy = o.bdz = cfg[9];
    o.integralRes = K; o.threads_num = threads;
    o.sv_manager = &w->sv;
    o.pcsmap_manager = &w->pcs;
    o.flatmap.reset(cfg[10], cfg[11], cfg[12], cfg[13], cfg[14], cfg[15]);
    w->pcs.occupancy_map = &w->grid;
    return w;
}
void ref_world_destroy(void *h) { delete (World *)h; }
// replaces the robot by the mesh Generalshape over (V, F) - vertices already in the body frame, i.e. after the constructor's
// poly_params transform (Shape.cpp:36-49)
void ref_world_set_mesh(void *h, const double *V, int nV, const int *F, int nF) {
    World *w = (World *)h;
    delete w->shape;
    w->shape = new shape::MeshGeneralshape(w->conf, V, nV, F, nF);
    w->sv.current_robot_shape = w->shape;
}
void ref_mesh_perturb(int ulps) { igl::g_fwn_ulps = ulps; }
// the robot shape by itself: getonlySDF + getonlyGrad1 on n body-frame points
void ref_world_shape_eval(void *h, const double *P, int n, double *sdf, double *grad) {
    World *w = (World *)h;
    for (int i = 0; i < n; i++) {
        const Eigen::RowVector3d p(P[3 * i], P[3 * i + 1], P[3 * i + 2]);
        sdf[i] = w->shape->getonlySDF(p);
        const Eigen::Vector3d g = w->shape->getonlyGrad1(p);
        for (int k = 0; k < 3; k++) grad[3 * i + k] = g(k);
    }
}

// ---- front end: the reference's kernel generation, map kernel and checkKernelValue
// fe: [kernel_size, kernel_max_roll, kernel_max_pitch, kernel_ang_res, front_end_safeh]; the shape is re-made with them
int ref_frontend_build(void *h, const double *fe, const char *shape_name) {
    World *w = (World *)h;
    Config &c = w->conf;
    c.kernel_size = (int)fe[0]; c.kernel_max_roll = fe[1]; c.kernel_max_pitch = fe[2]; c.kernel_ang_res = fe[3]; c.front_end_safeh = fe[4];
    delete w->shape;
    w->shape = make(shape_name, c);
    if (!w->shape) return -1;
    w->shape->initShapeKernels();
    SweptVolumeManager &sv = w->sv;
    sv.config = c;
    sv.current_robot_shape = w->shape;
    sv.xkernel_size = floor(2 * c.kernel_max_roll / c.kernel_ang_res) + 1;         // sw_manager.hpp:135-137
    sv.ykernel_size = floor(2 * c.kernel_max_pitch / c.kernel_ang_res) + 1;
    sv.kernelsize = c.kernel_size;
    delete[] sv.map_kernel;
    uint8_t *mk = w->pcs.generateMapKernel(c.kernel_size);                            // plan_manager.cpp:400-406
    sv.map_kernel = mk; sv.map_Xsize = w->grid.X_size; sv.map_Ysize = w->grid.Y_size; sv.map_Zsize = w->grid.Z_size;      // setMapKernel, sw_manager.hpp:158-164
    return 0;
}
void ref_frontend_dims(void *h, int *out) {
    World *w = (World *)h;
    const int k = w->conf.kernel_size, hh = (k - 1) / 2;
    out[0] = w->sv.xkernel_size; out[1] = w->sv.ykernel_size; out[2] = k * k * ((k + 7) / 8);
    out[3] = w->grid.X_size + 2 * hh; out[4] = w->grid.Y_size + 2 * hh; out[5] = (w->grid.Z_size + 2 * hh + 7) / 8;
}
void ref_frontend_shape_kernels(void *h, unsigned char *out) {
    World *w = (World *)h;
    const int k = w->conf.kernel_size, per = k * k * ((k + 7) / 8), n = w->sv.xkernel_size * w->sv.ykernel_size;
    for (int i = 0; i < n; i++) memcpy(out + (size_t)i * per, w->shape->byte_shape_kernels[i].map, per);
}
void ref_frontend_map_kernel(void *h, unsigned char *out) {
    World *w = (World *)h;
    int d[6]; ref_frontend_dims(h, d);
    memcpy(out, w->sv.map_kernel, (size_t)d[3] * d[4] * d[5]);
}
int ref_frontend_check(void *h, const int *ind, double fr, double fp, double *cr, double *cp) {
    World *w = (World *)h;
    return w->sv.checkKernelValue(fr, fp, *cr, *cp, Vector3i(ind[0], ind[1], ind[2])) ? 1 : 0;
}
int ref_frontend_kernel_conv(void *h, int ki, int kj, const int *ind) {
    World *w = (World *)h;
    return w->sv.kernelConv<true>(ki, kj, Vector3i(ind[0], ind[1], ind[2])) ? 1 : 0;
}

// the SE(3) A* of the front end, driven like PlannerManager::generatePath (plan_manager.cpp:181-198): AstarPathSearch, getPath,
// the nodes' roll / pitch along the father chain (what getastarSE3Path :378-403 turns into rotations), reset.
// Returns -1 when the search fails, else the number of path nodes (written up to cap).  stats: [total_kernel]
int ref_astar_search(void *h, const double *start, const double *goal, int cap, double *path_xyz, double *path_rp, long *stats) {
    World *w = (World *)h;
    if (!w->astar) {
        w->astar = new AstarPathSearcher();
        w->astar->initGridMap(&w->pcs, &w->sv);          // plan_manager.cpp:408
    }
    AstarPathSearcher &a = *w->astar;
    a.total_kernel = 0;
    a.AstarPathSearch(Vector3d(start[0], start[1], start[2]), Vector3d(goal[0], goal[1], goal[2]));
    int n = -1;
    if (a.success_flag) {
        vector<Vector3d> path = a.getPath();
        n = (int)path.size();
        vector<GridNode *> chain;
        for (GridNode *p = a.terminatePtr; p; p = p->father) chain.push_back(p);
        assert((int)chain.size() == n);
        for (int i = 0; i < n && i < cap; i++) {
            for (int k = 0; k < 3; k++) path_xyz[3 * i + k] = path[i](k);
            path_rp[2 * i] = chain[n - 1 - i]->roll; path_rp[2 * i + 1] = chain[n - 1 - i]->pitch;
        }
    }
    if (stats) stats[0] = a.total_kernel;
    a.reset();
    return n;
}

// occupancy grid (uint8, z fastest) -> GridMap3D::grid_map; optionally the reference's own generateESDF3d
void ref_world_set_occupancy(void *h, const unsigned char *occ, int nx, int ny, int nz, const double *bmin, const double *bmax, double res) {
    World *w = (World *)h;
    GridMap3D &g = w->grid;
    g.grid_resolution = res;
    Vector3d lo(bmin[0], bmin[1], bmin[2]), hi(bmax[0], bmax[1], bmax[2]);
    g.createGridMap(lo, hi);
    assert(g.X_size == nx && g.Y_size == ny && g.Z_size == nz);
    for (int i = 0; i < g.totalsize; i++) g.grid_map[i] = occ[i] ? 1.0 : 0.0;
    w->pcs.boundary_xyzmin = lo; w->pcs.boundary_xyzmax = hi;
}
void ref_world_generate_esdf(void *h, double *out) {
    World *w = (World *)h;
    w->grid.generateESDF3d();
    for (int i = 0; i < w->grid.totalsize; i++) out[i] = w->grid.grid_esdf[i];
}
// trilinear samplers of GridMap3D.h:114-193 on the reference's ESDF
double ref_world_esdf_sample(void *h, const double *pos, double *grad) {
    World *w = (World *)h;
    Vector3d p(pos[0], pos[1], pos[2]), g;
    const double v = w->grid.getSDFValueWithGrad(p, g);
    for (int i = 0; i < 3; i++) grad[i] = g(i);
    return v;
}
int ref_world_points_in_aabb(void *h, const double *center, double half, double *out, int cap) {
    World *w = (World *)h;
    std::vector<Eigen::Vector3d> pts;
    w->pcs.getPointsInAABB(Vector3d(center[0], center[1], center[2]), half, half, half, pts);
    const int n = (int)pts.size();
    for (int i = 0; i < std::min(n, cap); i++) for (int a = 0; a < 3; a++) out[3 * i + a] = pts[i](a);
    return n;
}
// The obstacle-point set as the planner gathers it (plan_manager.cpp:232-251): aabb_points cleared, then for every waypoint
// getPointsInAABBOutOfLastOne(wp, previous waypoint (first: (999, 999, 999)), half, offset).  Returns the number of points; ids / xyz
// (capacity `cap`) receive the map's entries in ITS iteration order (unspecified: callers compare as sets keyed by the voxel id).
int ref_world_gather_points(void *h, const double *way, int n_way, const double *half, const double *offset, int *ids, double *xyz, int cap) {
    World *w = (World *)h;
    PCSmapManager &m = w->pcs;
    m.aabb_points.clear();
    const Eigen::Vector3d off(offset[0], offset[1], offset[2]);
    Eigen::Vector3d tmp_pos(999, 999, 999);
    for (int i = 0; i < n_way; i++) {
        const Eigen::Vector3d wp(way[3 * i], way[3 * i + 1], way[3 * i + 2]);
        m.getPointsInAABBOutOfLastOne(wp, tmp_pos, half[0], half[1], half[2], off);
        tmp_pos = wp;
    }
    int n = 0;
    for (const auto &pair : m.aabb_points) {
        if (n < cap) { if (ids) ids[n] = pair.first; if (xyz) for (int a = 0; a < 3; a++) xyz[3 * n + a] = pair.second(a); }
        n++;
    }
    return n;
}
void ref_world_set_points(void *h, const double *xyz, int M) {
    World *w = (World *)h;
    w->opt.parallel_points.clear();
    for (int i = 0; i < M; i++) w->opt.parallel_points.push_back(Vector3d(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]));
    w->opt.parallel_points_num = M;
    w->opt.lastTstar.assign(M, 0.0);
}

// addSaftyPenaOnSweptVolumeParallel: accumulates into cost / gradT / gradC (column-major 6N x 3); tstar: M
void ref_sweep_swept(void *h, int N, const double *T, const double *coeffs, double *cost, double *gradT, double *gradC, double *tstar) {
    World *w = (World *)h;
    Eigen::VectorXd Tv; Eigen::MatrixX3d b;
    make_traj(N, T, coeffs, w->opt.step_traj, Tv, b);
    w->sv.updateTraj(w->opt.step_traj);                 // costFunctionLmbm :381-382
    Eigen::VectorXd gT(N); Eigen::MatrixX3d gC(6 * N, 3);
    for (int i = 0; i < N; i++) gT(i) = gradT[i];
    for (int c = 0; c < 3; c++) for (int r = 0; r < 6 * N; r++) gC(r, c) = gradC[(size_t)c * 6 * N + r];
    for (size_t i = 0; i < w->opt.lastTstar.size(); i++) w->opt.lastTstar[i] = tstar[i];
    double cst = *cost;
    TrajOptimizer::addSaftyPenaOnSweptVolumeParallel(&w->opt, Tv, b, cst, gT, gC);
    *cost = cst;
    for (int i = 0; i < N; i++) gradT[i] = gT(i);
    for (int c = 0; c < 3; c++) for (int r = 0; r < 6 * N; r++) gradC[(size_t)c * 6 * N + r] = gC(r, c);
    for (size_t i = 0; i < w->opt.lastTstar.size(); i++) tstar[i] = w->opt.lastTstar[i];
}
// addTimeIntPenaltyParallel
void ref_sweep_integral(void *h, int N, const double *T, const double *coeffs, double *cost, double *gradT, double *gradC) {
    World *w = (World *)h;
    Eigen::VectorXd Tv; Eigen::MatrixX3d b;
    Trajectory<5> tr;
    make_traj(N, T, coeffs, tr, Tv, b);
    Eigen::VectorXd gT(N); Eigen::MatrixX3d gC(6 * N, 3);
    for (int i = 0; i < N; i++) gT(i) = gradT[i];
    for (int c = 0; c < 3; c++) for (int r = 0; r < 6 * N; r++) gC(r, c) = gradC[(size_t)c * 6 * N + r];
    double cst = *cost, pos_cost = 0.0;
    TrajOptimizer::addTimeIntPenaltyParallel(&w->opt, Tv, b, cst, pos_cost, gT, gC);
    *cost = cst;
    for (int i = 0; i < N; i++) gradT[i] = gT(i);
    for (int c = 0; c < 3; c++) for (int r = 0; r < 6 * N; r++) gradC[(size_t)c * 6 * N + r] = gC(r, c);
}
// one swept-volume query: getSDFofSweptVolume<true>(p, t*, grad, false)
double ref_swept_sdf(void *h, int N, const double *T, const double *coeffs, const double *p, double *tstar, double *grad) {
    World *w = (World *)h;
    Eigen::VectorXd Tv; Eigen::MatrixX3d b;
    make_traj(N, T, coeffs, w->opt.step_traj, Tv, b);
    w->sv.updateTraj(w->opt.step_traj);
    Vector3d g(grad[0], grad[1], grad[2]);
    double ts = *tstar;
    const double s = w->sv.getSDFofSweptVolume<true>(Vector3d(p[0], p[1], p[2]), ts, g, false);
    *tstar = ts;
    for (int i = 0; i < 3; i++) grad[i] = g(i);
    return s;
}
// grad_cost_p for one pose (pos, quat w x y z): returns costp, fills gradp[3], grad_quat[4]
double ref_grad_cost_p(void *h, const double *pos, const double *quat, double *gradp, double *gradq, int *active) {
    World *w = (World *)h;
    Eigen::Vector4d q(quat[0], quat[1], quat[2], quat[3]);
    Matrix3d R = Eigen::Quaterniond(quat[0], quat[1], quat[2], quat[3]).toRotationMatrix();
    Vector3d gp; Eigen::Vector4d gq; double costp = 0.0;
    const bool a = w->opt.grad_cost_p(Vector3d(pos[0], pos[1], pos[2]), R, q, gp, gq, costp);
    *active = a ? 1 : 0;
    for (int i = 0; i < 3; i++) gradp[i] = gp(i);
    for (int i = 0; i < 4; i++) gradq[i] = gq(i);
    return costp;
}
}
