/*
 * isdf_accel.h — C ABI of the MI355X-native collision cost/gradient engine.
 *
 * This is the drop-in boundary for ONE hot path of ZJU-FAST-Lab/Implicit-SDF-Planner: the per-optimizer-step
 * sweeps that TrajOptimizer::costFunctionLmbm calls
 *   (src/planner_algorithm/include/planner_algorithm/back_end_optimizer.hpp:386-391 and :399-405).
 * Plain C, plain pointers and sizes; no Eigen / torch / C++ types cross it.  Every entry point returns
 * ISDF_OK (0) or a negative isdf_status and never throws; isdf_last_error() gives the message.
 *
 * Conventions (SURVEY.md §8 "Conventions"):
 *   N  = pieces of one trajectory,  K = integralIntervs (K+1 samples per piece),  M = obstacle points.
 *   coeffs = the optimizer's Eigen::MatrixX3d, 6N x 3, COLUMN-major: element (r,c) at data[c*6N + r];
 *            rows 6i..6i+5 are the ascending-power coefficients c0..c5 of piece i (minco.hpp:402,545).
 *   gradC  has the same layout; gradT and T have N entries.
 *   Outputs of the host entry points are ACCUMULATED (+=) into caller-owned storage, exactly like
 *   addSaftyPenaOnSweptVolumeParallel / addTimeIntPenaltyParallel (back_end_optimizer.hpp:557-562, :432-438).
 *   A ctx may be used by one host thread at a time (the reference's LMBM trampolines are process-global
 *   statics, lmbm.cpp:4-6, so the reference is not re-entrant either).
 */
#ifndef ISDF_ACCEL_H
#define ISDF_ACCEL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ISDF_ABI_VERSION 1

typedef enum isdf_status {
    ISDF_OK = 0,
    ISDF_ERR_INVALID_ARG = -1,
    ISDF_ERR_NO_DEVICE = -2,     /* no usable HIP device: the product path has NO CPU fallback */
    ISDF_ERR_HIP = -3,           /* a HIP runtime call failed; see isdf_last_error */
    ISDF_ERR_STATE = -4,         /* grid / shape / points not set for the requested sweep */
    ISDF_ERR_OVERFLOW = -5,      /* a bounded device-side work list overflowed (result NOT valid) */
    ISDF_ERR_UNSUPPORTED = -6
} isdf_status;

/* Which sweep isdf_eval runs.  V1 is the reference's live path, V2 its dormant integral path, V3 the
 * ESDF-tile kernel of BASELINE.json's north_star (SURVEY.md §8 "Variants", Appendix A.3-A.5). */
typedef enum isdf_variant {
    ISDF_V1_SWEPT = 1,     /* addSaftyPenaOnSweptVolumeParallel, back_end_optimizer.hpp:557-649            */
    ISDF_V2_OCC_TILE = 2,  /* addTimeIntPenaltyParallel :432-554 with grad_cost_p :766-824 (occupancy tile)  */
    ISDF_V3_ESDF_TILE = 3  /* same sweep, voxel qualifies iff esdf <= occ_thresh, optional whole-tile cull   */
} isdf_variant;

typedef enum isdf_grid_kind { ISDF_GRID_OCCUPANCY = 0, ISDF_GRID_ESDF = 1 } isdf_grid_kind;
typedef enum isdf_dtype { ISDF_U8 = 0, ISDF_F32 = 1, ISDF_F64 = 2 } isdf_dtype;

/* Robot-shape plugin kinds = the analytic-shape registry of
 * src/swept_volume/include/swept_volume/sw_manager.hpp:74-123 plus Box/Ball/mesh.
 * params[] meaning per kind (reference constants are what isdf_shape_default() fills in):
 *   TORUS              [0]=major radius (2.5 | Torus_big 3.5) [1]=minor radius (0.3)            Shape.hpp:824-893
 *   CAPPEDTORUS        [0]=sc.x (sin(40 rad)) [1]=sc.y (cos(40 rad)) [2]=ra 3.5 [3]=rb 0.3      Shape.hpp:895-931
 *   CAPPEDCONE         [0]=ra 2 [1]=rb 0.8 [2..4]=a (0,0,-1) [5..7]=b (0,0,1)                   Shape.hpp:933-998
 *   ROUNDEDCONE        [0]=r1 1.5 [1]=r2 0.6 [2]=h 4.5                                          Shape.hpp:1000-1047
 *   WIREFRAMEBOX       [0..2]=size (1.8,2.5,3.5) [3]=thickness 0.1                              Shape.hpp:1049-1103
 *   BENDLINEAR         [0]=capsule half length (2 | _big 3.2) [1]=radius (0.25 | _big 0.45)     Shape.hpp:1105-1234
 *   TWISTBOX           [0..2]=size (2,2,2) [3]=k (pi/6)                                         Shape.hpp:1236-1288
 *   BENDBOX            [0..2]=size (2,2,2) [3]=k (0.5)                                          Shape.hpp:1290-1341
 *   TABLE              [0..2]=a1 [3..5]=b1 [6..8]=a2 [9..11]=b2                                 Shape.hpp:1343-1405
 *   TREFOIL            [0]=r 3.5 [1],[2]=box half sizes 0.2 [3]=rounding 0.05 [4]=scale 0.4     Shape.hpp:1442-1515
 *   SMOOTHDIFFERENCE   [0..2]=box size (3,3,0.5) [3]=sphere radius 1 [4]=k 0.25                 Shape.hpp:1517-1570
 *   SMOOTHINTERSECTION [0..2]=box size (3,3,0.5 | _big 9,9,1.5) [3]=radius (1 | 3) [4]=k 0.25   Shape.hpp:1572-1682
 *   CSG                [0]=sphere r 3 [1]=box edge 4.5 [2]=cylinder r 1.5                       Shape.hpp:1684-2317
 *   BOX                [0..2]=half extents (conf.box_x/y/z)                                     Shape.hpp:2320-2390
 *   BALL               [0]=radius (Point == radius 0)                                           Shape.hpp:603-665
 *   MESH               triangle soup: sign from the fast winding number, distance to the closest triangle
 *                                                                                               Shape.cpp:105-151 */
typedef enum isdf_shape_kind {
    ISDF_SHAPE_TORUS = 0,
    ISDF_SHAPE_CAPPEDTORUS = 1,
    ISDF_SHAPE_CAPPEDCONE = 2,
    ISDF_SHAPE_ROUNDEDCONE = 3,
    ISDF_SHAPE_WIREFRAMEBOX = 4,
    ISDF_SHAPE_BENDLINEAR = 5,
    ISDF_SHAPE_TWISTBOX = 6,
    ISDF_SHAPE_BENDBOX = 7,
    ISDF_SHAPE_TABLE = 8,
    ISDF_SHAPE_TREFOIL = 9,
    ISDF_SHAPE_SMOOTHDIFFERENCE = 10,
    ISDF_SHAPE_SMOOTHINTERSECTION = 11,
    ISDF_SHAPE_CSG = 12,
    ISDF_SHAPE_BOX = 13,
    ISDF_SHAPE_BALL = 14,
    ISDF_SHAPE_MESH = 15,
    ISDF_SHAPE_GRID = 16,   /* a body-frame lattice of (unit gradient, distance) sampled from ANY host shape: isdf_set_shape_grid */
    ISDF_SHAPE_KIND_COUNT = 17
} isdf_shape_kind;

/* How getonlyGrad1 / getSDFwithGrad1 form the body-frame gradient. */
typedef enum isdf_grad_mode {
    ISDF_GRAD_DEFAULT = 0,      /* what the reference class of this kind does                                  */
    ISDF_GRAD_CENTRAL = 1,      /* DEFINE_USEFUL_FUNCTION: central difference dx=5e-6, normalised Shape.hpp:32-88 */
    ISDF_GRAD_BOX_FORWARD = 2,  /* Box::getonlyGrad1: forward difference dx=0.01, NOT normalised  Shape.hpp:2363-2377 */
    ISDF_GRAD_ANALYTIC_BALL = 3,/* Ball/Point: p / |p|                                           Shape.hpp:622-630 */
    ISDF_GRAD_GRID = 4          /* ISDF_SHAPE_GRID: trilinear blend of the nodes' gradients, normalised Shape.hpp:520-553 */
} isdf_grad_mode;

typedef struct isdf_shape {
    int32_t kind;          /* isdf_shape_kind */
    int32_t grad_mode;     /* isdf_grad_mode */
    double params[16];
    double trans[3];       /* yaml poly_params xyz; every analytic SDF first maps p -> (p - trans) * Rotate */
    double rotate[9];      /* row-major 3x3 Rotate = yaw*pitch*roll (Shape.cpp:38-43)                       */
    double bound_radius;   /* >0: a radius R with sdf(p) >= |p| - R for every body-frame point p (enables the V3
                              whole-tile cull); 0 disables the cull for this shape                            */
    double bbox_center[3]; /* body-frame box with sdf(p) >= distance(p, box) for every p outside it, i.e. the   */
    double bbox_half[3];   /* shape lies inside the box and its SDF never under-estimates by more than the box  */
                           /* does.  All bbox_half > 0 lets the scan prune voxel rows that cannot reach the      */
                           /* penalty band; zeros disable the pruning.  Results are identical either way.        */
    /* MESH only (already transformed into the body frame exactly as Generalshape's constructor does).  For a CLOSED mesh
       isdf_set_shape also samples a lattice of node-to-surface distances on the device (a few milliseconds, 4 MB): the tile
       sweep's pre-filter for this kind (it drops listed voxels that provably carry no penalty; results agree to rounding with
       and without it) and the swept-volume scans' way of telling which samples need a hierarchy query at all (bitwise the same
       results).  ISDF_NO_F32_FILTER=1 builds none. */
    const double *mesh_vertices;  /* nV x 3 row-major */
    const int32_t *mesh_faces;    /* nF x 3 row-major */
    int32_t n_vertices;
    int32_t n_faces;
} isdf_shape;

/* Mirrors the Config fields the hot path reads (src/utils/include/utils/config.hpp; yaml in
 * src/plan_manager/config).  Defaults of isdf_config_default() are config_CappedCone.yaml (demo1). */
typedef struct isdf_config {
    int32_t device;            /* HIP device ordinal                                                        */
    int32_t variant;           /* isdf_variant                                                              */
    int32_t kernel_size;       /* tile edge in voxels; bd = kernel_size * occupancy_resolution (:692)        */
    int32_t integral_intervs;  /* K                                                                         */
    int32_t enable_dyn;        /* 1: velocity / body-rate / tilt penalties of addTimeIntPenaltyParallel      */
    int32_t enable_pos;        /* 1: add the collision term grad_cost_p to that sweep (V2/V3)                */
    int32_t enable_cull;       /* V3: skip a pose whose trilinear esdf(pos) proves every penalty is zero     */
    int32_t reserved0;
    double safety_hor;
    double weight_p, weight_v, weight_omg, weight_theta;
    double vmax, omgmax, thetamax;
    double smoothing_eps;      /* mu of smoothedL1 in the integral sweep (V1 hard-codes 0.01, :851)          */
    double occ_thresh;         /* V3: voxel qualifies iff esdf <= occ_thresh (0 == "occupied")               */
    double vehicle_mass, grav_acc, horiz_drag, vert_drag, paras_drag, speed_eps;   /* flatness.hpp:36-51 */
} isdf_config;

typedef struct isdf_ctx isdf_ctx;   /* opaque; owns all device state */

/* ---- lifetime -------------------------------------------------------------------------------------------- */
void isdf_config_default(isdf_config *cfg);
int isdf_shape_default(isdf_shape *shape, int kind);       /* reference constants for an analytic kind   */
int isdf_shape_from_name(isdf_shape *shape, const char *obj_stem); /* registry lookup, sw_manager.hpp:74-123,
                                                              e.g. "RoundedCone", "Torus_big"; <0 if the
                                                              stem is not an analytic shape (=> mesh)     */
int isdf_create(isdf_ctx **out, const isdf_config *cfg);
/* ONE host process driving n_devices GPUs (SURVEY.md 8(b) "Threading"; the reference's caller is one ROS process,
 * back_end_optimizer.hpp:386-391,399-405): the returned ctx is used EXACTLY like a single-device one - cfg.device is ignored,
 * devices[0] is where device-resident arguments of isdf_eval_device live.  Once-per-plan state (isdf_set_grid / _shape /
 * _points / _pointcloud / _generate_esdf / _gather_points) is replicated on every device; every step (isdf_eval,
 * isdf_eval_device, isdf_cost_function[_lmbm], isdf_optimize_lbfgs) shards the constraint points by piece (V2 / V3) or by
 * obstacle point (V1) like isdf_set_shard would, queues every shard on its own device's stream FROM THE CALLING THREAD (no host
 * threads are created) and sums the shards' packed [cost | gradT | gradC] on devices[0] - ONE exchange per sweep:
 *   ISDF_MULTI_PEER_SUM  a kernel on devices[0] reading the peers' buffers over xGMI in rank order (default when peer access
 *                        exists; bitwise reproducible),
 *   ISDF_MULTI_STAGED    peer copies to devices[0] + the same sum (no peer access needed),
 *   ISDF_MULTI_RCCL      one ncclAllReduce(sum, ncclDouble) over all devices (environment ISDF_MULTI_COLLECTIVE=rccl; librccl.so
 *                        is dlopen-ed then - the library itself does not link RCCL; needs DISTINCT devices).
 * A device may be listed more than once (the shards then share it: how the single-GPU tests drive this path).
 * Not available on such a ctx: isdf_set_shard, isdf_xchg_*, isdf_optimize_lbfgs_batch, isdf_eval_swept_at_tstar. */
#define ISDF_MULTI_NONE 0
#define ISDF_MULTI_PEER_SUM 1
#define ISDF_MULTI_STAGED 2
#define ISDF_MULTI_RCCL 3
int isdf_create_multi(isdf_ctx **out, const isdf_config *cfg, const int *devices, int n_devices);
int isdf_multi_info(const isdf_ctx *ctx, int *n_devices_out, int *collective_out);
/* What isdf_set_shape found and decided about the installed MESH robot (all zero for another kind):
 *   [0] faces  [1] nodes of the winding-number hierarchy  [2] its depth  [3] 1: one swept-volume task per workgroup (quad walks)
 *   [4] 1: every edge is shared by two faces with opposite directions (closed, consistently oriented)
 *   [5] 1: the EXACT winding number is 0 / 1 on the two sides of every face (the surface bounds a solid: no nested sheets of one
 *       orientation, no overlapping or inverted bodies, no tears) - or becomes so within a tenth of the lattice's reach of the face:
 *       a DEFECT pocket of known thickness ([12], [13]; the reference's Trefoil.obj has 18 folded sliver faces), which the lattice's
 *       users allow for; 0: it is not; -1: not tested (no lattice wanted)
 *   [6..8] nodes of the distance lattice per axis (0: none - the mesh failed [4], [5] or the measured range below, or
 *       ISDF_NO_F32_FILTER=1; every query then walks the hierarchy, results are the same bits)
 *   [9], [10] 1e6 x the smallest / largest |1 - 2 w| the reference's approximate winding number took at the lattice's sample
 *       points away from the surface (nodes, cell centres, edge midpoints; a lattice is kept only for 0.96 ... 1.04)
 *   [11] (node, child) slots of the FLAT evaluation small meshes get in the swept-volume sweep (<= 64 slots: the reference's
 *       12- to 20-face robots; 0: the hierarchy is walked).
 *   [12] 1e9 x the thickness (m) of the thickest defect pocket found next to a face (0: a clean surface)  [13] 1e3 x the largest
 *       |1 - 2 w| inside one  [14], [15] 0.
 * Replaces: nothing (the reference builds its libigl structures without checks, Shape.cpp:60-103). */
int isdf_mesh_info(const isdf_ctx *ctx, int info_out[16]);
int isdf_destroy(isdf_ctx *ctx);
const char *isdf_last_error(const isdf_ctx *ctx);          /* ctx may be NULL: last create() failure      */
int isdf_abi_version(void);

/* ---- once-per-plan state (replaces TrajOptimizer::setGridMap / setEnvironment and the parallel_points
 *      assembly of plan_manager.cpp:232-254) ------------------------------------------------------------------ */
/* voxels: nx*ny*nz values, z fastest: addr = ix*ny*nz + iy*nz + iz (GridMap3D.h:194).
 * origin = boundary_xyzmin; boundary_max = boundary_xyzmax (NULL => origin + n*resolution; the reference
 * sizes the grid as ceil((max-min)/res), Gridmap3D.cpp:29-31, so max may lie inside the last voxel).
 * An occupancy grid and an ESDF grid may both be set (same geometry); V2 reads the former, V3 the latter. */
int isdf_set_grid(isdf_ctx *ctx, const void *voxels, int dtype, int nx, int ny, int nz,
                  const double origin[3], const double boundary_max[3], double resolution, int grid_kind);
int isdf_set_shape(isdf_ctx *ctx, const isdf_shape *shape);
/* ---- a robot shape the library has never seen: the reference's plugin promise is "subclass Generalshape, add a constructor to
 * the registry" (Shape.hpp:469-472, sw_manager.hpp:74-123); device code cannot call host virtuals, so such a shape comes in the
 * form the reference itself tabulates at start-up: BasicShape::initShape (Shape.hpp:361-404) fills num_sdf_map, a body-frame
 * lattice of NumSDFGridCell {unit gradient, distance} with node (i, j, k) at (-nd/2 + i*nres) per axis, ceil(nd / nres) nodes
 * per axis, by calling the shape's own getSDFwithGrad1 once per node; getonlySDFNum / getonlyGrad1Num / getSDFwithGrad1Num
 * (:481-600) sample it trilinearly (value: 1e20 outside the lattice; gradient: blend of the 8 nodes' gradients, normalised;
 * zero outside).  ISDF_SHAPE_GRID is exactly that sampler on the device - a documented APPROXIMATION of the shape it was
 * sampled from (error O(nres^2 x curvature) on the distance), exact parity with the reference's *Num functions.
 * isdf_set_shape_grid: cells = nx*ny*nz x 4 doubles (gx, gy, gz, distance), address (i*ny + j)*nz + k like toAddr (:466);
 * grid_min = position of node (0,0,0); bound_radius / bbox_* as in isdf_shape (0 / zeros: derived from the lattice box, which is
 * always valid because the sampler returns 1e20 outside it).
 * isdf_set_shape_sampled: builds the cells the way initShape does, by calling `fn` (== getSDFwithGrad1 of any host class:
 * returns the distance, writes the gradient) at every node, then installs them. */
typedef double (*isdf_sdf_with_grad_fn)(void *user, const double p_rel[3], double grad_out[3]);
int isdf_set_shape_grid(isdf_ctx *ctx, const double *cells, int nx, int ny, int nz, const double grid_min[3], double nres,
                        double bound_radius, const double *bbox_center, const double *bbox_half);
int isdf_set_shape_sampled(isdf_ctx *ctx, isdf_sdf_with_grad_fn fn, void *user, double ndx, double ndy, double ndz, double nres,
                           double bound_radius, const double *bbox_center, const double *bbox_half);

int isdf_set_points(isdf_ctx *ctx, const double *xyz, int M);   /* V1: M x 3 row-major obstacle points  */
/* Multi-GPU: this ctx evaluates only its share of the constraint points (pieces for V2/V3, obstacle points
 * for V1); outputs of all ranks SUM to the full result (one all-reduce of [cost|gradT|gradC] per step). */
int isdf_set_shard(isdf_ctx *ctx, int rank, int world_size);

/* ---- once-per-plan map products, built on the device ------------------------------------------------------------ */
/* Point cloud -> occupancy grid (PCSmapManager::rcvGlobalMapHandler, src/map_manager/src/PCSmap_manager.cpp:87-200):
 * xyz = n_points x 3 floats (pcl::PointXYZ).  bmin/bmax NULL => the tight box of the cloud ("measure boundary",
 * :110-141).  Grid size ceil((max-min)/res) (Gridmap3D.cpp:29-31); a voxel is occupied iff it collected >=
 * sta_threshold points; points outside the box count for voxel (0,0,0) like getGridIndex does (:137-140).
 * Replaces any grid set before.  dims_out (nullable) receives nx, ny, nz. */
int isdf_set_pointcloud(isdf_ctx *ctx, const float *xyz, long long n_points, const double *bmin, const double *bmax,
                        double resolution, int sta_threshold, int dims_out[3]);
/* Occupancy -> unsigned ESDF in metres (GridMap3D::generateESDF3d, Gridmap3D.cpp:361-414: the exact Euclidean distance
 * to the nearest occupied voxel centre, res*sqrt(d2)); installs it as the ISDF_GRID_ESDF grid (float32). */
int isdf_generate_esdf(isdf_ctx *ctx);
/* Download a grid (out may be NULL to query the geometry only).  dtype must be ISDF_F32 for the ESDF, ISDF_U8 for the
 * occupancy grid. */
int isdf_get_grid(isdf_ctx *ctx, int grid_kind, void *out, int dtype, int dims_out[3], double origin_out[3], double bmax_out[3]);
/* Obstacle-point set of the swept-volume sweep (plan_manager.cpp:232-254): for every waypoint, the occupied voxel
 * centres inside the box [w - half + offset, w + half + offset] that are outside the previous waypoint's box
 * [w_prev - half, w_prev + half] (getPointsInAABBOutOfLastOne, PCSmap_manager.h:182-216; the first "previous" is
 * (999,999,999)), united over the waypoints.  Installs the set like isdf_set_points (ordered by voxel index; the
 * reference's unordered_map order is unspecified) and resets lastTstar to 0.  offset may be NULL. */
int isdf_gather_points(isdf_ctx *ctx, const double *waypoints, int n_waypoints, const double half[3], const double *offset,
                       int *M_out);
/* Copies up to `capacity` points (M x 3 row-major) and returns M (>= 0), or a negative isdf_status. */
int isdf_get_points(isdf_ctx *ctx, double *xyz_out, int capacity);

/* The environment ESDF sampled at n WORLD points (xyz: n x 3 row-major): value_out[i] = GridMap3D::getSDFValue(pos)
 * (src/map_manager/include/map_manager/GridMap3D.h:114-146), grad_out[i] (n x 3) the analytic gradient of the trilinear interpolant
 * as getSDFValueWithGrad returns it (:155-193) - including the reference's quirks: base cell of pos - res/2, a position outside
 * the map reads cell (0,0,0) (Gridmap3D.cpp:137-140), an invalid corner reads 0 (:535-542).  Needs an ESDF grid (isdf_set_grid
 * with ISDF_GRID_ESDF, or isdf_generate_esdf).  Either output may be NULL.  Bit for bit the reference's doubles on the float32
 * grid.  The _device form takes device pointers and a stream (asynchronous). */
int isdf_esdf_sample(isdf_ctx *ctx, const double *xyz, long long n, double *value_out, double *grad_out);
int isdf_esdf_sample_device(isdf_ctx *ctx, const double *d_xyz, long long n, double *d_value_out, double *d_grad_out, void *stream);
/* The same for points in NO particular order (a point cloud, random queries): sampled from a bricked copy of the ESDF - 2 x 2 x 2-cell
 * bricks with their one-voxel apron, one 128-byte line each, so the eight corners of any cell come from ONE line instead of four
 * z-rows (4x the grid's memory, built once per map on first use).  Same arithmetic on the same values: results bit for bit those
 * of isdf_esdf_sample.  Points along a trajectory are better served by the plain form (neighbours share their lines). */
int isdf_esdf_sample_scattered(isdf_ctx *ctx, const double *xyz, long long n, double *value_out, double *grad_out);
int isdf_esdf_sample_scattered_device(isdf_ctx *ctx, const double *d_xyz, long long n, double *d_value_out, double *d_grad_out, void *stream);

/* The installed robot shape by itself, on n BODY-FRAME points (p_rel: n x 3 row-major): sdf_out[i] =
 * BasicShape::getonlySDF(pos_rel) and grad_out[i] (n x 3) = getonlyGrad1(pos_rel) (Shape.hpp:32-57: central difference
 * dx = 5e-6, normalised; Box :2363-2377; Ball :622-625; mesh Generalshape Shape.cpp:105-139: (1 - 2 w) * distance with
 * w = igl::fast_winding_number(fwn_bvh, 2.0, ...)).  Either output may be NULL.  Evaluated on the device. */
int isdf_shape_eval(isdf_ctx *ctx, const double *p_rel, int n, double *sdf_out, double *grad_out);

/* The float atan2 inside the mesh kind's winding number, evaluated on the HOST (no device needed; for tests): the reference's
 * UTsignedSolidAngleTri ends in the C library's atan2f (FastWindingNumberForSoups.h:325-326, :6083) and the swept-volume
 * argmin is sensitive to its last bit, so the device restates glibc's algorithm (sysdeps/ieee754/flt-32/e_atan2f.c,
 * s_atanf.c) operation for operation; out[i] = that function at (y[i], x[i]). */
int isdf_mesh_atan2f(const float *y, const float *x, long long n, float *out);

/* ---- per-step evaluation ----------------------------------------------------------------------------------- */
/* Host entry point, synchronous, drop-in for the reference sweeps.  n_traj trajectories (1 in the reference);
 * N[b] pieces each; T[b] -> N[b] doubles; coeffs[b] -> 6N[b] x 3 column-major.
 * cost_inout[b], gradT_inout[b][..], gradC_inout[b][..] are ACCUMULATED.  tstar_inout (V1 only, may be NULL,
 * n_traj == 1): lastTstar[M], read and written like TrajOptimizer::lastTstar (:59-62, :576-578). */
int isdf_eval(isdf_ctx *ctx, int n_traj, const int *N, const double *const *T,
              const double *const *coeffs, double *cost_inout, double *const *gradT_inout,
              double *const *gradC_inout, double *tstar_inout);

/* How the LAST host-array step (isdf_eval, isdf_cost_function) crossed PCIe - for callers that want to know which of the
 * boundary's latencies they are getting (INTEGRATION.md "Boundary cost"):
 *   COPY           one H2D copy, the launches, one D2H copy, one stream synchronisation (batches, V1, mesh robots, small-BAR
 *                  systems with ISDF_NO_HOST_DIRECT=1, any step that is not one fused launch)
 *   DIRECT_MAPPED  ONE launch: its first workgroups fetch the inputs from host-mapped memory, its last store the results and a
 *                  completion flag into host-mapped memory (no large PCIe BAR, or ISDF_NO_BAR_WRITES=1)
 *   DIRECT_BAR     ONE launch: the CPU has written the inputs straight into device memory through the PCIe BAR (verified per
 *                  ctx by a kernel-visible probe), results and flag as above
 * The three environment switches ISDF_NO_HOST_DIRECT / ISDF_NO_BAR_WRITES / ISDF_NO_FUSE are read once per ctx, in isdf_create.
 * Replaces nothing in the reference (its sweeps run in the caller's address space). */
#define ISDF_HOST_PATH_COPY 0
#define ISDF_HOST_PATH_DIRECT_MAPPED 1
#define ISDF_HOST_PATH_DIRECT_BAR 2
/*   DEVICE_CALLBACK  isdf_cost_function with its MINCO half on the device: x goes down (through the BAR, or fetched from
 *                  host-mapped memory), (cost, g) and a completion word come back into host-mapped memory; no copy commands */
#define ISDF_HOST_PATH_DEVICE_CALLBACK 3
int isdf_host_path(const isdf_ctx *ctx);
/* The DIRECT_* and DEVICE_CALLBACK hand-overs end with the kernel storing results and then a completion word into host-mapped
 * memory.  Those are separate PCIe writes from different wavefronts: the word can be visible to the CPU before every result is
 * (seen on MI355X for the FIRST step of a ctx, about one fresh process in twenty: cost and word there, the gradient rows still
 * zero).  The host therefore fills the result area with an all-ones pattern no result can have before the launch and, once the
 * word is there, waits until none of it is left.  info_out: [0] hand-overs so far, [1] how many of them had results still
 * missing when the word arrived, [2] polls spent waiting for those; [3..7] reserved (0). */
int isdf_host_info(const isdf_ctx *ctx, int64_t info_out[8]);

/* Device-resident entry point, asynchronous on `stream` (a hipStream_t passed as void*; NULL = default
 * stream).  All trajectories have N pieces.  d_T: n_traj*N, d_coeffs: n_traj * (6N x 3 col-major),
 * d_out: n_traj * (1 + N + 18N) doubles, OVERWRITTEN with [cost | gradT | gradC(col-major)] per trajectory
 * (shard-partial sums when a shard is set).  d_tstar: V1 only, M doubles, may be NULL.
 * Results change hands between workgroups through self-resetting slots behind BOUNDED waits (forward progress of a launch is
 * not something the hardware contract promises): a wait that expires never hangs the stream - it raises the overflow word
 * and leaves a NaN cost.  isdf_eval reports that as ISDF_ERR_OVERFLOW by itself; a caller of THIS entry point must look:
 * a non-finite d_out[0], or isdf_get_stats(...).overflow != 0 (reading it drains the device, clears the word and empties
 * every slot again, so that a producer that published late cannot feed the next step; the in-kernel exchange of a
 * multi-GPU step should be switched off after an overflow, isdf_xchg_fuse(ctx, 0)).
 * Steps of one ctx share its scratch (result slots, the dispatch order the device derives from earlier steps): they have to
 * execute one after the other - the same stream, or streams ordered by events as isdf_optimize_lbfgs_batch does. */
int isdf_eval_device(isdf_ctx *ctx, int n_traj, int N, const double *d_T, const double *d_coeffs,
                     double *d_out, double *d_tstar, void *stream);
size_t isdf_out_stride(int N);   /* 1 + N + 18N */
/* The swept-volume sweep with the minimisers GIVEN instead of searched for: obstacle point k is evaluated at time
 * d_tstar[k] (M doubles on the device; a negative or NaN entry means "no time interval qualified": min sdf = 10, no penalty,
 * sw_manager.hpp:717) - i.e. lines :578-646 of addSaftyPenaOnSweptVolumeParallel after getSDFofSweptVolume returned
 * (getSDFAtTimeStamp sw_manager.hpp:550-556 + getGradPrelAtTimeStamp :566-572 at that time).  Same output as
 * isdf_eval_device; the ctx's internal lastTstar takes the given values.  Use: re-evaluating cost / gradient at minimisers found elsewhere (the
 * argmin over t is a chain of accept/reject comparisons; this entry point isolates the arithmetic behind it). */
int isdf_eval_swept_at_tstar(isdf_ctx *ctx, int N, const double *d_T, const double *d_coeffs, double *d_out,
                             const double *d_tstar, void *stream);
/* the same with host arrays, synchronous, ACCUMULATING into cost / gradT[N] / gradC[18N] like isdf_eval (tstar: M doubles) */
int isdf_eval_swept_at_tstar_host(isdf_ctx *ctx, int N, const double *T, const double *coeffs, const double *tstar,
                                  double *cost_inout, double *gradT_inout, double *gradC_inout);

/* ---- full objective callback ------------------------------------------------------------------------------ */
/* TrajOptimizer::costFunctionLmbm (back_end_optimizer.hpp:358-430): x = [tau(N) | inner waypoints 3(N-1)] ->
 * cost, g.  MINCO (minco.hpp:397-655: setParameters, energy and its partials, propogateGrad) and the sweeps: for
 * ISDF_V1_SWEPT the swept-volume sweep followed by the integral sweep without a collision term (the reference's live
 * configuration), otherwise the integral sweep of cfg.variant.  cost = energy + sweeps + rho * sum(T).
 * MINCO has two forms here.  On the DEVICE (csrc/minco_dev.hip, minco_pcr.hpp): the trajectory in its junction states - a
 * symmetric positive definite block-tridiagonal system with 2 x 2 blocks - solved by parallel cyclic reduction in
 * ceil(log2(N - 1)) rounds, as two small kernels either side of the sweeps; the callback then moves n doubles down and n + 5
 * up and nothing else crosses PCIe.  On the HOST (csrc/minco_host.hpp): the reference's banded LU, a chain of 6N dependent
 * pivots, pivot for pivot.  They agree to rounding (coefficients 1e-10 relative, tests/test_minco_pcr.py).
 * isdf_set_minco_mode: 0 = whichever is faster for the configuration (default: the device, except single-trajectory tile-sweep
 * steps of <= 64 pieces on one GPU, where the host's 10 us beat two more kernel launches), 1 = host, 2 = device (trajectories of
 * more than 400 pieces always take the host).  isdf_minco_path tells which one the last callback took (1 = device).
 * isdf_set_trajectory == minco.setConditions + the `rho` of setParam (head/tail: 3x3 column-major, columns =
 * position, velocity, acceleration).  isdf_pack_variables == backwardT/backwardP (back_end_optimizer.cpp:22-28),
 * isdf_unpack_variables == forwardT/forwardP + setParameters (T: N, coeffs: 6N x 3 column-major; either may be NULL). */
int isdf_set_trajectory(isdf_ctx *ctx, int N, const double head_pva[9], const double tail_pva[9], double rho);
int isdf_num_variables(const isdf_ctx *ctx);               /* N + 3(N-1), 0 before isdf_set_trajectory     */
int isdf_pack_variables(isdf_ctx *ctx, const double *T, const double *waypoints, double *x);
int isdf_unpack_variables(isdf_ctx *ctx, const double *x, double *T, double *coeffs);
int isdf_cost_function(isdf_ctx *ctx, const double *x, double *g, int n, double *cost_out);
int isdf_set_minco_mode(isdf_ctx *ctx, int mode);          /* 0 = auto (default), 1 = host band LU, 2 = device */
int isdf_minco_path(const isdf_ctx *ctx);                  /* of the last callback: 1 = device, 0 = host      */
/* the same with the lmbm_evaluate_t signature (src/utils/include/utils/lmbm.h:206-209); instance = isdf_ctx*,
 * returns +inf on error */
double isdf_cost_function_lmbm(void *instance, const double *x, double *g, const int n);
/* Multi-GPU form (one process per GPU after isdf_set_shard): _launch runs MINCO and queues this rank's share of the
 * sweeps on `stream` (hipStream_t as void*), returning the device buffer of partial sums (count doubles); the caller
 * sums it over the ranks in place on the same stream (ONE all-reduce per step: ncclAllReduce(sum, ncclDouble) /
 * torch.distributed.all_reduce); _finish downloads it and completes the callback - identical (cost, g) on every rank. */
int isdf_cost_function_launch(isdf_ctx *ctx, const double *x, int n, void *stream, double **d_partial_out, size_t *count_out);
int isdf_cost_function_finish(isdf_ctx *ctx, double *g, double *cost_out, void *stream);
/* energy | swept-volume sweep | integral sweep | rho*sum(T) of the last isdf_cost_function call */
int isdf_cost_parts(const isdf_ctx *ctx, double parts[4]);

/* ---- multi-GPU exchange over xGMI peer stores (one node) -------------------------------------------------------- */
/* The exchange INSIDE the step (csrc/tile_sweep.hip, fused launch): on = 1 makes every following isdf_eval_device a complete
 * multi-GPU step - the workgroup that owns a piece stores its 19 output rows and its cost straight into every peer's board
 * (IPC-mapped, uncached), the peers' workgroups for that piece poll them into their own output, so on return every rank's
 * output holds the FULL [cost | gradT | gradC] (bitwise identical on all ranks) and no isdf_xchg_allreduce / ncclAllReduce
 * follows.  A step that runs as one fused launch (analytic shape with identity body offset, one trajectory or a small batch)
 * carries the exchange in that launch; any other step of the integral sweep (larger shards, body offsets, mesh robots) in its
 * tail launch.  Requires isdf_set_shard(rank, world) equal to the exchange's, all ranks switching together and evaluating in
 * lock-step.  Waits are bounded (about one second): a missing peer sets the overflow flag and a NaN cost instead of hanging
 * the stream - after which the exchange should be switched off (the boards may be out of step).
 * Replaces: the reference has no multi-GPU path; SURVEY 8(e) "prefer one-shot P2P reduce over xGMI". */
int isdf_xchg_fuse(isdf_ctx *ctx, int on);
/* The sum of the ranks' packed vectors is the ONE exchange step of the sharded path.  Any all-reduce works (RCCL through
 * torch.distributed: INTEGRATION.md); for vectors this small (6-50 KB) its latency rivals the whole optimizer step, so the
 * library also provides a one-shot peer-to-peer form: every rank stores its vector into a mailbox slot on every peer
 * (IPC-mapped, uncached device memory), raises a flag, waits (bounded) for its own mailbox and adds the slots in rank order -
 * one kernel per step, bit-identical sums on all ranks.  Setup: every rank calls isdf_xchg_create (64-byte IPC handle out),
 * the host all-gathers the handles (any transport), every rank calls isdf_xchg_connect with the world x 64 bytes in rank
 * order.  isdf_xchg_allreduce is asynchronous on `stream` and sums d_buf[0..count) in place; isdf_xchg_status returns 1 if
 * a wait ever timed out (the result of that exchange is invalid: fall back to RCCL).  All ranks must be on one node with
 * peer access between their devices. */
int isdf_xchg_create(isdf_ctx *ctx, int rank, int world_size, size_t max_doubles, void *ipc_handle_out_64bytes);
int isdf_xchg_connect(isdf_ctx *ctx, const void *ipc_handles_world_x_64bytes);
int isdf_xchg_allreduce(isdf_ctx *ctx, double *d_buf, size_t count, void *stream);
int isdf_xchg_status(isdf_ctx *ctx);
int isdf_xchg_destroy(isdf_ctx *ctx);
/* The bound T of every wait of both exchange forms, in MILLISECONDS OF THE DEVICE'S WALL CLOCK (wall_clock64(), constant rate:
 * the same duration on every box; default 2 000 - generous: the ranks are separate processes -, settable 1 ... 10 000): how much later than this rank a peer may start a step
 * before this rank calls the exchange failed.  A failed isdf_xchg_allreduce kernel ends after T on the rank whose wait expired
 * and at once on the late rank (which reads the first one's verdict); a healthy one waits at most 2.25 T for the verdicts.
 * Replaces: nothing in the reference (no multi-GPU path); the counterpart of NCCL's watchdog time-out, SURVEY 8(e). */
double isdf_xchg_timeout_ms(isdf_ctx *ctx);
int isdf_xchg_set_timeout_ms(isdf_ctx *ctx, double milliseconds);

/* ---- optimizer driver ----------------------------------------------------------------------------------------- */
/* L-BFGS behind the callback: lbfgs::lbfgs_optimize of src/utils/include/utils/lbfgs.hpp:480-835 (the LBFGS-Lite fork
 * the reference ships; its mid end calls it at src/planner_algorithm/src/mid_end.cpp:48-62).  Field names and defaults
 * are lbfgs_parameter_t's (:15-129).  weak_wolfe = 0 is the fork's Armijo-only line search (:373-386), 1 the
 * Lewis-Overton test it comments out; reference_patches = 1 keeps the fork's steepest-descent fallbacks (:788-819).
 * status uses the reference's codes: 0 convergence, 1 stop (delta test), 2 cancelled, negative = LBFGSERR_* (:133-160). */
typedef struct isdf_lbfgs_params {
    int32_t mem_size, past, max_iterations, max_linesearch, weak_wolfe, reference_patches;
    double g_epsilon, delta, min_step, max_step, f_dec_coeff, s_curv_coeff, cautious_factor, machine_prec, dir_norm_cap;
} isdf_lbfgs_params;
typedef struct isdf_lbfgs_result {
    double f;                /* cost at the returned x                              */
    double wall_ms;          /* host wall time of the whole minimisation            */
    int32_t status, iterations, evaluations, reserved;
} isdf_lbfgs_result;
typedef double (*isdf_evaluate_fn)(void *instance, const double *x, double *g, const int n);   /* == lmbm_evaluate_t */
void isdf_lbfgs_params_default(isdf_lbfgs_params *p);
/* any callback (no device needed) */
int isdf_lbfgs_minimize(isdf_evaluate_fn evaluate, void *instance, double *x_inout, int n,
                        const isdf_lbfgs_params *p, isdf_lbfgs_result *out);
/* Progress / cancel hook of the drivers: lbfgs_progress_t (lbfgs.hpp:256-262) with plain pointers - called once per iteration
 * after the line search with the iterate x, its gradient g (n doubles each), the cost, the accepted step, the iteration count k
 * and the number of evaluations ls of this line search; a non-zero return cancels the minimisation (status 2 = LBFGS_CANCELED,
 * x_inout = the iterate the hook was shown).  The reference's callers pass one to stop or watch a run (earlyExit /
 * earlyexitLmbm, back_end_optimizer.hpp:888-960); LMBM's lmbm_progress_t (lmbm.h:211-213) is its (instance, x, k) subset. */
typedef int (*isdf_progress_fn)(void *instance, const double *x, const double *g, double fx, double step, int k, int ls);
/* isdf_lbfgs_minimize with the hook; `instance` goes to both callbacks, like the reference's callback_data_t */
int isdf_lbfgs_minimize_progress(isdf_evaluate_fn evaluate, isdf_progress_fn progress, void *instance, double *x_inout, int n,
                                 const isdf_lbfgs_params *p, isdf_lbfgs_result *out);
/* The hook of the ctx's own drivers (NULL: none).  isdf_optimize_lbfgs calls progress(instance, ...); in
 * isdf_optimize_lbfgs_batch trajectory t's hook gets (char *)instance + t * batch_instance_stride (stride 0: the same pointer)
 * and is called on that trajectory's host thread - concurrently with the other trajectories' - so that one trajectory can be
 * cancelled (its result: status 2) while the rest of the batch runs on. */
int isdf_set_progress(isdf_ctx *ctx, isdf_progress_fn progress, void *instance, size_t batch_instance_stride);
/* the ctx's own objective: isdf_cost_function */
int isdf_optimize_lbfgs(isdf_ctx *ctx, double *x_inout, int n, const isdf_lbfgs_params *p, isdf_lbfgs_result *out);

/* A batch of trajectories optimised CONCURRENTLY on the shared map (BASELINE.json configs[2]): trajectory t has its own
 * boundary states heads_pva[9t..], tails_pva[9t..] (3x3 column-major each, like isdf_set_trajectory) and its own
 * variables x_inout[t*n .. (t+1)*n), n = N + 3(N-1); all share N, rho and the ctx's map / robot / weights.  Each runs
 * the same L-BFGS driver on its own callback (one host thread per trajectory); the callbacks of a round are evaluated as ONE
 * batched integral sweep on the device.  The trajectories are dealt into 4 fixed groups (ISDF_BATCH_GROUPS), a group's round
 * starts when all its live members wait - every round is full - and up to three rounds are in flight (ISDF_BATCH_SLOTS), so that
 * one group's sweep runs while the others' host threads do their L-BFGS updates: 128 x 40 pieces x 30 iterations in 75 ms = 4.9e8
 * point-evals/s end to end, 93 % of the batched sweep's own rate.  MINCO (minco.hpp:43-198,433-513,530-654) of a round runs on
 * the device, one workgroup per trajectory, either side of the sweep (csrc/minco_dev.hip); isdf_set_minco_mode(ctx, 1) keeps it on
 * the trajectories' host threads.  The iterates are bit for bit those of optimising every trajectory alone with
 * isdf_optimize_lbfgs UNDER THE SAME MINCO MODE (mode 0 picks the host form for a single trajectory of <= 64 pieces and the
 * device form for the batch: set the mode explicitly to compare).  results[t].reserved = number of device rounds of the whole
 * batch.  V2 / V3 contexts only; not on a sharded ctx (shard the batch across ranks instead). */
int isdf_optimize_lbfgs_batch(isdf_ctx *ctx, int n_traj, int N, const double *heads_pva, const double *tails_pva, double rho,
                              double *x_inout, const isdf_lbfgs_params *p, isdf_lbfgs_result *results, double *wall_ms_out);

/* ---- front end: pose feasibility by kernel convolution ------------------------------------- */
/* The A* front end decides whether the robot fits at a voxel by AND-ing a bit-packed voxelisation of the robot at a
 * (roll, pitch) attitude against the bit-packed occupancy map, trying attitudes breadth-first from the parent's
 * (SweptVolumeManager::checkKernelValue, sw_manager.hpp:911-942).  The tables are built on the device from the installed
 * shape and occupancy grid; queries are answered in batches.  All results are integer / byte work: bit-identical to the
 * reference's. */
typedef struct isdf_frontend_config {
    int32_t kernel_size;        /* Config::kernel_size: odd, side of the robot voxelisation in voxels (<= 31)            */
    int32_t reserved;
    double kernel_max_roll;     /* degrees: attitudes -max .. +max in steps of kernel_ang_res (Shape.hpp:297-298,          */
    double kernel_max_pitch;    /*          :416-421); the shipped configs use 45 / 45 / 9 -> 11 x 11 attitudes            */
    double kernel_ang_res;
    double front_end_safeh;     /* a voxel belongs to the robot when sdf <= max(front_end_safeh, resolution / 2) (:415)    */
} isdf_frontend_config;
/* Builds (a) the robot's attitude kernels - BasicShape::initShape<true,...> kernel part (Shape.hpp:400-459): for every
 * attitude, voxel (a, b, c) is set when getonlySDF(pos, Rx(roll) * Ry(pitch)) <= margin - and (b) the inflated, bit-packed
 * occupancy map of PCSmapManager::generateMapKernel (PCSmap_manager.h:46-78).  Needs isdf_set_shape and an occupancy grid
 * (isdf_set_grid ISDF_GRID_OCCUPANCY or isdf_set_pointcloud).  Call again after either changes. */
int isdf_frontend_build(isdf_ctx *ctx, const isdf_frontend_config *cfg);
/* Read-back in the REFERENCE's byte layouts (bit z of a row in byte z / 8 under mask 0x80 >> (z % 8)).
 * Shape kernels: ByteShapeKernel::map of every attitude, attitude-major ((i * ykernel + j) * k * k * ((k + 7) / 8) bytes;
 * what plan_manager.cpp:545 reads through getOccupied).  dims_out = {xkernel_size, ykernel_size, bytes per attitude}.
 * Map kernel: the array SweptVolumeManager::setMapKernel receives; dims_out = inflated {X, Y, bytes per z-row}.
 * `out` may be NULL to query the sizes. */
int isdf_frontend_get_shape_kernels(isdf_ctx *ctx, uint8_t *out, int dims_out[3]);
int isdf_frontend_get_map_kernel(isdf_ctx *ctx, uint8_t *out, int dims_out[3]);
/* n queries of the per-neighbour test of AstarPathSearcher::AstarGetSucc (front_end_Astar.hpp:214-217):
 *   ok[q] = isIndexValid(ind) && !isIndexOccupiedFlate(ind, 0) && checkKernelValue(father_roll, father_pitch, cr, cp, ind)
 * with ind = index[3q..3q+2] (voxel of the ORIGINAL map), and child_roll / child_pitch (degrees) of the first collision-free
 * attitude in the reference's breadth-first order (visit_kernels_by_distance, :850-909: level attitude first, then outward
 * from the parent's attitude, neighbours pushed in the order (0,+1) (0,-1) (+1,0) (-1,0)).  child_* are written only where
 * ok[q] = 1.  kernel_index_out (optional) receives i * ykernel + j of that attitude, -1 where ok = 0. */
int isdf_frontend_check(isdf_ctx *ctx, int n, const int32_t *index, const double *father_roll, const double *father_pitch,
                        uint8_t *ok, double *child_roll, double *child_pitch, int32_t *kernel_index_out);

/* The whole configuration space in one pass: for EVERY voxel of the map, which attitudes are collision-free
 * (kernelConv<true>(i, j, voxel), sw_manager.hpp:813-847, for all i, j).  free_mask_out (may be NULL: the table then only
 * stays on the device): 4 * ceil(attitudes / 128) dwords per voxel in the grid's own order (z fastest; 4 for the shipped 11 x 11),
 * bit (i * ykernel + j) set = that attitude fits; occupied voxels get 0.  An A* that holds this table answers checkKernelValue with a few
 * bit tests in the breadth-first order instead of k^2 byte-ANDs per attitude.  kernel_ms_out (optional): device time. */
int isdf_frontend_cspace(isdf_ctx *ctx, uint32_t *free_mask_out, double *kernel_ms_out);

/* The SE(3) A* of the front end (AstarPathSearcher, planner_algorithm/front_end_Astar.hpp:172-403) over the table above, called
 * like PlannerManager::generatePath calls the reference's (plan_manager.cpp:181-198):
 *   isdf_frontend_astar_search  = AstarPathSearch(start, end); result->success = success_flag;
 *   isdf_frontend_astar_path    = getPath() + getastarSE3Path() of that search (valid until the next search, which is also the
 *                                 reference's reset()).
 * The first search after isdf_frontend_build computes the whole configuration space on the device (isdf_frontend_cspace) and
 * keeps it in pinned host memory; a search then runs on the calling thread and answers every neighbour test
 * (isIndexValid && !isIndexOccupiedFlate && checkKernelValue, :214-216) with bit tests on a voxel's word of the table (128 bits per 128 attitudes) in the
 * reference's breadth-first attitude order.  Same open-set order (that of a multimap keyed by the fScore at insertion, first inserted
 * first among equals), same 26+1 neighbour order, same re-opening of closed nodes, same (roll, pitch) bookkeeping (a node's
 * attitude is overwritten by every expansion that finds it feasible, :227-228) => the same path and attitudes, node for node.
 * start / goal: world coordinates; outside the map => success = 0 (the reference logs an error and returns, :244-249). */
typedef struct isdf_astar_result {
    int32_t success;            /* AstarPathSearcher::success_flag                                                         */
    int32_t n_path;             /* nodes on the path, start and goal cells included (0 when the search failed)             */
    int64_t expansions;         /* nodes taken off the open set                                                            */
    int64_t checks;             /* neighbour tests = the reference's total_kernel (:218)                                   */
    double cspace_ms;           /* device time of the configuration-space pass when THIS call ran it, else 0               */
    double table_ms;            /* wall time of that pass + bringing the table to the host, else 0                         */
    double search_ms;           /* wall time of the search itself                                                          */
} isdf_astar_result;
int isdf_frontend_astar_search(isdf_ctx *ctx, const double start[3], const double goal[3], isdf_astar_result *result);
/* Path of the last successful search, start -> goal: xyz = cube centres (3 doubles per node), roll_pitch = degrees (2 per
 * node), rot = SE3State::rot = AngleAxis(roll, X) * AngleAxis(pitch, Y) as a row-major 3x3 (9 per node); any may be NULL.
 * Writes at most `capacity` nodes; returns the number of nodes of the path (0 = no path), or a negative isdf_status. */
int isdf_frontend_astar_path(isdf_ctx *ctx, int capacity, double *xyz, double *roll_pitch, double *rot);

/* ---- the reference's own input files (host side; no device needed) ---------------------------------------------------- */
/* ASCII .pcd global map as pcl::io::loadPCDFile<pcl::PointXYZ> reads it (src/map_manager/src/globalmap_gene.cpp:433-460;
 * the shipped src/plan_manager/map_pcds are "FIELDS x y z / DATA ascii"): xyz_out = up to `capacity` points x 3 floats (may be
 * NULL to count).  Returns the number of points in the file, or a negative isdf_status (ISDF_ERR_UNSUPPORTED: binary data).
 * The points go to isdf_set_pointcloud (= PCSmapManager::rcvGlobalMapHandler, PCSmap_manager.cpp:87-200). */
long long isdf_read_pcd(const char *path, float *xyz_out, long long capacity);
/* Wavefront .obj as igl::read_triangle_mesh reads it (src/utils/src/Shape.cpp:36): vertices V_out (capV x 3), triangles F_out
 * (capF x 3, zero based; polygons fanned from their first vertex).  nV_out / nF_out always receive the counts in the file. */
int isdf_read_obj(const char *path, double *V_out, int capV, int32_t *F_out, int capF, int *nV_out, int *nF_out);
/* yaml poly_params [x, y, z, roll, pitch, yaw (degrees)] -> Rotate = yaw * pitch * roll with Eigen's AngleAxis matrices and
 * PI = 3.14159265358979323846 (Shape.cpp:23,38-43), row-major. */
int isdf_poly_rotation(const double poly_params[6], double rotate_out[9]);
/* Generalshape's constructor transform (Shape.cpp:37-49): V <- (V.homogeneous() * Trans^T).hnormalized() in place, with
 * Trans = [Rotate | trans]; trans_out / rotate_out (nullable) receive what isdf_shape.trans / .rotate take. */
int isdf_body_transform(const double poly_params[6], double *V_inout, int nV, double trans_out[3], double rotate_out[9]);
/* What a plan needs from the yaml files of src/plan_manager/config (rosparam -> Config::loadParameters,
 * src/utils/include/utils/config.hpp:13-203).  Fields a file leaves out keep Config's in-class defaults. */
typedef struct isdf_plan_config {
    isdf_config sweep;             /* the hot path's fields; variant = ISDF_V1_SWEPT (the live configuration)                  */
    isdf_frontend_config frontend; /* kernel_size, kernel_max_roll / _pitch, kernel_ang_res, front_end_safeh                   */
    double occupancy_resolution;   /* voxel edge of the occupancy grid                                                        */
    int32_t sta_threshold;         /* points per voxel for "occupied" (PCSmap_manager.cpp:160)                                */
    int32_t threads_num;           /* the reference's OpenMP team (baseline only)                                             */
    double rho, inittime, momentum;
    double traj_parlength;         /* 3.0: waypoint spacing in metres (plan_manager.cpp:153,206-213)                          */
    double poly_params[6];         /* body offset of the robot: x y z roll pitch yaw(deg)                                     */
    double offset_aabb[3];         /* offsetAABBbox                                                                           */
    double box[3];                 /* box_x / box_y / box_z                                                                   */
    double map_bound[6];           /* mapBound                                                                                */
    char inputdata[256];           /* e.g. "shapes/RoundedCone.obj" (relative to the plan_manager package)                    */
    char pcdmapname[128];
} isdf_plan_config;
void isdf_plan_config_default(isdf_plan_config *out);
int isdf_load_yaml_config(const char *yaml_path, isdf_plan_config *out);
/* SweptVolumeManager::initShape (sw_manager.hpp:239-275): the stem of inputdata is looked up in the analytic registry
 * (isdf_shape_from_name; trans / rotate from poly_params); any other stem is the mesh Generalshape - the obj is read from
 * package_dir/inputdata into the caller's buffers and put through isdf_body_transform.  ISDF_ERR_OVERFLOW: buffers too small. */
int isdf_shape_from_config(isdf_shape *shape, const isdf_plan_config *cfg, const char *package_dir, double *V_buf, int capV,
                           int32_t *F_buf, int capF);

/* ---- instrumentation --------------------------------------------------------------------------------------- */
/* on = N > 0: every N-th isdf_eval_device attaches HIP start/stop events to the dispatch of its dominant kernel on
 * `stream` (hipExtLaunchKernel: the kernel's own begin/end timestamps, the interval rocprofv3 reports);
 * isdf_profile_read synchronises and returns the number of launches recorded since the last read and their
 * mean duration in milliseconds.  on = 0 disables.  on = N | ISDF_PROFILE_SECONDARY also times the kernel that follows
 * the dominant one (two more events per instrumented launch; each timed dispatch costs the stream a few microseconds). */
#define ISDF_PROFILE_SECONDARY 0x10000
int isdf_profile_enable(isdf_ctx *ctx, int on);
int isdf_profile_read(isdf_ctx *ctx, int *n_launches, double *mean_ms);
/* Mean duration (ms) of the kernel that follows the dominant one in the same launches (tail_kernel / the V1 reduce),
 * valid after isdf_profile_read when ISDF_PROFILE_SECONDARY was set (0 otherwise). */
int isdf_profile_read_secondary(isdf_ctx *ctx, double *mean_ms);
/* Counters of the last evaluation (for tests / roofline bookkeeping). */
typedef struct isdf_stats {
    int64_t n_units;          /* constraint-point evaluations (poses for V2/V3, obstacle points for V1)  */
    int64_t n_units_culled;   /* V3 poses skipped by the whole-tile cull                                  */
    int64_t n_pairs;          /* (pose, qualifying voxel) robot-SDF evaluations                           */
    int64_t n_grad_pairs;     /* pairs whose penalty was active (gradient evaluated)                      */
    int32_t overflow;         /* nonzero: a bounded list overflowed                                       */
    int32_t reserved;
} isdf_stats;
int isdf_get_stats(isdf_ctx *ctx, isdf_stats *out);

#ifdef __cplusplus
}
#endif
#endif /* ISDF_ACCEL_H */
