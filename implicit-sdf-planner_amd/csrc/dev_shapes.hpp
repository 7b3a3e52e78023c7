// Device-side robot-shape plugin: analytic SDF kinds + body-frame gradient (gfx950).
// The host Shape registry (sw_manager.hpp:74-123) maps to DevShape.kind; the class constants of
// src/utils/include/utils/Shape.hpp:824-2391 travel in ShapeP (slot meaning: include/isdf_accel.h).
// KIND is a template parameter so the hot kernels inline exactly one shape; KIND = -1 dispatches at run time
// (kind is wave-uniform, so the switch does not diverge).  The scalar type T is double on the exact path; the
// float instantiation is ONLY used by tile_kernel's conservative pre-filter (a voxel the filter rejects provably has
// zero penalty; everything it keeps is re-evaluated in fp64).
#pragma once
#include "dev_math.hpp"
#include "../../include/isdf_accel.h"

namespace isdf {

// mesh kind (functions in dev_mesh.hpp)
constexpr int MESH_STACK = 64;             // per-lane traversal stack: >= 3 * depth + 1, checked by isdf_set_shape
constexpr int MESH_FRAMES = (MESH_STACK - 1) / 3;      // frames of the winding-number walk: one per level of the hierarchy
constexpr int MESH_Q_LEVELS = 12;          // quad-cooperative walks (dev_mesh.hpp): levels of frames kept in LDS; deeper hierarchies take the wave-cooperative walks
constexpr int MESH_Q_REC = 40;             // floats per (node, child) in DevMesh::fwn_boxq
constexpr int MESH_Q_TRI = 10;             // doubles per (node, child) in DevMesh::fwn_triq
constexpr float MESH_FWN_ACCURACY = 2.0f;  // accuracy_scale of igl::fast_winding_number(fwn_bvh, 2.0, ...), Shape.cpp:110
struct DevMesh {
    const double *tri;   // nF x 9: a.xyz b.xyz c.xyz
    const float *trif;   // nF x 9 (float copies for the winding number)
    int nF;
    const int *fwn_child;   // 4 per node: triangle index | -1 empty | node number with the top bit set (csrc/fwn_host.hpp)
    const float *fwn_box;   // 92 per node: per-child radius^2, centre and order-2 expansion coefficients
    int fwn_nodes;
    // for the quad-cooperative walks (dev_mesh.hpp), child-major so that ONE round of loads gives a lane all it needs of a node:
    const float *fwn_boxq;  // 40 per (node, child): [k] = fwn_box[92 * node + 4 * k + child], k = 0..22; [23] = the child word
                            // (fwn_child) as bits; [24..32] = the child's triangle (trif) when it is one; pad
    const double *fwn_triq; // 10 per (node, child): the child's triangle (tri) when it is one; pad
    int depth;              // levels of the hierarchy
    // The tile sweep's pre-filter for this kind (built on the device by isdf_set_shape, shape_eval.hip): a lattice of node-to-surface
    // DISTANCES, rounded down, with the sign bit set where the node lies inside the mesh (reference SDF <= 0); node (i, j, k) at
    // dl_min + (i, j, k) / dl_inv, value dl[(i * dln[1] + j) * dln[2] + k]; dl_h = the farthest a point can be from its nearest
    // node.  A point p whose nearest node n has dl[n] - dl_h >= band is farther than `band` from the surface and on the node's
    // side of it (the segment n -> p cannot cross a surface that is farther from n than p is).  null: no lattice.
    const float *dl;
    int dln[3];
    float dl_min[3], dl_inv, dl_h;
    // defects (shape_eval.hip, mesh_validity_kernel): within dl_tau of the surface a pocket may exist in which |1 - 2 w| is not ~1;
    // a point in one has |sdf| <= dl_slack (= 1.05 x the largest |1 - 2 w| seen in a pocket x dl_tau).  0, 0 for a clean mesh.
    float dl_tau, dl_slack;
    // Small meshes (the reference's own robots: 12-52 faces), FLAT evaluation (dev_mesh.hpp, mesh_sdf_grad_below_flat): the whole hierarchy
    // as a table of its (node, child) SLOTS in level order, so that the four lanes of a query sweep a level's slots side by side instead
    // of walking node by node.  flat: a self-describing blob (isdf_host.hip build_blob: a header of 24 ints, per slot 4 ints, per triangle
    // its slot, per node 5 ints, then the slots' records and fp64 triangles).  null: the mesh does not qualify (more than MESH_FLAT_SLOTS slots).
    // Behind the tables, in the same allocation: the slots' records (flat_rec, in 4-byte units from `flat`: MESH_Q_REC floats per slot, the
    // fwn_boxq record of the slot's (node, child)) and their fp64 triangles (flat_trec: MESH_Q_TRI doubles per slot).  The swept-volume
    // kernels copy the whole blob (flat_words 4-byte words, a multiple of 4) into LDS once per workgroup.
    const int *flat;
    int flat_slots, flat_nodes, flat_levels;
    int flat_words, flat_rec, flat_trec;
};
constexpr int MESH_FLAT_SLOTS = 64;        // (node, child) slots a flat-evaluated mesh may have: one bit each in a query's 64-bit "opened" mask
constexpr int MESH_FLAT_LEVELS = 8;

template <typename T> struct ShapeP {
    T p[16];
    T trans[3];
    T rot[9];
    T csg_r2[9], csg_r3[9];   // rotate_to(c, X, Y) and rotate_to(c, X, Z) matrices, built on the host
    int ident;                // 1: trans == 0 and rot == I exactly (poly_params all zero, the shipped configs): the body offset is a no-op
};

struct DevShape {
    int kind, grad_mode;
    int filter_f32;           // 1: the fp32 pre-filter may be used for this shape (continuous analytic SDF)
    int prune_rows;           // 1: bbox_lo/hi valid -> the scan may prune voxel rows outside the inflated box
    float bbox_lo[3], bbox_hi[3];
    ShapeP<double> d;
    ShapeP<float> f;
    double bound_radius;
    const DevMesh *mesh;
    int mesh_wg;              // 1: the swept-volume kernels give this mesh one task per WORKGROUP (quad-cooperative walks)
    int mesh_levels;          // ... and the levels of walk frames a wavefront needs for it (the hierarchy's depth)
    int mesh_flat;            // 1: ... as a FLAT evaluation instead (small meshes: DevMesh::flat)
    int mesh_flat_words, mesh_flat_slots;      // ... the size of its table blob (4-byte words) and its slot count (launch-time LDS sizing)
    // ISDF_SHAPE_GRID: num_sdf_map of BasicShape (Shape.hpp:376-404): 4 doubles per node (unit gradient, distance)
    const double *grid;
    int gn[3];
    double gres, gmin[3];
};

template <typename T> __device__ __forceinline__ T clipT(T v, T lo, T hi) { return m_max(m_min(v, hi), lo); }
template <typename T> __device__ __forceinline__ T box_q(T qx, T qy, T qz) {
    const T mx = m_max(qx, T(0)), my = m_max(qy, T(0)), mz = m_max(qz, T(0));
    return m_sqrt(mx * mx + my * my + mz * mz) + m_min(m_max(qx, m_max(qy, qz)), T(0));
}

template <typename T> __device__ __forceinline__ T sdf_torus(const T *P, v3<T> p) {
    const T qx = m_sqrt(p.x * p.x + p.z * p.z) - P[0];
    return m_sqrt(qx * qx + p.y * p.y) - P[1];
}
template <typename T> __device__ __forceinline__ T sdf_cappedtorus(const T *P, v3<T> p) {
    const T scx = P[0], scy = P[1], ra = P[2], rb = P[3];
    p.x = m_abs(p.x);
    const T k = (scy * p.x > scx * p.y) ? (p.x * scx + p.y * scy) : m_sqrt(p.x * p.x + p.y * p.y);
    return m_sqrt(dot3(p, p) + ra * ra - T(2) * ra * k) - rb;
}
template <typename T> __device__ __forceinline__ T sdf_cappedcone(const T *P, v3<T> p) {
    const T ra = P[0], rb = P[1];
    const v3<T> a = mkv<T>(P[2], P[3], P[4]), b = mkv<T>(P[5], P[6], P[7]);
    const T rba = rb - ra;
    const v3<T> ba = b - a, pa = p - a;
    const T baba = dot3(ba, ba);
    const T papa = dot3(pa, pa);
    const T paba = dot3(pa, ba) / baba;
    const T x = m_sqrt(papa - paba * paba * baba);
    const T cax = m_max(T(0), x - (paba < T(0.5) ? ra : rb));
    const T cay = m_abs(paba - T(0.5)) - T(0.5);
    const T k = rba * rba + baba;
    const T f = clipT((rba * (x - ra) + paba * baba) / k, T(0), T(1));
    const T cbx = x - ra - f * rba;
    const T cby = paba - f;
    const T s = (cbx < 0 && cay < 0) ? T(-1) : T(1);
    const T d = m_sqrt(m_min(cax * cax + cay * cay * baba, cbx * cbx + cby * cby * baba));
    return s * m_sqrt(m_abs(d)) / m_abs(baba);
}
template <typename T> __device__ __forceinline__ T sdf_roundedcone(const T *P, v3<T> p) {
    const T r1 = P[0], r2 = P[1], h = P[2];
    const T qx = m_sqrt(p.x * p.x + p.y * p.y), qy = p.z;
    const T b = (r1 - r2) / h;
    const T a = m_sqrt(T(1) - b * b);
    const T k = -b * qx + a * qy;
    // c1 = |q| - r1 below the cone's base, c2 = |q - (0, h)| - r2 beyond its tip, c3 on the flank: the two square roots are
    // never both needed, so the ARGUMENT is selected and one root is taken (the value returned is the same bit for bit)
    const bool low = k < 0, high = k > a * h;
    const T dy = low ? qy : qy - h;
    const T c12 = m_sqrt(qx * qx + dy * dy) - (low ? r1 : r2);
    const T c3 = (a * qx + b * qy) - r1;
    return (low || high) ? c12 : c3;
}
template <typename T> __device__ __forceinline__ T sdf_wireframebox(const T *P, v3<T> p) {
    const T th = P[3];
    const T px = m_abs(p.x) - P[0] / 2 - th / 2, py = m_abs(p.y) - P[1] / 2 - th / 2, pz = m_abs(p.z) - P[2] / 2 - th / 2;
    const T qx = m_abs(px + th / 2) - th / 2, qy = m_abs(py + th / 2) - th / 2, qz = m_abs(pz + th / 2) - th / 2;
    return m_min(m_min(box_q(px, qy, qz), box_q(qx, py, qz)), box_q(qx, qy, pz));
}
template <typename T> __device__ __forceinline__ T sdf_bendlinear(const T *P, v3<T> p) {
    const T L = P[0], radius = P[1];
    // p0 = (0,0,-1), p1 = (0,0,1), v = (-1,0,0), ab = (0,0,2)
    const T t = clipT(((p.x - T(0)) * T(0) + (p.y - T(0)) * T(0) + (p.z + T(1)) * T(2)) / T(4), T(0), T(1));
    const T u = 2 * t - 1;
    const T e = (t < T(0.5)) ? (2 * t * t) : (T(-0.5) * (u * (u - 2) - 1));
    const v3<T> pt = mkv<T>(e * T(-1) + p.x, e * T(0) + p.y, e * T(0) + p.z);
    const v3<T> pa = mkv<T>(pt.x, pt.y, pt.z + L);
    const v3<T> ba = mkv<T>(T(0), T(0), 2 * L);
    const T h = m_max(m_min(dot3(pa, ba) / dot3(ba, ba), T(1)), T(0));
    return norm3(pa - h * ba) - radius;
}
template <typename T> __device__ __forceinline__ T sdf_twistbend(const T *P, v3<T> p, bool bend) {
    const T ang = bend ? P[3] * p.x : P[3] * p.z;
    T s, c;
    m_sincos(ang, s, c);
    const T rx = c * p.x - s * p.y, ry = s * p.x + c * p.y;
    return box_q(m_abs(rx) - P[0] / T(2), m_abs(ry) - P[1] / T(2), m_abs(p.z) - P[2] / T(2));
}
template <typename T> __device__ __forceinline__ T sdf_table(const T *P, v3<T> p) {
    const T x = m_abs(p.x), y = m_abs(p.y), z = p.z;
    const T f1 = box_q(m_abs(x - (P[0] + P[3]) / 2) - (P[3] - P[0]) / T(2), m_abs(y - (P[1] + P[4]) / 2) - (P[4] - P[1]) / T(2),
                       m_abs(z - (P[2] + P[5]) / 2) - (P[5] - P[2]) / T(2));
    const T f2 = box_q(m_abs(x - (P[6] + P[9]) / 2) - (P[9] - P[6]) / T(2), m_abs(y - (P[7] + P[10]) / 2) - (P[10] - P[7]) / T(2),
                       m_abs(z - (P[8] + P[11]) / 2) - (P[11] - P[8]) / T(2));
    return m_min(f1, f2);
}
template <typename T> __device__ __forceinline__ void rot2d(T &qx, T &qy, T a) {
    T sa, ca;
    m_sincos(a, sa, ca);
    const T nx = qx * ca + qy * sa, ny = qy * ca + (-qx) * sa;
    qx = nx; qy = ny;
}
template <typename T> __device__ __forceinline__ T sdf_trefoil(const T *P, v3<T> p) {
    const T PI = T(3.14159265358979323846);
    T qx = p.x, qy = p.y;
    const T a = m_atan2(qy, qx);
    qx = m_sqrt(qx * qx + qy * qy) - P[0];
    qy = -p.z;
    rot2d(qx, qy, T(1.5) * a);
    rot2d(qx, qy, -PI * (m_floor(m_atan2(qy, qx) / PI + T(0.5))));
    qx -= T(1);
    const T dx = m_abs(qx) - P[1], dy = m_abs(qy) - P[2];
    const T mx = m_max(dx, T(0)), my = m_max(dy, T(0));
    const T d = (m_min(m_max(dx, dy), T(0)) + m_sqrt(mx * mx + my * my)) - P[3];
    return P[4] * m_min(d, T(100));
}
template <typename T> __device__ __forceinline__ T sdf_smooth(const T *P, v3<T> p, bool intersection) {
    const T box_sdf = box_q(m_abs(p.x) - P[0] / T(2), m_abs(p.y) - P[1] / T(2), m_abs(p.z) - P[2] / T(2));
    const T sphere_sdf = norm3(p) - P[3];
    const T kk = P[4];
    if (!intersection) {
        const T h = clipT(T(0.5) - T(0.5) * (box_sdf + sphere_sdf) / kk, T(0), T(1));
        return (box_sdf - (box_sdf + sphere_sdf) * h) + kk * h * (T(1) - h);
    }
    const T h = clipT(T(0.5) - T(0.5) * (sphere_sdf - box_sdf) / kk, T(0), T(1));
    return (sphere_sdf + (box_sdf - sphere_sdf) * h) + kk * h * (T(1) - h);
}
template <typename T> __device__ __forceinline__ T sdf_csg(const ShapeP<T> &S, v3<T> p) {
    const T *P = S.p;
    const T f = m_max(norm3(p) - P[0], box_q(m_abs(p.x) - P[1] / T(2), m_abs(p.y) - P[1] / T(2), m_abs(p.z) - P[1] / T(2)));
    const T c1 = m_sqrt(p.x * p.x + p.y * p.y) - P[2];
    const T *A = S.csg_r2, *B = S.csg_r3;
    const T ax = A[0] * p.x + A[1] * p.y + A[2] * p.z, ay = A[3] * p.x + A[4] * p.y + A[5] * p.z;
    const T bx = B[0] * p.x + B[1] * p.y + B[2] * p.z, by = B[3] * p.x + B[4] * p.y + B[5] * p.z;
    const T c2 = m_sqrt(ax * ax + ay * ay) - P[2];
    const T c3 = m_sqrt(bx * bx + by * by) - P[2];
    return m_max(f, -m_min(m_min(c1, c2), c3));
}

// the class formula at a point already in the shape's own frame (after the body offset); analytic kinds except Ball
template <int KIND, typename T>
__device__ __forceinline__ T analytic_local(const ShapeP<T> &S, int kind_rt, v3<T> p) {
    const int kind = (KIND >= 0) ? KIND : kind_rt;
    switch (kind) {
    case ISDF_SHAPE_TORUS: return sdf_torus(S.p, p);
    case ISDF_SHAPE_CAPPEDTORUS: return sdf_cappedtorus(S.p, p);
    case ISDF_SHAPE_CAPPEDCONE: return sdf_cappedcone(S.p, p);
    case ISDF_SHAPE_ROUNDEDCONE: return sdf_roundedcone(S.p, p);
    case ISDF_SHAPE_WIREFRAMEBOX: return sdf_wireframebox(S.p, p);
    case ISDF_SHAPE_BENDLINEAR: return sdf_bendlinear(S.p, p);
    case ISDF_SHAPE_TWISTBOX: return sdf_twistbend(S.p, p, false);
    case ISDF_SHAPE_BENDBOX: return sdf_twistbend(S.p, p, true);
    case ISDF_SHAPE_TABLE: return sdf_table(S.p, p);
    case ISDF_SHAPE_TREFOIL: return sdf_trefoil(S.p, p);
    case ISDF_SHAPE_SMOOTHDIFFERENCE: return sdf_smooth(S.p, p, false);
    case ISDF_SHAPE_SMOOTHINTERSECTION: return sdf_smooth(S.p, p, true);
    case ISDF_SHAPE_CSG: return sdf_csg(S, p);
    case ISDF_SHAPE_BOX: return box_q(m_abs(p.x) - S.p[0], m_abs(p.y) - S.p[1], m_abs(p.z) - S.p[2]);
    default: return T(1e9);
    }
}
// (pos_rel - trans) * Rotate: the body offset every analytic class applies first (e.g. Shape.hpp:843)
// IDENT = true: the caller (a kernel instantiation chosen on the host) KNOWS the offset is the identity, so the general path
// and the 12 scalars it keeps alive are not even compiled in.
template <bool IDENT = false, typename T> __device__ __forceinline__ v3<T> body_offset(const ShapeP<T> &S, v3<T> pr) {
    if constexpr (IDENT) return pr;
    if (S.ident) return pr;               // wave-uniform; (p - 0) * I == p bit for bit (finite p)
    const T dx = pr.x - S.trans[0], dy = pr.y - S.trans[1], dz = pr.z - S.trans[2];
    const T *R = S.rot;
    return mkv<T>(dx * R[0] + dy * R[3] + dz * R[6], dx * R[1] + dy * R[4] + dz * R[7], dx * R[2] + dy * R[5] + dz * R[8]);
}

// analytic kinds only (kind != MESH)
template <int KIND, typename T, bool IDENT = false>
__device__ __forceinline__ T analytic_sdf(const ShapeP<T> &S, int kind_rt, v3<T> pr) {
    const int kind = (KIND >= 0) ? KIND : kind_rt;
    if (kind == ISDF_SHAPE_BALL) return norm3(pr) - S.p[0];
    return analytic_local<KIND, T>(S, kind_rt, body_offset<IDENT>(S, pr));
}

// ---- ISDF_SHAPE_GRID: getonlySDFNum / getonlyGrad1Num / getSDFwithGrad1Num (Shape.hpp:481-600), operation for operation
struct GridCellRef { int base[8]; double xd, yd, zd; bool inside; };
__device__ __forceinline__ GridCellRef grid_locate(const DevShape &S, d3 p) {
#pragma clang fp contract(off)
    GridCellRef c;
    const double fx = (p.x - S.gmin[0]) / S.gres, fy = (p.y - S.gmin[1]) / S.gres, fz = (p.z - S.gmin[2]) / S.gres;
    const int i = (int)floor(fx), j = (int)floor(fy), k = (int)floor(fz);
    const int X = S.gn[0], Y = S.gn[1], Z = S.gn[2];
    c.inside = i >= 0 && i < X - 1 && j >= 0 && j < Y - 1 && k >= 0 && k < Z - 1;
    c.xd = fx - i; c.yd = fy - j; c.zd = fz - k;
    const int ijk = (i * Y + j) * Z + k;
    c.base[0] = ijk;              c.base[1] = ijk + Y * Z;          // ijk, i1_jk
    c.base[2] = ijk + Z;          c.base[3] = ijk + Y * Z + Z;      // ij1_k, i1_j1_k
    c.base[4] = ijk + 1;          c.base[5] = ijk + Y * Z + 1;      // ij_k1, i1_j_k1
    c.base[6] = ijk + Z + 1;      c.base[7] = ijk + Y * Z + Z + 1;  // ij1_k1, i1_j1_k1
    return c;
}
__device__ __forceinline__ double grid_value(const DevShape &S, const GridCellRef &c) {
#pragma clang fp contract(off)      // (the reference's products and sums, not fused: same bits as its getonlySDFNum)
    const double *G = S.grid;
    const double xd = c.xd, yd = c.yd, zd = c.zd;
    const double c00 = G[4 * (size_t)c.base[0] + 3] * (1 - xd) + G[4 * (size_t)c.base[1] + 3] * xd;
    const double c10 = G[4 * (size_t)c.base[2] + 3] * (1 - xd) + G[4 * (size_t)c.base[3] + 3] * xd;
    const double c01 = G[4 * (size_t)c.base[4] + 3] * (1 - xd) + G[4 * (size_t)c.base[5] + 3] * xd;
    const double c11 = G[4 * (size_t)c.base[6] + 3] * (1 - xd) + G[4 * (size_t)c.base[7] + 3] * xd;
    const double c0 = c00 * (1 - yd) + c10 * yd;
    const double c1 = c01 * (1 - yd) + c11 * yd;
    return c0 * (1 - zd) + c1 * zd;
}
__device__ __forceinline__ d3 grid_gradient(const DevShape &S, const GridCellRef &c) {
#pragma clang fp contract(off)
    const double *G = S.grid;
    const double xd = c.xd, yd = c.yd, zd = c.zd;
    // g000 * (1-xd) * (1-yd) * (1-zd) + g100 * xd * (1-yd) * (1-zd) + ...: Eigen multiplies a vector by its scalars from the left,
    // component by component - ((g a) b) c - and adds the eight terms from the left
    const double fa[8] = {1 - xd, xd, 1 - xd, xd, 1 - xd, xd, 1 - xd, xd};
    const double fb[8] = {1 - yd, 1 - yd, yd, yd, 1 - yd, 1 - yd, yd, yd};
    const double fc[8] = {1 - zd, 1 - zd, 1 - zd, 1 - zd, zd, zd, zd, zd};
    d3 g = mk3(0, 0, 0);
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const double *n = G + 4 * (size_t)c.base[q];
        const d3 t = mk3(((n[0] * fa[q]) * fb[q]) * fc[q], ((n[1] * fa[q]) * fb[q]) * fc[q], ((n[2] * fa[q]) * fb[q]) * fc[q]);
        g = q == 0 ? t : mk3(g.x + t.x, g.y + t.y, g.z + t.z);
    }
    return normalized3(g);
}
__device__ __forceinline__ double grid_sdf(const DevShape &S, d3 p) {
    const GridCellRef c = grid_locate(S, p);
    return c.inside ? grid_value(S, c) : 1e20;
}
__device__ __forceinline__ d3 grid_grad(const DevShape &S, d3 p) {
    const GridCellRef c = grid_locate(S, p);
    return c.inside ? grid_gradient(S, c) : mk3(0, 0, 0);
}

__device__ double mesh_sdf(const DevMesh *m, d3 p);              // dev_mesh.hpp
__device__ double mesh_sdf_grad(const DevMesh *m, d3 p, d3 &g);  // dev_mesh.hpp

// BasicShape::getonlySDF(pos_rel), exact path (fp64)
template <int KIND, bool IDENT = false>
__device__ __forceinline__ double shape_sdf(const DevShape &S, d3 pr) {
    if constexpr (KIND == ISDF_SHAPE_MESH) return mesh_sdf(S.mesh, pr);     // the mesh kind has its own instantiation
    else {
        if constexpr (KIND < 0) if (S.kind == ISDF_SHAPE_GRID) return grid_sdf(S, pr);      // (run-time kinds: Ball, sampled grid)
        return analytic_sdf<KIND, double, IDENT>(S.d, S.kind, pr);
    }
}

// BasicShape::getonlySDF(pos, R_obj) (Shape.hpp:470), the form the front end's kernel generation uses (:451): every
// analytic class evaluates its formula at (pos - trans) * Rotate * R_obj (e.g. :850-855, Box :2353-2361) - the body offset
// first, then R_obj; Ball ignores both (:621-624); the mesh Generalshape uses pos * R_obj (Shape.cpp:115-124).
template <int KIND>
__device__ __forceinline__ double shape_sdf_rotated(const DevShape &S, d3 pos, const double *Ro) {
    if constexpr (KIND == ISDF_SHAPE_MESH) {
        return mesh_sdf(S.mesh, mk3(pos.x * Ro[0] + pos.y * Ro[3] + pos.z * Ro[6], pos.x * Ro[1] + pos.y * Ro[4] + pos.z * Ro[7], pos.x * Ro[2] + pos.y * Ro[5] + pos.z * Ro[8]));
    } else {
        if (S.kind == ISDF_SHAPE_BALL) return norm3(pos) - S.d.p[0];
        if constexpr (KIND < 0) if (S.kind == ISDF_SHAPE_GRID)       // the lattice holds the shape incl. its body offset: sampled at pos * R_obj
            return grid_sdf(S, mk3(pos.x * Ro[0] + pos.y * Ro[3] + pos.z * Ro[6], pos.x * Ro[1] + pos.y * Ro[4] + pos.z * Ro[7], pos.x * Ro[2] + pos.y * Ro[5] + pos.z * Ro[8]));
        const d3 q = body_offset(S.d, pos);
        return analytic_local<KIND, double>(S.d, S.kind, mk3(q.x * Ro[0] + q.y * Ro[3] + q.z * Ro[6], q.x * Ro[1] + q.y * Ro[4] + q.z * Ro[7], q.x * Ro[2] + q.y * Ro[5] + q.z * Ro[8]));
    }
}

// getonlyGrad1: central difference dx = 5e-6 (temp -= dx; temp += 2dx), normalised — Shape.hpp:32-57;
// Box: forward difference dx = 0.01, not normalised — Shape.hpp:2363-2377; Ball: p/|p| — Shape.hpp:622-625.
template <int KIND, bool IDENT = false>
__device__ __forceinline__ d3 shape_grad(const DevShape &S, d3 pr) {
    if constexpr (KIND == ISDF_SHAPE_MESH) { d3 g; mesh_sdf_grad(S.mesh, pr, g); return g; }
    const int mode = S.grad_mode;   // resolved (never DEFAULT) by the host
    if (mode == ISDF_GRAD_ANALYTIC_BALL) return normalized3(pr);
    if constexpr (KIND < 0) if (mode == ISDF_GRAD_GRID) return grid_grad(S, pr);
    if (mode == ISDF_GRAD_BOX_FORWARD) {
        const double dx = 0.01;
        const double s0 = shape_sdf<KIND, IDENT>(S, pr);
        const double gx = shape_sdf<KIND, IDENT>(S, mk3(pr.x + dx, pr.y, pr.z)) - s0;
        const double gy = shape_sdf<KIND, IDENT>(S, mk3(pr.x, pr.y + dx, pr.z)) - s0;
        const double gz = shape_sdf<KIND, IDENT>(S, mk3(pr.x, pr.y, pr.z + dx)) - s0;
        return mk3(gx / dx, gy / dx, gz / dx);
    }
    const double dx = 0.000005;
    double lo, hi;
    lo = pr.x - dx; hi = lo + 2 * dx;
    const double gx = shape_sdf<KIND, IDENT>(S, mk3(hi, pr.y, pr.z)) - shape_sdf<KIND, IDENT>(S, mk3(lo, pr.y, pr.z));
    lo = pr.y - dx; hi = lo + 2 * dx;
    const double gy = shape_sdf<KIND, IDENT>(S, mk3(pr.x, hi, pr.z)) - shape_sdf<KIND, IDENT>(S, mk3(pr.x, lo, pr.z));
    lo = pr.z - dx; hi = lo + 2 * dx;
    const double gz = shape_sdf<KIND, IDENT>(S, mk3(pr.x, pr.y, hi)) - shape_sdf<KIND, IDENT>(S, mk3(pr.x, pr.y, lo));
#ifdef ISDF_LEAN_MATH
    return normalized3_lean(mk3(gx, gy, gz));      // normalising makes the common factor 1 / (2 dx) irrelevant
#else
    return normalized3(mk3(gx / (2 * dx), gy / (2 * dx), gz / (2 * dx)));
#endif
}

} // namespace isdf
