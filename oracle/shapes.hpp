// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked into the product path.
// CPU restatement of the robot-shape plugin (SDF + body-frame gradient) used on the hot path.
// Each function names the class in src/utils/include/utils/Shape.hpp whose getonlySDF it follows; the
// class constants are carried in isdf_shape.params (include/isdf_accel.h documents the slot meaning).
// Parity status: every analytic kind is pinned against the reference's OWN class (oracle/_ref/libref_shapes.so:
// Shape.hpp:824-2391 extracted at build time, compiled against the Eigen stand-in; tests/test_oracle_ref.py).
#pragma once
#include "vecmath.hpp"
#include "mesh_sdf.hpp"
#include "../include/isdf_accel.h"
#include <cfloat>

namespace orc {

inline double clipd(double v, double lo, double hi) { return std::max(std::min(v, hi), lo); }

// (pos_rel - trans) * Rotate   (row vector times matrix; Shape.hpp e.g. :843)
inline V3 body_offset(const isdf_shape &S, const V3 &p) {
    const double dx = p.x - S.trans[0], dy = p.y - S.trans[1], dz = p.z - S.trans[2];
    const double *R = S.rotate;
    return {dx * R[0] + dy * R[3] + dz * R[6], dx * R[1] + dy * R[4] + dz * R[7], dx * R[2] + dy * R[5] + dz * R[8]};
}

// q = |p| - half ; ||max(q,0)|| + min(max(q.x,q.y,q.z),0)   — the box formula repeated all over Shape.hpp
inline double box_q(const V3 &q) {
    const double qx = std::max(q.x, 0.0), qy = std::max(q.y, 0.0), qz = std::max(q.z, 0.0);
    return std::sqrt(qx * qx + qy * qy + qz * qz) + std::min(std::max(q.x, std::max(q.y, q.z)), 0.0);
}

// Torus / Torus_big  Shape.hpp:839-848, :875-884
inline double sdf_torus(const double *P, const V3 &p) {
    const double qx = std::sqrt(p.x * p.x + p.z * p.z) - P[0], qy = p.y;
    return std::sqrt(qx * qx + qy * qy) - P[1];
}
// Cappedtorus  Shape.hpp:914-920  (sc = (sin 40, cos 40) in RADIANS, 2.0f literal)
inline double sdf_cappedtorus(const double *P, V3 p) {
    const double scx = P[0], scy = P[1], ra = P[2], rb = P[3];
    p.x = std::fabs(p.x);
    const double k = (scy * p.x > scx * p.y) ? (p.x * scx + p.y * scy) : std::sqrt(p.x * p.x + p.y * p.y);
    return std::sqrt(dot(p, p) + ra * ra - 2.0f * ra * k) - rb;
}
// CappedCone  Shape.hpp:958-975  (note the double sqrt on d — reproduced)
inline double sdf_cappedcone(const double *P, const V3 &p) {
    const double ra = P[0], rb = P[1];
    const V3 a(P[2], P[3], P[4]), b(P[5], P[6], P[7]);
    const double rba = rb - ra;
    const double baba = dot(b - a, b - a);
    const double papa = dot(p - a, p - a);
    const double paba = dot(p - a, b - a) / baba;
    const double x = std::sqrt(papa - paba * paba * baba);
    const double cax = std::max(0.0, x - (paba < 0.5 ? ra : rb));
    const double cay = std::fabs(paba - 0.5) - 0.5;
    const double k = rba * rba + baba;
    const double f = clipd((rba * (x - ra) + paba * baba) / k, 0.0, 1.0);
    const double cbx = x - ra - f * rba;
    const double cby = paba - f;
    const double s = (cbx < 0 && cay < 0) ? -1 : 1;
    const double d = std::sqrt(std::min(cax * cax + cay * cay * baba, cbx * cbx + cby * cby * baba));
    return s * std::sqrt(std::fabs(d)) / std::fabs(baba);
}
// RoundedCone  Shape.hpp:1018-1029
inline double sdf_roundedcone(const double *P, const V3 &p) {
    const double r1 = P[0], r2 = P[1], h = P[2];
    const double qx = std::sqrt(p.x * p.x + p.y * p.y), qy = p.z;
    const double b = (r1 - r2) / h;
    const double a = std::sqrt(1.0 - b * b);
    const double k = -b * qx + a * qy;
    const double c1 = std::sqrt(qx * qx + qy * qy) - r1;
    const double c2 = std::sqrt(qx * qx + (qy - h) * (qy - h)) - r2;
    const double c3 = (a * qx + b * qy) - r1;
    return (k < 0) ? c1 : ((k > a * h) ? c2 : c3);
}
// WireframeBox  Shape.hpp:1067-1086
inline double wire_g(double a, double b, double c) {
    const double mx = std::max(a, 0.0), my = std::max(b, 0.0), mz = std::max(c, 0.0);
    return std::sqrt(mx * mx + my * my + mz * mz) + std::min(std::max(a, std::max(b, c)), 0.0);
}
inline double sdf_wireframebox(const double *P, const V3 &p) {
    const double th = P[3];
    V3 ps, q;
    for (int i = 0; i < 3; i++) {
        ps[i] = std::fabs(p[i]) - P[i] / 2 - th / 2;
        q[i] = std::fabs(ps[i] + th / 2) - th / 2;
    }
    return std::min(std::min(wire_g(ps.x, q.y, q.z), wire_g(q.x, ps.y, q.z)), wire_g(q.x, q.y, ps.z));
}
// BendLinear / BendLinear_big  Shape.hpp:1114-1154, :1179-1219
inline double sdf_bendlinear(const double *P, const V3 &p) {
    const double L = P[0], radius = P[1];
    const V3 p0(0, 0, -1), p1(0, 0, 1), v(-1, 0, 0);
    const V3 ab = p1 - p0;
    const double t = clipd(dot(p - p0, ab) / dot(ab, ab), 0, 1);
    const double u = 2 * t - 1;
    const double qa = 2 * t * t;
    const double qb = -0.5 * (u * (u - 2) - 1);
    const double e = (t < 0.5) ? qa : qb;
    const V3 pt = e * v + p;
    const V3 ca(0, 0, -L), cb(0, 0, L);
    const V3 pa = pt - ca, ba = cb - ca;
    const double h = std::max(std::min(dot(pa, ba) / dot(ba, ba), 1.0), 0.0);
    return norm(pa - h * ba) - radius;
}
// TwistBox  Shape.hpp:1253-1267 ; BendBox :1307-1321
inline double sdf_twistbox(const double *P, const V3 &p, bool bend) {
    const double k = P[3];
    const double ang = bend ? k * p.x : k * p.z;
    const double c = std::cos(ang), s = std::sin(ang);
    const V3 r(c * p.x - s * p.y, s * p.x + c * p.y, p.z);
    const V3 q(std::fabs(r.x) - P[0] / 2.0, std::fabs(r.y) - P[1] / 2.0, std::fabs(r.z) - P[2] / 2.0);
    return box_q(q);
}
// Table  Shape.hpp:1362-1381
inline double sdf_table(const double *P, const V3 &p) {
    const V3 w(std::fabs(p.x), std::fabs(p.y), p.z);
    double f[2];
    for (int t = 0; t < 2; t++) {
        const double *a = P + 6 * t, *b = P + 6 * t + 3;
        V3 q;
        for (int i = 0; i < 3; i++) q[i] = std::fabs(w[i] - (a[i] + b[i]) / 2) - (b[i] - a[i]) / 2.0;
        f[t] = box_q(q);
    }
    return std::min(f[0], f[1]);
}
// Trefoil  Shape.hpp:1463-1499
inline void rot2d(double &qx, double &qy, double a) {
    const double ca = std::cos(a), sa = std::sin(a);
    const double nx = qx * ca + qy * sa, ny = qy * ca + (-qx) * sa;
    qx = nx; qy = ny;
}
inline double sdf_trefoil(const double *P, const V3 &p) {
    const double r = P[0], bx = P[1], by = P[2], rr = P[3], sc = P[4];
    double qx = p.x, qy = p.y;
    const double py = -p.z;
    double dMin = 100;
    const double a = std::atan2(qy, qx);
    qx = std::sqrt(qx * qx + qy * qy) - r;
    qy = py;
    rot2d(qx, qy, 1.5 * a);
    rot2d(qx, qy, -M_PI * (std::floor(std::atan2(qy, qx) / M_PI + 0.5)));
    qx -= 1.0;
    const double dx = std::fabs(qx) - bx, dy = std::fabs(qy) - by;
    const double mx = std::max(dx, 0.0), my = std::max(dy, 0.0);
    const double d = (std::min(std::max(dx, dy), 0.0) + std::sqrt(mx * mx + my * my)) - rr;
    if (d < dMin) dMin = d;
    return sc * dMin;
}
// SmoothDifference  Shape.hpp:1535-1547 ; SmoothIntersection(_big) :1590-1601, :1645-1656
inline double sdf_smooth(const double *P, const V3 &p, bool intersection) {
    const V3 q(std::fabs(p.x) - P[0] / 2.0, std::fabs(p.y) - P[1] / 2.0, std::fabs(p.z) - P[2] / 2.0);
    const double box_sdf = box_q(q);
    const double sphere_sdf = norm(p) - P[3];
    const double kk = P[4];
    if (!intersection) {
        const double h = clipd(0.5 - 0.5 * (box_sdf + sphere_sdf) / kk, 0.0, 1.0);
        const double m = box_sdf - (box_sdf + sphere_sdf) * h;
        return m + kk * h * (1.0 - h);
    }
    const double h = clipd(0.5 - 0.5 * (sphere_sdf - box_sdf) / kk, 0.0, 1.0);
    const double m = sphere_sdf + (box_sdf - sphere_sdf) * h;
    return m + kk * h * (1.0 - h);
}
// CSG  Shape.hpp:2286-2298 = difference(intersection(sphere, box), union(union(c1,c2),c3)), all with k == 0,
// c = cylinder(r) about z, c1 = rotate_to(c,X,X) (returns c), c2 = rotate_to(c,X,Y), c3 = rotate_to(c,X,Z);
// rotate()/rotate_to() follow :2016-2053.
inline M3 csg_rotate_matrix(double angle, V3 axis) {
    const V3 n = normalized(axis);
    const double x = n.x, y = n.y, z = n.z;
    const double s = std::sin(angle), c = std::cos(angle), m = 1 - c;
    M3 R;
    R.m[0][0] = m * x * x + c;     R.m[0][1] = m * x * y + z * s; R.m[0][2] = m * z * x - y * s;
    R.m[1][0] = m * x * y - z * s; R.m[1][1] = m * y * y + c;     R.m[1][2] = m * y * z + x * s;
    R.m[2][0] = m * z * x + y * s; R.m[2][1] = m * y * z - x * s; R.m[2][2] = m * z * z + c;
    return R;
}
inline M3 csg_rotate_to(V3 a, V3 b, bool &is_identity) {
    a = normalized(a); b = normalized(b);
    const double d = dot(b, a);
    is_identity = false;
    if (std::fabs(d - 1) < FLT_EPSILON) { is_identity = true; return M3::identity(); }
    // the anti-parallel branch (:2046-2049) is unreachable for the axes CSG uses
    const double angle = std::acos(d);
    return csg_rotate_matrix(angle, cross(b, a));
}
inline double sdf_csg(const double *P, const V3 &p) {
    const double sr = P[0], be = P[1], cr = P[2];
    const double d_sphere = norm(p) - sr;
    const V3 q(std::fabs(p.x) - be / 2.0, std::fabs(p.y) - be / 2.0, std::fabs(p.z) - be / 2.0);
    const double d_box = box_q(q);
    const double f = std::max(d_sphere, d_box);
    auto cyl = [cr](const V3 &w) { return std::sqrt(w.x * w.x + w.y * w.y) - cr; };
    bool id;
    const M3 R2 = csg_rotate_to(V3(1, 0, 0), V3(0, 1, 0), id);
    const M3 R3 = csg_rotate_to(V3(1, 0, 0), V3(0, 0, 1), id);
    const double c1 = cyl(p), c2 = cyl(mul(R2, p)), c3 = cyl(mul(R3, p));
    const double c4 = std::min(std::min(c1, c2), c3);
    return std::max(f, -c4);
}
// Box  Shape.hpp:2343-2351 (params are HALF extents: q = |p| - selfbox)
inline double sdf_box(const double *P, const V3 &p) {
    return box_q(V3(std::fabs(p.x) - P[0], std::fabs(p.y) - P[1], std::fabs(p.z) - P[2]));
}

// the class formula at a point already in the shape's own frame (after the body offset)
inline double analytic_local(const isdf_shape &S, const V3 &p) {
    const double *P = S.params;
    switch (S.kind) {
    case ISDF_SHAPE_TORUS: return sdf_torus(P, p);
    case ISDF_SHAPE_CAPPEDTORUS: return sdf_cappedtorus(P, p);
    case ISDF_SHAPE_CAPPEDCONE: return sdf_cappedcone(P, p);
    case ISDF_SHAPE_ROUNDEDCONE: return sdf_roundedcone(P, p);
    case ISDF_SHAPE_WIREFRAMEBOX: return sdf_wireframebox(P, p);
    case ISDF_SHAPE_BENDLINEAR: return sdf_bendlinear(P, p);
    case ISDF_SHAPE_TWISTBOX: return sdf_twistbox(P, p, false);
    case ISDF_SHAPE_BENDBOX: return sdf_twistbox(P, p, true);
    case ISDF_SHAPE_TABLE: return sdf_table(P, p);
    case ISDF_SHAPE_TREFOIL: return sdf_trefoil(P, p);
    case ISDF_SHAPE_SMOOTHDIFFERENCE: return sdf_smooth(P, p, false);
    case ISDF_SHAPE_SMOOTHINTERSECTION: return sdf_smooth(P, p, true);
    case ISDF_SHAPE_CSG: return sdf_csg(P, p);
    case ISDF_SHAPE_BOX: return sdf_box(P, p);
    default: return 1e9;
    }
}

// ---- ISDF_SHAPE_GRID: getonlySDFNum (Shape.hpp:481-513), getonlyGrad1Num (:515-553), getSDFwithGrad1Num (:555-600)
struct GridRef { int idx[8]; double xd, yd, zd; bool inside; };
inline GridRef grid_locate(const MeshSdf &G, const V3 &p) {
    GridRef c;
    const int X = G.gn[0], Y = G.gn[1], Z = G.gn[2];
    const int i = (int)std::floor((p.x - G.gmin[0]) / G.gres), j = (int)std::floor((p.y - G.gmin[1]) / G.gres), k = (int)std::floor((p.z - G.gmin[2]) / G.gres);
    c.inside = i >= 0 && i < X - 1 && j >= 0 && j < Y - 1 && k >= 0 && k < Z - 1;
    c.xd = (p.x - G.gmin[0]) / G.gres - i; c.yd = (p.y - G.gmin[1]) / G.gres - j; c.zd = (p.z - G.gmin[2]) / G.gres - k;
    const int ijk = i * Y * Z + j * Z + k;
    c.idx[0] = ijk; c.idx[1] = ijk + Y * Z; c.idx[2] = ijk + Z; c.idx[3] = ijk + Y * Z + Z;
    c.idx[4] = ijk + 1; c.idx[5] = ijk + Y * Z + 1; c.idx[6] = ijk + Z + 1; c.idx[7] = ijk + Y * Z + Z + 1;
    return c;
}
inline double grid_value(const MeshSdf &G, const GridRef &c) {
    const double *m = G.grid_cells.data();
    auto D = [&](int q) { return m[4 * (size_t)c.idx[q] + 3]; };
    const double xd = c.xd, yd = c.yd, zd = c.zd;
    const double c00 = D(0) * (1 - xd) + D(1) * xd, c10 = D(2) * (1 - xd) + D(3) * xd;
    const double c01 = D(4) * (1 - xd) + D(5) * xd, c11 = D(6) * (1 - xd) + D(7) * xd;
    const double c0 = c00 * (1 - yd) + c10 * yd, c1 = c01 * (1 - yd) + c11 * yd;
    return c0 * (1 - zd) + c1 * zd;
}
inline V3 grid_gradient(const MeshSdf &G, const GridRef &c) {
    const double *m = G.grid_cells.data();
    const double xd = c.xd, yd = c.yd, zd = c.zd;
    // Eigen evaluates g * a * b * c component-wise from the left: ((g a) b) c; the eight terms are added from the left
    const double fa[8] = {1 - xd, xd, 1 - xd, xd, 1 - xd, xd, 1 - xd, xd};
    const double fb[8] = {1 - yd, 1 - yd, yd, yd, 1 - yd, 1 - yd, yd, yd};
    const double fc[8] = {1 - zd, 1 - zd, 1 - zd, 1 - zd, zd, zd, zd, zd};
    V3 g;
    for (int q = 0; q < 8; q++) {
        const double *n = m + 4 * (size_t)c.idx[q];
        for (int a = 0; a < 3; a++) {
            const double t = ((n[a] * fa[q]) * fb[q]) * fc[q];
            g[a] = q == 0 ? t : g[a] + t;
        }
    }
    return normalized(g);
}

// BasicShape::getonlySDF(pos_rel)  (Shape.hpp:469)
inline double shape_sdf(const isdf_shape &S, const MeshSdf *mesh, const V3 &p_rel) {
    if (S.kind == ISDF_SHAPE_GRID) { const GridRef c = grid_locate(*mesh, p_rel); return c.inside ? grid_value(*mesh, c) : 1e20; }
    if (S.kind == ISDF_SHAPE_MESH) return mesh->sdf(p_rel);          // vertices are already body-frame
    if (S.kind == ISDF_SHAPE_BALL) return norm(p_rel) - S.params[0]; // Ball/Point ignore trans/Rotate :617-620
    return analytic_local(S, body_offset(S, p_rel));
}

// BasicShape::getonlySDF(pos, R_obj)  (Shape.hpp:470): every analytic class evaluates its formula at
// (pos - trans) * Rotate * R_obj (e.g. :850-855, Box :2353-2361) - the body offset FIRST, then R_obj; Ball ignores both
// (:621-624); the mesh Generalshape uses pos * R_obj (Shape.cpp:115-124).  R_obj row-major.
inline double shape_sdf_rotated(const isdf_shape &S, const MeshSdf *mesh, const V3 &pos, const double *Ro) {
    if (S.kind == ISDF_SHAPE_BALL) return norm(pos) - S.params[0];
    if (S.kind == ISDF_SHAPE_GRID)
        return shape_sdf(S, mesh, V3(pos.x * Ro[0] + pos.y * Ro[3] + pos.z * Ro[6], pos.x * Ro[1] + pos.y * Ro[4] + pos.z * Ro[7], pos.x * Ro[2] + pos.y * Ro[5] + pos.z * Ro[8]));
    if (S.kind == ISDF_SHAPE_MESH)
        return mesh->sdf(V3(pos.x * Ro[0] + pos.y * Ro[3] + pos.z * Ro[6], pos.x * Ro[1] + pos.y * Ro[4] + pos.z * Ro[7], pos.x * Ro[2] + pos.y * Ro[5] + pos.z * Ro[8]));
    const V3 q = body_offset(S, pos);
    return analytic_local(S, V3(q.x * Ro[0] + q.y * Ro[3] + q.z * Ro[6], q.x * Ro[1] + q.y * Ro[4] + q.z * Ro[7], q.x * Ro[2] + q.y * Ro[5] + q.z * Ro[8]));
}

inline int effective_grad_mode(const isdf_shape &S) {
    if (S.grad_mode != ISDF_GRAD_DEFAULT) return S.grad_mode;
    if (S.kind == ISDF_SHAPE_BOX) return ISDF_GRAD_BOX_FORWARD;
    if (S.kind == ISDF_SHAPE_BALL) return ISDF_GRAD_ANALYTIC_BALL;
    return ISDF_GRAD_CENTRAL;
}

// BasicShape::getonlyGrad1(pos_rel)  — DEFINE_USEFUL_FUNCTION Shape.hpp:32-57 (the "temp(a) -= dx ... += 2*dx"
// sequence is reproduced literally), Box::getonlyGrad1 :2363-2377, Ball :622-625, mesh Shape.cpp:127-139.
inline V3 shape_grad(const isdf_shape &S, const MeshSdf *mesh, const V3 &p_rel) {
    if (S.kind == ISDF_SHAPE_MESH) { V3 g; mesh->sdf_grad(p_rel, g); return g; }
    if (S.kind == ISDF_SHAPE_GRID) { const GridRef c = grid_locate(*mesh, p_rel); return c.inside ? grid_gradient(*mesh, c) : V3(); }
    const int mode = effective_grad_mode(S);
    if (mode == ISDF_GRAD_ANALYTIC_BALL) return normalized(p_rel);
    if (mode == ISDF_GRAD_BOX_FORWARD) {
        const double dx = 0.01;
        const double sdfold = shape_sdf(S, mesh, p_rel);
        V3 g;
        for (int a = 0; a < 3; a++) {
            V3 t = p_rel;
            t[a] += dx;
            g[a] = shape_sdf(S, mesh, t) - sdfold;
        }
        return g / dx;
    }
    const double dx = 0.000005;
    V3 g;
    for (int a = 0; a < 3; a++) {
        V3 t = p_rel;
        t[a] -= dx;
        const double sdfold = shape_sdf(S, mesh, t);
        t[a] += 2 * dx;
        g[a] = shape_sdf(S, mesh, t) - sdfold;
    }
    return normalized(g / (2 * dx));
}

// BasicShape::getSDFwithGrad1(pos_rel, grad)  Shape.hpp:58-87 (helperfunc), Box :2379-2388, Shape.cpp:140-151
inline double shape_sdf_grad(const isdf_shape &S, const MeshSdf *mesh, const V3 &p_rel, V3 &grad) {
    if (S.kind == ISDF_SHAPE_MESH) return mesh->sdf_grad(p_rel, grad);
    grad = shape_grad(S, mesh, p_rel);
    return shape_sdf(S, mesh, p_rel);
}

} // namespace orc
