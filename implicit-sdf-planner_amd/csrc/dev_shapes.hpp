// Device-side robot-shape plugin: analytic SDF kinds + body-frame gradient (gfx950, fp64).
// The host Shape registry (sw_manager.hpp:74-123) maps to DevShape.kind; the class constants of
// src/utils/include/utils/Shape.hpp:824-2391 travel in DevShape.p (slot meaning: include/isdf_accel.h).
// KIND is a template parameter so the hot kernels inline exactly one shape; KIND = -1 dispatches at run time
// (kind is wave-uniform, so the switch does not diverge).
#pragma once
#include "dev_math.hpp"
#include "../../include/isdf_accel.h"

namespace isdf {

// mesh kind (functions in dev_mesh.hpp)
struct DevMesh {
    const double *tri;   // nF x 9: a.xyz b.xyz c.xyz
    const float *trif;   // nF x 9 (float copies for the winding number)
    int nF;
};

struct DevShape {
    int kind, grad_mode;
    double p[16];
    double trans[3];
    double rot[9];
    double csg_r2[9], csg_r3[9];   // rotate_to(c, X, Y) and rotate_to(c, X, Z) matrices, built on the host
    double bound_radius;
    const DevMesh *mesh;
};

__device__ __forceinline__ double clipd(double v, double lo, double hi) { return fmax(fmin(v, hi), lo); }
__device__ __forceinline__ double box_q(double qx, double qy, double qz) {
    const double mx = fmax(qx, 0.0), my = fmax(qy, 0.0), mz = fmax(qz, 0.0);
    return sqrt(mx * mx + my * my + mz * mz) + fmin(fmax(qx, fmax(qy, qz)), 0.0);
}

__device__ __forceinline__ double sdf_torus(const double *P, d3 p) {
    const double qx = sqrt(p.x * p.x + p.z * p.z) - P[0];
    return sqrt(qx * qx + p.y * p.y) - P[1];
}
__device__ __forceinline__ double sdf_cappedtorus(const double *P, d3 p) {
    const double scx = P[0], scy = P[1], ra = P[2], rb = P[3];
    p.x = fabs(p.x);
    const double k = (scy * p.x > scx * p.y) ? (p.x * scx + p.y * scy) : sqrt(p.x * p.x + p.y * p.y);
    return sqrt(dot3(p, p) + ra * ra - 2.0 * ra * k) - rb;
}
__device__ __forceinline__ double sdf_cappedcone(const double *P, d3 p) {
    const double ra = P[0], rb = P[1];
    const d3 a = mk3(P[2], P[3], P[4]), b = mk3(P[5], P[6], P[7]);
    const double rba = rb - ra;
    const d3 ba = b - a, pa = p - a;
    const double baba = dot3(ba, ba);
    const double papa = dot3(pa, pa);
    const double paba = dot3(pa, ba) / baba;
    const double x = sqrt(papa - paba * paba * baba);
    const double cax = fmax(0.0, x - (paba < 0.5 ? ra : rb));
    const double cay = fabs(paba - 0.5) - 0.5;
    const double k = rba * rba + baba;
    const double f = clipd((rba * (x - ra) + paba * baba) / k, 0.0, 1.0);
    const double cbx = x - ra - f * rba;
    const double cby = paba - f;
    const double s = (cbx < 0 && cay < 0) ? -1.0 : 1.0;
    const double d = sqrt(fmin(cax * cax + cay * cay * baba, cbx * cbx + cby * cby * baba));
    return s * sqrt(fabs(d)) / fabs(baba);
}
__device__ __forceinline__ double sdf_roundedcone(const double *P, d3 p) {
    const double r1 = P[0], r2 = P[1], h = P[2];
    const double qx = sqrt(p.x * p.x + p.y * p.y), qy = p.z;
    const double b = (r1 - r2) / h;
    const double a = sqrt(1.0 - b * b);
    const double k = -b * qx + a * qy;
    const double c1 = sqrt(qx * qx + qy * qy) - r1;
    const double c2 = sqrt(qx * qx + (qy - h) * (qy - h)) - r2;
    const double c3 = (a * qx + b * qy) - r1;
    return (k < 0) ? c1 : ((k > a * h) ? c2 : c3);
}
__device__ __forceinline__ double sdf_wireframebox(const double *P, d3 p) {
    const double th = P[3];
    const double px = fabs(p.x) - P[0] / 2 - th / 2, py = fabs(p.y) - P[1] / 2 - th / 2, pz = fabs(p.z) - P[2] / 2 - th / 2;
    const double qx = fabs(px + th / 2) - th / 2, qy = fabs(py + th / 2) - th / 2, qz = fabs(pz + th / 2) - th / 2;
    return fmin(fmin(box_q(px, qy, qz), box_q(qx, py, qz)), box_q(qx, qy, pz));
}
__device__ __forceinline__ double sdf_bendlinear(const double *P, d3 p) {
    const double L = P[0], radius = P[1];
    // p0 = (0,0,-1), p1 = (0,0,1), v = (-1,0,0), ab = (0,0,2)
    const double t = clipd(((p.x - 0.0) * 0.0 + (p.y - 0.0) * 0.0 + (p.z + 1.0) * 2.0) / 4.0, 0.0, 1.0);
    const double u = 2 * t - 1;
    const double e = (t < 0.5) ? (2 * t * t) : (-0.5 * (u * (u - 2) - 1));
    const d3 pt = mk3(e * -1.0 + p.x, e * 0.0 + p.y, e * 0.0 + p.z);
    const d3 pa = mk3(pt.x, pt.y, pt.z + L);
    const d3 ba = mk3(0.0, 0.0, 2 * L);
    const double h = fmax(fmin(dot3(pa, ba) / dot3(ba, ba), 1.0), 0.0);
    return norm3(pa - h * ba) - radius;
}
__device__ __forceinline__ double sdf_twistbend(const double *P, d3 p, bool bend) {
    const double ang = bend ? P[3] * p.x : P[3] * p.z;
    double s, c;
    sincos(ang, &s, &c);
    const double rx = c * p.x - s * p.y, ry = s * p.x + c * p.y;
    return box_q(fabs(rx) - P[0] / 2.0, fabs(ry) - P[1] / 2.0, fabs(p.z) - P[2] / 2.0);
}
__device__ __forceinline__ double sdf_table(const double *P, d3 p) {
    const double x = fabs(p.x), y = fabs(p.y), z = p.z;
    const double f1 = box_q(fabs(x - (P[0] + P[3]) / 2) - (P[3] - P[0]) / 2.0, fabs(y - (P[1] + P[4]) / 2) - (P[4] - P[1]) / 2.0,
                            fabs(z - (P[2] + P[5]) / 2) - (P[5] - P[2]) / 2.0);
    const double f2 = box_q(fabs(x - (P[6] + P[9]) / 2) - (P[9] - P[6]) / 2.0, fabs(y - (P[7] + P[10]) / 2) - (P[10] - P[7]) / 2.0,
                            fabs(z - (P[8] + P[11]) / 2) - (P[11] - P[8]) / 2.0);
    return fmin(f1, f2);
}
__device__ __forceinline__ void rot2d(double &qx, double &qy, double a) {
    double sa, ca;
    sincos(a, &sa, &ca);
    const double nx = qx * ca + qy * sa, ny = qy * ca + (-qx) * sa;
    qx = nx; qy = ny;
}
__device__ __forceinline__ double sdf_trefoil(const double *P, d3 p) {
    const double PI = 3.14159265358979323846;
    double qx = p.x, qy = p.y;
    const double a = atan2(qy, qx);
    qx = sqrt(qx * qx + qy * qy) - P[0];
    qy = -p.z;
    rot2d(qx, qy, 1.5 * a);
    rot2d(qx, qy, -PI * (floor(atan2(qy, qx) / PI + 0.5)));
    qx -= 1.0;
    const double dx = fabs(qx) - P[1], dy = fabs(qy) - P[2];
    const double mx = fmax(dx, 0.0), my = fmax(dy, 0.0);
    const double d = (fmin(fmax(dx, dy), 0.0) + sqrt(mx * mx + my * my)) - P[3];
    return P[4] * fmin(d, 100.0);
}
__device__ __forceinline__ double sdf_smooth(const double *P, d3 p, bool intersection) {
    const double box_sdf = box_q(fabs(p.x) - P[0] / 2.0, fabs(p.y) - P[1] / 2.0, fabs(p.z) - P[2] / 2.0);
    const double sphere_sdf = norm3(p) - P[3];
    const double kk = P[4];
    if (!intersection) {
        const double h = clipd(0.5 - 0.5 * (box_sdf + sphere_sdf) / kk, 0.0, 1.0);
        return (box_sdf - (box_sdf + sphere_sdf) * h) + kk * h * (1.0 - h);
    }
    const double h = clipd(0.5 - 0.5 * (sphere_sdf - box_sdf) / kk, 0.0, 1.0);
    return (sphere_sdf + (box_sdf - sphere_sdf) * h) + kk * h * (1.0 - h);
}
__device__ __forceinline__ double sdf_csg(const DevShape &S, d3 p) {
    const double *P = S.p;
    const double f = fmax(norm3(p) - P[0], box_q(fabs(p.x) - P[1] / 2.0, fabs(p.y) - P[1] / 2.0, fabs(p.z) - P[1] / 2.0));
    const double c1 = sqrt(p.x * p.x + p.y * p.y) - P[2];
    const double *A = S.csg_r2, *B = S.csg_r3;
    const double ax = A[0] * p.x + A[1] * p.y + A[2] * p.z, ay = A[3] * p.x + A[4] * p.y + A[5] * p.z;
    const double bx = B[0] * p.x + B[1] * p.y + B[2] * p.z, by = B[3] * p.x + B[4] * p.y + B[5] * p.z;
    const double c2 = sqrt(ax * ax + ay * ay) - P[2];
    const double c3 = sqrt(bx * bx + by * by) - P[2];
    return fmax(f, -fmin(fmin(c1, c2), c3));
}

__device__ double mesh_sdf(const DevMesh *m, d3 p);              // dev_mesh.hpp
__device__ double mesh_sdf_grad(const DevMesh *m, d3 p, d3 &g);  // dev_mesh.hpp

template <int KIND>
__device__ __forceinline__ double shape_sdf(const DevShape &S, d3 pr) {
    const int kind = (KIND >= 0) ? KIND : S.kind;
    if (kind == ISDF_SHAPE_MESH) return mesh_sdf(S.mesh, pr);
    if (kind == ISDF_SHAPE_BALL) return norm3(pr) - S.p[0];
    const double dx = pr.x - S.trans[0], dy = pr.y - S.trans[1], dz = pr.z - S.trans[2];
    const double *R = S.rot;
    const d3 p = mk3(dx * R[0] + dy * R[3] + dz * R[6], dx * R[1] + dy * R[4] + dz * R[7], dx * R[2] + dy * R[5] + dz * R[8]);
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
    case ISDF_SHAPE_BOX: return box_q(fabs(p.x) - S.p[0], fabs(p.y) - S.p[1], fabs(p.z) - S.p[2]);
    default: return 1e9;
    }
}

// getonlyGrad1: central difference dx = 5e-6 (temp -= dx; temp += 2dx), normalised — Shape.hpp:32-57;
// Box: forward difference dx = 0.01, not normalised — Shape.hpp:2363-2377; Ball: p/|p| — Shape.hpp:622-625.
template <int KIND>
__device__ __forceinline__ d3 shape_grad(const DevShape &S, d3 pr) {
    const int kind = (KIND >= 0) ? KIND : S.kind;
    if (kind == ISDF_SHAPE_MESH) { d3 g; mesh_sdf_grad(S.mesh, pr, g); return g; }
    const int mode = S.grad_mode;   // resolved (never DEFAULT) by the host
    if (mode == ISDF_GRAD_ANALYTIC_BALL) return normalized3(pr);
    if (mode == ISDF_GRAD_BOX_FORWARD) {
        const double dx = 0.01;
        const double s0 = shape_sdf<KIND>(S, pr);
        const double gx = shape_sdf<KIND>(S, mk3(pr.x + dx, pr.y, pr.z)) - s0;
        const double gy = shape_sdf<KIND>(S, mk3(pr.x, pr.y + dx, pr.z)) - s0;
        const double gz = shape_sdf<KIND>(S, mk3(pr.x, pr.y, pr.z + dx)) - s0;
        return mk3(gx / dx, gy / dx, gz / dx);
    }
    const double dx = 0.000005;
    double lo, hi;
    lo = pr.x - dx; hi = lo + 2 * dx;
    const double gx = shape_sdf<KIND>(S, mk3(hi, pr.y, pr.z)) - shape_sdf<KIND>(S, mk3(lo, pr.y, pr.z));
    lo = pr.y - dx; hi = lo + 2 * dx;
    const double gy = shape_sdf<KIND>(S, mk3(pr.x, hi, pr.z)) - shape_sdf<KIND>(S, mk3(pr.x, lo, pr.z));
    lo = pr.z - dx; hi = lo + 2 * dx;
    const double gz = shape_sdf<KIND>(S, mk3(pr.x, pr.y, hi)) - shape_sdf<KIND>(S, mk3(pr.x, pr.y, lo));
    return normalized3(mk3(gx / (2 * dx), gy / (2 * dx), gz / (2 * dx)));
}

} // namespace isdf
