// Device-side mesh robot shape (Generalshape, src/utils/src/Shape.cpp:105-151):
//   sdf = (1 - 2 w) * dist,  grad = normalize((1 - 2 w) * (p - c)),
// w = winding number of the triangle soup (float, like libigl's UT_SolidAngle<float,float>), c = closest point.
// Round-1 implementation: flat loops over the triangles (exact solid-angle sum, exact closest point), triangles
// stored SoA-packed as 9 doubles + 9 floats per face.  Suited to the demo meshes (mybox 12, Lthick 20 faces);
// a BVH with libigl's far-field expansion is the planned follow-up for the 2-12 k face meshes.
#pragma once
#include "dev_shapes.hpp"

namespace isdf {

// struct DevMesh lives in dev_shapes.hpp

__device__ __forceinline__ d3 closest_on_triangle(d3 p, d3 a, d3 b, d3 c) {
    const d3 ab = b - a, ac = c - a, ap = p - a;
    const double d1 = dot3(ab, ap), d2 = dot3(ac, ap);
    if (d1 <= 0.0 && d2 <= 0.0) return a;
    const d3 bp = p - b;
    const double d3_ = dot3(ab, bp), d4 = dot3(ac, bp);
    if (d3_ >= 0.0 && d4 <= d3_) return b;
    const double vc = d1 * d4 - d3_ * d2;
    if (vc <= 0.0 && d1 >= 0.0 && d3_ <= 0.0) return a + (d1 / (d1 - d3_)) * ab;
    const d3 cp = p - c;
    const double d5 = dot3(ab, cp), d6 = dot3(ac, cp);
    if (d6 >= 0.0 && d5 <= d6) return c;
    const double vb = d5 * d2 - d1 * d6;
    if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) return a + (d2 / (d2 - d6)) * ac;
    const double va = d3_ * d6 - d5 * d4;
    if (va <= 0.0 && (d4 - d3_) >= 0.0 && (d5 - d6) >= 0.0) return b + ((d4 - d3_) / ((d4 - d3_) + (d5 - d6))) * (c - b);
    const double denom = 1.0 / (va + vb + vc);
    return a + (vb * denom) * ab + (vc * denom) * ac;
}

__device__ inline double mesh_query(const DevMesh *m, d3 p, d3 &closest, double &s_out) {
    double best = 1.7976931348623157e308;
    float total = 0.0f;
    const float qx = (float)p.x, qy = (float)p.y, qz = (float)p.z;
    for (int f = 0; f < m->nF; f++) {
        const double *t = m->tri + 9 * f;
        const d3 c = closest_on_triangle(p, mk3(t[0], t[1], t[2]), mk3(t[3], t[4], t[5]), mk3(t[6], t[7], t[8]));
        const d3 dd = p - c;
        const double d = dot3(dd, dd);
        if (d < best) { best = d; closest = c; }
        const float *u = m->trif + 9 * f;
        float ax = u[0] - qx, ay = u[1] - qy, az = u[2] - qz;
        float bx = u[3] - qx, by = u[4] - qy, bz = u[5] - qz;
        float cx = u[6] - qx, cy = u[7] - qy, cz = u[8] - qz;
        const float la = sqrtf(ax * ax + ay * ay + az * az), lb = sqrtf(bx * bx + by * by + bz * bz),
                    lc = sqrtf(cx * cx + cy * cy + cz * cz);
        if (la == 0 || lb == 0 || lc == 0) continue;
        ax /= la; ay /= la; az /= la; bx /= lb; by /= lb; bz /= lb; cx /= lc; cy /= lc; cz /= lc;
        const float det = ax * (by * cz - bz * cy) - ay * (bx * cz - bz * cx) + az * (bx * cy - by * cx);
        const float den = 1.0f + (ax * bx + ay * by + az * bz) + (bx * cx + by * cy + bz * cz) + (cx * ax + cy * ay + cz * az);
        total += 2.0f * atan2f(det, den);
    }
    const double w = (double)(total / (4.0f * 3.14159265358979323846f));
    s_out = 1. - 2. * w;
    return best;
}

__device__ inline double mesh_sdf(const DevMesh *m, d3 p) {
    d3 c; double s;
    const double d2 = mesh_query(m, p, c, s);
    return s * sqrt(d2);
}
__device__ inline double mesh_sdf_grad(const DevMesh *m, d3 p, d3 &g) {
    d3 c; double s;
    const double d2 = mesh_query(m, p, c, s);
    g = normalized3(s * (p - c));
    return s * sqrt(d2);
}

} // namespace isdf
