// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked into the product path.
// Mesh robot shape: sdf = (1 - 2 w(p)) * dist(p, mesh), gradient = normalize((1 - 2w) * (p - c)).
//   follows src/utils/src/Shape.cpp:105-151 (Generalshape::getonlySDF / getonlyGrad1 / getSDFwithGrad1).
// The reference gets w from libigl 2.4.0 igl::fast_winding_number(fwn_bvh, 2.0, p, w)
// (UT_SolidAngle<float,float>, order 2, Shape.cpp:86,110) and (c, dist^2) from igl::AABB::squared_distance
// (Shape.cpp:85,112).  libigl's core .cpp files and lib/libigl.a are NOT in /root/reference
// (SURVEY.md §8c), so:
//   * distance / closest point: exact geometry, brute force over triangles with the region-based
//     closest-point-on-triangle test (Ericson, Real-Time Collision Detection §5.1.5 — the algorithm behind
//     igl::point_simplex_squared_distance);
//   * winding number: the exact sum of triangle solid angles (Van Oosterom & Strackee 1983 — the formula
//     UT_SolidAngle uses for leaf triangles), evaluated in float like the reference, divided by 4*pi.
//     libigl's far-field order-2 Taylor clusters are an approximation of this sum; they are only used for
//     queries farther than 2x a cluster's radius, where the collision penalty is inactive.
// PARITY UNPINNED for this kind: no reference test or golden vector exists at this boundary; oracle/_ref
// builds the vendored FastWindingNumberForSoups.h to quantify the far-field deviation (tests/test_oracle_ref.py).
#pragma once
#include "vecmath.hpp"
#include <vector>
#include <cstdint>

namespace orc {

struct MeshSdf {
    std::vector<V3> V;
    std::vector<int32_t> F;   // 3 per face

    void set(const double *verts, int nV, const int32_t *faces, int nF) {
        V.resize(nV);
        for (int i = 0; i < nV; i++) V[i] = V3(verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]);
        F.assign(faces, faces + 3 * (size_t)nF);
    }
    int n_faces() const { return (int)(F.size() / 3); }

    static V3 closest_on_triangle(const V3 &p, const V3 &a, const V3 &b, const V3 &c) {
        const V3 ab = b - a, ac = c - a, ap = p - a;
        const double d1 = dot(ab, ap), d2 = dot(ac, ap);
        if (d1 <= 0.0 && d2 <= 0.0) return a;
        const V3 bp = p - b;
        const double d3 = dot(ab, bp), d4 = dot(ac, bp);
        if (d3 >= 0.0 && d4 <= d3) return b;
        const double vc = d1 * d4 - d3 * d2;
        if (vc <= 0.0 && d1 >= 0.0 && d3 <= 0.0) {
            const double v = d1 / (d1 - d3);
            return a + v * ab;
        }
        const V3 cp = p - c;
        const double d5 = dot(ab, cp), d6 = dot(ac, cp);
        if (d6 >= 0.0 && d5 <= d6) return c;
        const double vb = d5 * d2 - d1 * d6;
        if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) {
            const double w = d2 / (d2 - d6);
            return a + w * ac;
        }
        const double va = d3 * d6 - d5 * d4;
        if (va <= 0.0 && (d4 - d3) >= 0.0 && (d5 - d6) >= 0.0) {
            const double w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
            return b + w * (c - b);
        }
        const double denom = 1.0 / (va + vb + vc);
        const double v = vb * denom, w = vc * denom;
        return a + v * ab + w * ac;
    }

    double squared_distance(const V3 &p, V3 &closest) const {
        double best = DBL_MAX;
        for (int f = 0; f < n_faces(); f++) {
            const V3 c = closest_on_triangle(p, V[F[3 * f]], V[F[3 * f + 1]], V[F[3 * f + 2]]);
            const double d = sqnorm(p - c);
            if (d < best) { best = d; closest = c; }
        }
        return best;
    }

    // float arithmetic like UT_SolidAngle<float,float>; 2*atan2(det, denom) per triangle
    double winding_number(const V3 &p) const {
        float total = 0.0f;
        const float qx = (float)p.x, qy = (float)p.y, qz = (float)p.z;
        for (int f = 0; f < n_faces(); f++) {
            float a[3], b[3], c[3];
            const V3 &A = V[F[3 * f]], &B = V[F[3 * f + 1]], &C = V[F[3 * f + 2]];
            a[0] = (float)A.x - qx; a[1] = (float)A.y - qy; a[2] = (float)A.z - qz;
            b[0] = (float)B.x - qx; b[1] = (float)B.y - qy; b[2] = (float)B.z - qz;
            c[0] = (float)C.x - qx; c[1] = (float)C.y - qy; c[2] = (float)C.z - qz;
            const float la = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
            const float lb = std::sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
            const float lc = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
            if (la == 0 || lb == 0 || lc == 0) continue;
            for (int i = 0; i < 3; i++) { a[i] /= la; b[i] /= lb; c[i] /= lc; }
            const float det = a[0] * (b[1] * c[2] - b[2] * c[1]) - a[1] * (b[0] * c[2] - b[2] * c[0]) +
                              a[2] * (b[0] * c[1] - b[1] * c[0]);
            const float ab = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
            const float bc = b[0] * c[0] + b[1] * c[1] + b[2] * c[2];
            const float ca = c[0] * a[0] + c[1] * a[1] + c[2] * a[2];
            const float den = 1.0f + ab + bc + ca;
            total += 2.0f * std::atan2(det, den);
        }
        return (double)(total / (4.0f * 3.14159265358979323846f));
    }

    double sdf(const V3 &p) const {
        V3 c;
        const double w = winding_number(p);
        const double s = 1. - 2. * w;
        return s * std::sqrt(squared_distance(p, c));
    }
    double sdf_grad(const V3 &p, V3 &grad) const {
        V3 c;
        const double w = winding_number(p);
        const double s = 1. - 2. * w;
        const double d2 = squared_distance(p, c);
        grad = normalized(s * (-(c - p)));
        return s * std::sqrt(d2);
    }
};

} // namespace orc
