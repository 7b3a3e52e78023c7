// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked into the product path.
// Mesh robot shape: sdf = (1 - 2 w(p)) * dist(p, mesh), gradient = normalize((1 - 2w) * (p - c)).
//   follows src/utils/src/Shape.cpp:105-151 (Generalshape::getonlySDF / getonlyGrad1 / getSDFwithGrad1).
// The reference gets w from libigl 2.4.0 igl::fast_winding_number(fwn_bvh, 2.0, p, w)
// (UT_SolidAngle<float,float>, order 2, Shape.cpp:86,110) and (c, dist^2) from igl::AABB::squared_distance
// (Shape.cpp:85,112).  libigl's core .cpp files and lib/libigl.a are NOT in /root/reference, but the
// HDK winding-number core IS (vendored header src/utils/include/igl/FastWindingNumberForSoups.h), so:
//   * winding number: the REFERENCE'S OWN library - oracle/_ref/libref_fwn.so (that header compiled where it lies, see
//     ref_build/ref_fwn.cpp) is loaded at run time and asked for w.  Its far-field order-2 clusters deviate from the
//     exact winding number by up to ~2e-3, so a restatement of "the exact sum" is NOT the reference's function; the
//     exact float sum (Van Oosterom & Strackee 1983, the leaf formula of UT_SolidAngle) is kept only as the fallback
//     when libref_fwn.so has not been built (backend() == 0: parity UNPINNED for the mesh kind in that case, and the
//     mesh parity tests skip);
//   * distance / closest point: exact geometry, the region-based closest-point-on-triangle test (Ericson, Real-Time
//     Collision Detection 5.1.5 - the algorithm behind igl::point_simplex_squared_distance) over ALL triangles; igl::AABB
//     only accelerates the same exact query.  Meshes of more than 64 faces go through a plain bounding-box tree here too
//     (branch and bound, the exhaustive search's answer bit for bit including its lowest-face tie rule; ORACLE_MESH_EXHAUSTIVE=1
//     at set() time keeps the exhaustive loop - tests/test_oracle_selfcheck.py compares the two).
#pragma once
#include "vecmath.hpp"
#include <algorithm>
#include <cfloat>
#include <vector>
#include <cstdint>
#include <cstdlib>
#include <string>
#include <dlfcn.h>

namespace orc {

// oracle/_ref/libref_fwn.so (the vendored HDK library behind libigl's fast_winding_number), loaded once
struct HdkFwn {
    void *(*create)(const double *, int, const int *, int, int) = nullptr;
    void (*destroy)(void *) = nullptr;
    void (*eval)(void *, const double *, int, double, double *) = nullptr;
    bool ok = false;
    static const HdkFwn &get() {
        static HdkFwn inst = [] {
            HdkFwn h;
            const char *off = std::getenv("ORACLE_MESH_EXACT_WINDING");
            if (off && off[0] == '1') return h;
            Dl_info info;
            if (!dladdr((const void *)&HdkFwn::get, &info) || !info.dli_fname) return h;
            std::string dir(info.dli_fname);
            const size_t slash = dir.find_last_of('/');
            dir = (slash == std::string::npos) ? std::string(".") : dir.substr(0, slash);
            void *lib = dlopen((dir + "/_ref/libref_fwn.so").c_str(), RTLD_NOW | RTLD_LOCAL);
            if (!lib) return h;
            h.create = (decltype(h.create))dlsym(lib, "ref_fwn_create");
            h.destroy = (decltype(h.destroy))dlsym(lib, "ref_fwn_destroy");
            h.eval = (decltype(h.eval))dlsym(lib, "ref_fwn_eval");
            h.ok = h.create && h.destroy && h.eval;
            return h;
        }();
        return inst;
    }
};

struct MeshSdf {
    // ISDF_SHAPE_GRID: BasicShape's num_sdf_map (Shape.hpp:267-288) - 4 doubles per node (unit gradient, distance), toAddr order
    std::vector<double> grid_cells;
    int gn[3] = {0, 0, 0};
    double gres = 0.0, gmin[3] = {0, 0, 0};
    std::vector<V3> V;
    std::vector<int32_t> F;   // 3 per face
    void *hdk = nullptr;      // fwn_bvh of Shape.cpp:86 when the vendored library is available
    // bounding-box tree over the faces (closest point only): node = box, children (inner) or a run of `order` (leaf)
    struct BvhNode { double lo[3], hi[3]; int left, right, first, count; };
    std::vector<BvhNode> bvh;
    std::vector<int> order;

    MeshSdf() = default;
    MeshSdf(const MeshSdf &) = delete;
    MeshSdf &operator=(const MeshSdf &) = delete;
    ~MeshSdf() { if (hdk) HdkFwn::get().destroy(hdk); }
    int backend() const { return hdk ? 1 : 0; }     // 1: vendored HDK library, 0: exact sum (fallback)

    void set(const double *verts, int nV, const int32_t *faces, int nF) {
        V.resize(nV);
        for (int i = 0; i < nV; i++) V[i] = V3(verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]);
        F.assign(faces, faces + 3 * (size_t)nF);
        const HdkFwn &lib = HdkFwn::get();
        if (hdk) { lib.destroy(hdk); hdk = nullptr; }
        if (lib.ok) hdk = lib.create(verts, nV, faces, nF, 2);            // igl::fast_winding_number(V, F, 2, fwn_bvh)
        bvh.clear(); order.clear();
        const char *ex = std::getenv("ORACLE_MESH_EXHAUSTIVE");
        if (nF > 64 && !(ex && ex[0] == '1')) build_bvh();
    }
    void face_box(int f, double lo[3], double hi[3]) const {
        for (int a = 0; a < 3; a++) { lo[a] = DBL_MAX; hi[a] = -DBL_MAX; }
        for (int k = 0; k < 3; k++) {
            const V3 &v = V[F[3 * f + k]];
            const double c[3] = {v.x, v.y, v.z};
            for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], c[a]); hi[a] = std::max(hi[a], c[a]); }
        }
    }
    int build_node(int first, int count) {
        BvhNode nd;
        for (int a = 0; a < 3; a++) { nd.lo[a] = DBL_MAX; nd.hi[a] = -DBL_MAX; }
        double clo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, chi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
        for (int k = first; k < first + count; k++) {
            double lo[3], hi[3];
            face_box(order[k], lo, hi);
            for (int a = 0; a < 3; a++) {
                nd.lo[a] = std::min(nd.lo[a], lo[a]); nd.hi[a] = std::max(nd.hi[a], hi[a]);
                const double c = 0.5 * (lo[a] + hi[a]);
                clo[a] = std::min(clo[a], c); chi[a] = std::max(chi[a], c);
            }
        }
        nd.left = nd.right = -1; nd.first = first; nd.count = count;
        const int me = (int)bvh.size();
        bvh.push_back(nd);
        if (count > 8) {
            int ax = 0;
            for (int a = 1; a < 3; a++) if (chi[a] - clo[a] > chi[ax] - clo[ax]) ax = a;
            auto cen = [&](int f) { double lo[3], hi[3]; face_box(f, lo, hi); return 0.5 * (lo[ax] + hi[ax]); };
            const int mid = first + count / 2;
            std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + first + count, [&](int x, int y) { return cen(x) < cen(y); });
            const int l = build_node(first, mid - first), r = build_node(mid, first + count - mid);
            bvh[me].left = l; bvh[me].right = r; bvh[me].count = 0;
        }
        return me;
    }
    void build_bvh() {
        order.resize(n_faces());
        for (int f = 0; f < n_faces(); f++) order[f] = f;
        bvh.reserve(2 * (size_t)n_faces() / 4 + 8);
        build_node(0, n_faces());
    }
    static double box_dist2(const BvhNode &n, const V3 &p) {
        const double c[3] = {p.x, p.y, p.z};
        double s = 0.0;
        for (int a = 0; a < 3; a++) { const double e = std::max(std::max(n.lo[a] - c[a], c[a] - n.hi[a]), 0.0); s += e * e; }
        return s * (1.0 - 1e-12);           // (never above the true distance to anything inside, rounding included)
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
        if (!bvh.empty()) {
            // branch and bound; a node is skipped only when it cannot even TIE the best (equal distances go to the lowest face,
            // which is what the exhaustive loop's strict '<' in face order amounts to)
            int best_f = -1;
            int stack[128], sp = 0;
            stack[sp++] = 0;
            while (sp > 0) {
                const BvhNode &n = bvh[stack[--sp]];
                if (box_dist2(n, p) > best) continue;
                if (n.left < 0) {
                    for (int k = n.first; k < n.first + n.count; k++) {
                        const int f = order[k];
                        const V3 c = closest_on_triangle(p, V[F[3 * f]], V[F[3 * f + 1]], V[F[3 * f + 2]]);
                        const double d = sqnorm(p - c);
                        if (d < best || (d == best && f < best_f)) { best = d; closest = c; best_f = f; }
                    }
                } else {
                    const double dl = box_dist2(bvh[n.left], p), dr = box_dist2(bvh[n.right], p);
                    if (dl <= dr) { stack[sp++] = n.right; stack[sp++] = n.left; } else { stack[sp++] = n.left; stack[sp++] = n.right; }
                }
            }
            return best;
        }
        for (int f = 0; f < n_faces(); f++) {
            const V3 c = closest_on_triangle(p, V[F[3 * f]], V[F[3 * f + 1]], V[F[3 * f + 2]]);
            const double d = sqnorm(p - c);
            if (d < best) { best = d; closest = c; }
        }
        return best;
    }

    // igl::fast_winding_number(fwn_bvh, 2.0, p, w) (Shape.cpp:110); fallback: the exact sum in float arithmetic like
    // UT_SolidAngle<float,float>, 2*atan2(det, denom) per triangle
    double winding_number(const V3 &p) const {
        if (hdk) {
            const double q[3] = {p.x, p.y, p.z};
            double w;
            HdkFwn::get().eval(hdk, q, 1, 2.0, &w);
            return w;
        }
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
