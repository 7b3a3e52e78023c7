// Device-side mesh robot shape (Generalshape, src/utils/src/Shape.cpp:105-151):
//   sdf = (1 - 2 w) * dist,  grad = normalize((1 - 2 w) * (p - c)),
// w = igl::fast_winding_number(fwn_bvh, 2.0, p): libigl's HDK UT_SolidAngle<float,float>, order 2 - a 4-wide BVH whose far
// children are replaced by a Taylor expansion about their area-weighted centre.  That approximation moves w by up to ~2e-3,
// so the device walks the SAME hierarchy with the SAME coefficients (built on the host by csrc/fwn_host.hpp, which
// tests/test_fwn_host.py checks node by node against the vendored library) and applies the same descend test
// (|q - P|^2 <= maxP2 * accuracy^2, FastWindingNumberForSoups.h:7119-7258), all in float like the reference.
// c = closest point, dist = |p - c|: exact (igl::AABB::squared_distance is exact, Shape.cpp:112), found by a
// branch-and-bound walk of the same hierarchy using each child's bounding sphere (centre P, radius sqrt(maxP2)), in fp64.
// Both walks keep a per-lane stack of MESH_STACK node numbers; isdf_set_shape refuses meshes whose tree is deeper.
#pragma once
#include "dev_shapes.hpp"

namespace isdf {

// struct DevMesh lives in dev_shapes.hpp

// The closest-point arithmetic of the mesh kind is written with EXPLICIT fused multiply-adds, contraction off: every form of the
// evaluation (per lane, wave-, quad-cooperative, flat) inlines these functions into another context, and where the compiler is free
// to contract a * b + c * d it may fuse one product in one context and the other elsewhere - on an axis-aligned face (the reference's
// mybox.obj) that turns an exact zero into +-1e-17, flips the region test below and moves the closest point by an ulp: the forms
// then disagree in the last bits of the gradient (found by the flat evaluation, round 6).  Written out, every context rounds alike.
__device__ __forceinline__ double mesh_dot(d3 a, d3 b) {
#pragma clang fp contract(off)
    return __builtin_fma(a.x, b.x, __builtin_fma(a.y, b.y, a.z * b.z));
}
// a * b - c * d
__device__ __forceinline__ double mesh_diff2(double a, double b, double c, double d) {
#pragma clang fp contract(off)
    return __builtin_fma(a, b, -(c * d));
}
__device__ __forceinline__ d3 mesh_axpy(double t, d3 v, d3 a) {      // a + t v
#pragma clang fp contract(off)
    return mk3(__builtin_fma(t, v.x, a.x), __builtin_fma(t, v.y, a.y), __builtin_fma(t, v.z, a.z));
}
__device__ __forceinline__ d3 closest_on_triangle(d3 p, d3 a, d3 b, d3 c) {
#pragma clang fp contract(off)
    const d3 ab = b - a, ac = c - a, ap = p - a;
    const double d1 = mesh_dot(ab, ap), d2 = mesh_dot(ac, ap);
    if (d1 <= 0.0 && d2 <= 0.0) return a;
    const d3 bp = p - b;
    const double d3_ = mesh_dot(ab, bp), d4 = mesh_dot(ac, bp);
    if (d3_ >= 0.0 && d4 <= d3_) return b;
    const double vc = mesh_diff2(d1, d4, d3_, d2);
    if (vc <= 0.0 && d1 >= 0.0 && d3_ <= 0.0) return mesh_axpy(d1 / (d1 - d3_), ab, a);
    const d3 cp = p - c;
    const double d5 = mesh_dot(ab, cp), d6 = mesh_dot(ac, cp);
    if (d6 >= 0.0 && d5 <= d6) return c;
    const double vb = mesh_diff2(d5, d2, d1, d6);
    if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) return mesh_axpy(d2 / (d2 - d6), ac, a);
    const double va = mesh_diff2(d3_, d6, d5, d4);
    if (va <= 0.0 && (d4 - d3_) >= 0.0 && (d5 - d6) >= 0.0) return mesh_axpy((d4 - d3_) / ((d4 - d3_) + (d5 - d6)), c - b, b);
    const double denom = 1.0 / (va + vb + vc);
    return mesh_axpy(vc * denom, ac, mesh_axpy(vb * denom, ab, a));
}
// squared distance p - c and the unit gradient s (p - c) / |.|, the same way in every form
__device__ __forceinline__ double mesh_dist2(d3 p, d3 c) { const d3 dd = p - c; return mesh_dot(dd, dd); }
__device__ __forceinline__ d3 mesh_unit(double s, d3 v) {
#pragma clang fp contract(off)
    const d3 a = mk3(s * v.x, s * v.y, s * v.z);
    const double z = mesh_dot(a, a);
    if (z > 0.0) { const double n = sqrt(z); return mk3(a.x / n, a.y / n, a.z / n); }
    return a;
}


// atan2f the way the reference's host computes it.  UTsignedSolidAngleTri ends in SYSatan2(float, float) = ::atan2f of the C
// library (FastWindingNumberForSoups.h:325-326, :6083), and the swept-volume argmin compares SDF values that differ by less than
// one float ulp of w - so WHICH atan2f matters.  glibc's (2.31 of the reference's ROS-noetic platform through 2.35
// of this image: sysdeps/ieee754/flt-32/e_atan2f.c + s_atanf.c, the fdlibm algorithm) is a short chain of IEEE float
// operations: restated here with contraction off it returns the same bits (checked against libm on 2e8 arguments on the host,
// tests/test_fwn_host.py::test_atan2f_restatement_is_glibcs), where the device library's own atan2f differs in the last ulp.
__host__ __device__ __forceinline__ float glibc_atanf(float x) {
#pragma clang fp contract(off)
    const float hi0 = 4.6364760399e-01f, hi1 = 7.8539812565e-01f, hi2 = 9.8279368877e-01f, hi3 = 1.5707962513e+00f;
    const float lo0 = 5.0121582440e-09f, lo1 = 3.7748947079e-08f, lo2 = 3.4473217170e-08f, lo3 = 7.5497894159e-08f;
    const int hx = __builtin_bit_cast(int, x), ix = hx & 0x7fffffff;
    float ahi = 0.f, alo = 0.f;
    int id = -1;
    if (ix >= 0x4c000000) {                               // |x| >= 2^25
        if (ix > 0x7f800000) return x + x;
        return hx > 0 ? hi3 + lo3 : -hi3 - lo3;
    }
    if (ix < 0x3ee00000) {                                // |x| < 0.4375
        if (ix < 0x31000000) return x;                    // |x| < 2^-29
    } else {
        x = __builtin_fabsf(x);
        if (ix < 0x3f980000) {
            if (ix < 0x3f300000) { id = 0; ahi = hi0; alo = lo0; x = (2.0f * x - 1.0f) / (2.0f + x); }
            else { id = 1; ahi = hi1; alo = lo1; x = (x - 1.0f) / (x + 1.0f); }
        } else {
            if (ix < 0x401c0000) { id = 2; ahi = hi2; alo = lo2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
            else { id = 3; ahi = hi3; alo = lo3; x = -1.0f / x; }
        }
    }
    const float z = x * x, w = z * z;
    const float s1 = z * (3.3333334327e-01f + w * (1.4285714924e-01f + w * (9.0908870101e-02f + w * (6.6610731184e-02f + w * (4.9768779427e-02f + w * 1.6285819933e-02f)))));
    const float s2 = w * (-2.0000000298e-01f + w * (-1.1111110449e-01f + w * (-7.6918758452e-02f + w * (-5.8335702866e-02f + w * -3.6531571299e-02f))));
    if (id < 0) return x - x * (s1 + s2);
    const float r = ahi - ((x * (s1 + s2) - alo) - x);
    return hx < 0 ? -r : r;
}
__host__ __device__ __forceinline__ float glibc_atan2f(float y, float x) {
#pragma clang fp contract(off)
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const int hx = __builtin_bit_cast(int, x), ix = hx & 0x7fffffff, hy = __builtin_bit_cast(int, y), iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return glibc_atanf(y);
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) return m < 2 ? y : (m == 2 ? pi + tiny : -pi - tiny);
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) return m == 0 ? pi_o_4 + tiny : (m == 1 ? -pi_o_4 - tiny : (m == 2 ? 3.0f * pi_o_4 + tiny : -3.0f * pi_o_4 - tiny));
        return m == 0 ? 0.0f : (m == 1 ? -0.0f : (m == 2 ? pi + tiny : -pi - tiny));
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = glibc_atanf(__builtin_fabsf(y / x));
    if (m == 0) return z;
    if (m == 1) return __builtin_bit_cast(float, __builtin_bit_cast(int, z) ^ (int)0x80000000);
    if (m == 2) return pi - (z - pi_lo);
    return (z - pi_lo) - pi;
}

// UTsignedSolidAngleTri (FastWindingNumberForSoups.h:6044-6082), float
// Pointers into the hierarchy as the wave-cooperative walks use them: CONSTANT address space - a load whose address is
// wave-uniform then becomes a scalar load (one request for the wavefront, the result in scalar registers) instead of 64
// identical vector loads; the compiler cannot prove that for a plain global pointer in a kernel that also stores.
template <typename T> using mesh_cptr = const T __attribute__((address_space(4))) *;
template <typename T> __device__ __forceinline__ mesh_cptr<T> mesh_const(const T *p) { return (mesh_cptr<T>)(unsigned long long)p; }
// ... and wave-uniform as far as the compiler is concerned (arguments of a device FUNCTION count as divergent)
template <typename T> __device__ __forceinline__ mesh_cptr<T> mesh_const_uni(const T *p) {
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    return (mesh_cptr<T>)(((unsigned long long)hi << 32) | lo);
}

template <typename FP>
__device__ __forceinline__ float mesh_tri_solid_angle(FP u, float qx, float qy, float qz) {
#pragma clang fp contract(off)
    float ax = u[0] - qx, ay = u[1] - qy, az = u[2] - qz;
    float bx = u[3] - qx, by = u[4] - qy, bz = u[5] - qz;
    float cx = u[6] - qx, cy = u[7] - qy, cz = u[8] - qz;
    const float la = sqrtf(ax * ax + ay * ay + az * az), lb = sqrtf(bx * bx + by * by + bz * bz), lc = sqrtf(cx * cx + cy * cy + cz * cz);
    if (la == 0 || lb == 0 || lc == 0) return 0.f;
    // UT_FixedVector::operator/=(T): the reciprocal, then three products (:3405-3417) - not three divisions
    const float ia = 1.0f / la, ib = 1.0f / lb, ic = 1.0f / lc;
    ax *= ia; ay *= ia; az *= ia; bx *= ib; by *= ib; bz *= ib; cx *= ic; cy *= ic; cz *= ic;
    const float ux = bx - ax, uy = by - ay, uz = bz - az, vx = cx - ax, vy = cy - ay, vz = cz - az;
    const float num = ax * (uy * vz - uz * vy) + ay * (uz * vx - ux * vz) + az * (ux * vy - uy * vx);
    if (num == 0) return 0.f;
    const float den = 1.0f + (ax * bx + ay * by + az * bz) + (ax * cx + ay * cy + az * cz) + (bx * cx + by * cy + bz * cz);
    return 2.0f * glibc_atan2f(num, den);
}

// UT_SolidAngle::computeSolidAngle, order 2 (SolidAngleFunctors, :7119-7258); returns the solid angle (w = it / 4 pi).
// The library's traversal (BVH<4>::traverseVectorHelper, :4794-4826) gives every node a VALUE: pre() = the sum of the far-field
// approximations of the children that are far enough (lanes 0..3 in order, zeros for the others), post() adds the values of the
// children it descended into, in lane order - so the float sum has the SHAPE OF THE TREE.  The device keeps that shape (an
// explicit stack of frames: node, lanes still to visit, the two partial sums) - with it, the glibc atan2f above and the same
// coefficients (csrc/fwn_host.hpp, bitwise the library's) w comes out bit for bit the host library's
// (tests/test_fwn_host.py on the host, tests/test_gpu_mesh.py on the device).
template <typename FP, typename IP>
__device__ inline unsigned mesh_node_pre_t(FP r, IP ch, float qx, float qy, float qz, float acc2, float &A) {
#pragma clang fp contract(off)
    unsigned descend = 0;
    float sum = 0.f;
    for (int l = 0; l < 4; l++) {
        if (ch[l] == -1) break;                                   // empty children are last
        float x = qx - r[4 + l], y = qy - r[8 + l], z = qz - r[12 + l];
        const float ql2 = x * x + y * y + z * z;
        float om = 0.f;
        bool use = false;
        if (!(ql2 <= r[l] * acc2)) {
            const float m2 = 1.0f / ql2, m1 = sqrtf(m2);
            x *= m1; y *= m1; z *= m1;
            om = -m2 * (x * r[16 + l] + y * r[20 + l] + z * r[24 + l]);
            const float x2 = x * x, y2 = y * y, z2 = z * z;
            const float m3 = m2 * m1;
            om += m3 * (r[28 + l] + r[32 + l] + r[36 + l]
                        - 3.0f * ((x2 * r[28 + l] + y2 * r[32 + l] + z2 * r[36 + l]) + x * y * r[40 + l] + x * z * r[48 + l] + y * z * r[44 + l]));
            const float x3 = x2 * x, y3 = y2 * y, z3 = z2 * z;
            const float m4 = m2 * m2;
            const float t0x = r[80 + l] + r[84 + l], t0y = r[88 + l] + r[68 + l], t0z = r[72 + l] + r[76 + l];
            const float t1x = y * r[68 + l] + z * r[72 + l], t1y = z * r[76 + l] + x * r[80 + l], t1z = x * r[84 + l] + y * r[88 + l];
            om += m4 * (1.5f * (x * (3.0f * r[52 + l] + t0x) + y * (3.0f * r[56 + l] + t0y) + z * (3.0f * r[60 + l] + t0z))
                        - 7.5f * ((x3 * r[52 + l] + y3 * r[56 + l] + z3 * r[60 + l]) + x * y * z * r[64 + l] + (x2 * t1x + y2 * t1y + z2 * t1z)));
            use = isfinite(om);
        }
        if (!use) { om = 0.f; descend |= 1u << l; }
        sum = l == 0 ? om : sum + om;
    }
    A = sum;
    return descend;
}
__device__ inline unsigned mesh_node_pre(const DevMesh *m, int nodei, float qx, float qy, float qz, float acc2, float &A) {
    return mesh_node_pre_t(m->fwn_box + (size_t)92 * nodei, m->fwn_child + 4 * nodei, qx, qy, qz, acc2, A);
}
__device__ inline float mesh_solid_angle(const DevMesh *m, float qx, float qy, float qz) {
#pragma clang fp contract(off)
    const float acc2 = MESH_FWN_ACCURACY * MESH_FWN_ACCURACY;
    int f_node[MESH_FRAMES];
    unsigned f_bits[MESH_FRAMES];
    float f_A[MESH_FRAMES], f_D[MESH_FRAMES];
    float A;
    unsigned d = mesh_node_pre(m, 0, qx, qy, qz, acc2, A);
    if (!d) return A;
    int sp = 0;
    f_node[0] = 0; f_bits[0] = d; f_A[0] = A; f_D[0] = 0.f;
    for (;;) {
        const unsigned bits = f_bits[sp];
        if (bits == 0) {                                          // every lane of this node visited: post()
            const float v = f_A[sp] + f_D[sp];
            if (sp == 0) return v;
            --sp;
            f_D[sp] += v;
            continue;
        }
        const int l = __ffs(bits) - 1;
        f_bits[sp] = bits & (bits - 1);
        const int ci = m->fwn_child[4 * f_node[sp] + l];
        if (ci >= 0) f_D[sp] += mesh_tri_solid_angle(m->trif + (size_t)9 * ci, qx, qy, qz);
        else {
            const int child = ci & 0x7fffffff;
            float A2;
            const unsigned d2 = mesh_node_pre(m, child, qx, qy, qz, acc2, A2);
            if (!d2) f_D[sp] += A2;
            else { ++sp; f_node[sp] = child; f_bits[sp] = d2; f_A[sp] = A2; f_D[sp] = 0.f; }
        }
    }
}

// per-lane lower bound (squared) of the distance from p to the bounding box of child l of `node` (the child-major record's floats
// 34..39, rounded outwards on the host - the same bound as the quad walks': around the flat faces of a small robot a box has no
// thickness where the winding number's sphere has the face's radius)
template <typename FP>
__device__ __forceinline__ double mesh_child_lb2_box(FP boxq, int node, int l, d3 p) {
    const FP q = boxq + ((size_t)4 * node + l) * MESH_Q_REC + 34;
    const double ex = fmax(fmax((double)q[0] - p.x, p.x - (double)q[3]), 0.0);
    const double ey = fmax(fmax((double)q[1] - p.y, p.y - (double)q[4]), 0.0);
    const double ez = fmax(fmax((double)q[2] - p.z, p.z - (double)q[5]), 0.0);
    return (ex * ex + ey * ey + ez * ez) * 0.999999999999;
}
// squared distance to the closest triangle and that point: branch and bound over the hierarchy's bounding boxes
// `cutoff2`: only triangles closer than sqrt(cutoff2) are of interest; the result is min(exact squared distance, cutoff2)
// and `closest` is written only when a triangle beat the cutoff (the walk prunes against it from the start)
__device__ inline double mesh_closest(const DevMesh *m, d3 p, d3 &closest, double cutoff2 = 1.7976931348623157e308) {
    int stk[MESH_STACK];
    double stk_lb[MESH_STACK];
    int sp = 0;
    double best = cutoff2;
    int bestf = -1;
    stk[0] = 0; stk_lb[0] = 0.0; sp = 1;
    while (sp > 0) {
        --sp;
        if (stk_lb[sp] >= best) continue;
        const int nodei = stk[sp];
        const int *ch = m->fwn_child + 4 * nodei;
        int cid[4]; double clb[4]; int n = 0;
        for (int l = 0; l < 4; l++) {
            const int ci = ch[l];
            if (ci == -1) break;
            const double lb2 = mesh_child_lb2_box(m->fwn_boxq, nodei, l, p);
            if (lb2 >= best) continue;
            if (ci >= 0) {
                const double *t = m->tri + (size_t)9 * ci;
                const d3 c = closest_on_triangle(p, mk3(t[0], t[1], t[2]), mk3(t[3], t[4], t[5]), mk3(t[6], t[7], t[8]));
                const double d = mesh_dist2(p, c);
                if (d < best || (d == best && bestf >= 0 && ci < bestf)) { best = d; closest = c; bestf = ci; }      // equal distances: the lower face (every walk form, whatever its order)
            } else {
                int k = n++;                                       // keep (cid, clb) sorted by decreasing bound
                while (k > 0 && clb[k - 1] < lb2) { clb[k] = clb[k - 1]; cid[k] = cid[k - 1]; --k; }
                clb[k] = lb2; cid[k] = ci & 0x7fffffff;
            }
        }
        for (int k = 0; k < n; k++) if (clb[k] < best) { stk[sp] = cid[k]; stk_lb[sp] = clb[k]; ++sp; }     // nearest child on top
    }
    return best;
}

__device__ inline double mesh_query(const DevMesh *m, d3 p, d3 &closest, double &s_out) {
    const float sa = mesh_solid_angle(m, (float)p.x, (float)p.y, (float)p.z);
    const double w = (double)sa / (4.0 * 3.14159265358979323846);      // libigl: computeSolidAngle(...) / (4.0 * igl::PI)
    s_out = 1. - 2. * w;
    return mesh_closest(m, p, closest);
}

// For the tile term (grad_cost_p): the penalty smoothedL1(cap - sdf) vanishes for sdf >= cap, which is the case for almost
// every voxel of the tile.  Outside the mesh (s = 1 - 2w > 0) sdf = s * dist < cap <=> dist < cap / s, so the closest-point
// walk starts with that cutoff and usually ends at the root.  Returns false when sdf >= cap is proven; otherwise the exact
// sdf and gradient, identical to mesh_sdf_grad.
__device__ inline bool mesh_sdf_grad_below(const DevMesh *m, d3 p, double cap, double &sdf, d3 &g) {
    const float sa = mesh_solid_angle(m, (float)p.x, (float)p.y, (float)p.z);
    const double w = (double)sa / (4.0 * 3.14159265358979323846);
    const double s = 1. - 2. * w;
    double cutoff2 = 1.7976931348623157e308;
    if (s > 0.0) {
        if (!(cap > 0.0)) return false;                 // outside and sdf >= 0 >= cap
        const double dc = cap / s;
        cutoff2 = dc * dc * (1.0 + 1e-12);                 // a hair above: the comparison below errs towards "evaluate"
    }
    d3 c = mk3(0, 0, 0);
    const double d2 = mesh_closest(m, p, c, cutoff2);
    if (d2 >= cutoff2) return false;
    sdf = s * sqrt(d2);
    g = mesh_unit(s, p - c);
    return true;
}

__device__ inline double mesh_sdf(const DevMesh *m, d3 p) {
    d3 c; double s;
    const double d2 = mesh_query(m, p, c, s);
    return s * sqrt(d2);
}
__device__ inline double mesh_sdf_grad(const DevMesh *m, d3 p, d3 &g) {
    d3 c; double s;
    const double d2 = mesh_query(m, p, c, s);
    g = mesh_unit(s, p - c);
    return s * sqrt(d2);
}

// ---------------------------------------------------------------------------------------------------------------------------
// QUAD-COOPERATIVE queries (round 4), for robots of more than a handful of faces.  Per-wavefront time stamps of the swept-volume
// sweep with a 720-face robot (profiles/r4_v1_timing_mesh720_base.txt) showed both launches ending on a TAIL: every workgroup
// resident within 1.5 ms, the median wavefront done after 0.8 ms, the last after 4.9 ms - single descent passes of up to 1.1 ms,
// i.e. one wavefront issuing ~1.6e5 dependent instructions for a point next to the surface (every cluster within twice its
// radius is opened, every triangle's solid angle taken, one after the other).  The hierarchy is 4-wide, so the FOUR CHILDREN of
// a node are now evaluated side by side:
//   lanes 4q .. 4q+3 of a wavefront work on query q (16 queries per wavefront), lane l of the quad on child l of the node -
//   its far-field term / descend test (mesh_quad_pre: ~75 instead of ~300 instructions per node), its triangle's solid angle,
//   its bounding-sphere bound and closest-point-on-triangle; the quad combines through DPP quad_perm moves (no LDS).
// The walk itself stays wave-uniform (node and stack in scalar registers, a node visited when ANY query needs it), and every
// query keeps exactly its own arithmetic:
//   winding number: A = ((om0 + om1) + om2) + om3 over the node's present children and D += value(child) in child order, per
//                   LEVEL - the tree-shaped float sum of the library (mesh_solid_angle), so w stays the library's bit for bit;
//   closest point : branch and bound with one best per LANE (the triangles that lane has seen) and the quad's minimum as the
//                   pruning bound; the minimum over the triangles is exact in any visiting order.
// The swept-volume kernels give one (point, interval) task to a WORKGROUP of four wavefronts and deal the active queries of a
// pass to them (swept_sweep.hip mesh_eval_wg).
// A first version of this walk was LATENCY bound (profiles/r4_v1_timing_mesh720_quad2.txt: 1.3 us per node visit = the child
// word, then the record, then the triangle - three dependent trips to L2 - plus the frames in scratch memory).  So:
//   * ONE trip per node: DevMesh::fwn_boxq holds, per (node, child), 40 floats = the 23 numbers of the record, the child word
//     and - for a triangle child - its nine float coordinates; DevMesh::fwn_triq the fp64 triangle for the closest point.  A lane
//     asks for all of it at once (the addresses depend on (node, l) only);
//   * the frames of the levels above (per quad: A, D, descend mask; per lane: its triangle's solid angle) and the closest-point
//     stack live in LDS (MeshFrames, 12 levels: the reference's 12 000-face Trefoil.obj needs 9).
// MeshFrames is a VIEW of one wavefront's frames: `lv` levels (the hierarchy's depth, so a 720-face robot pays for 6 levels of LDS,
// not for the 12 the deepest supported hierarchy needs - the swept-volume kernels are latency bound and every workgroup more per CU
// counts), 126 lv + 8 words at `w`.
struct MeshFrames {
    unsigned *w; int lv;
    __device__ __forceinline__ unsigned long long &set(int L) const { return reinterpret_cast<unsigned long long *>(w)[L]; }
    __device__ __forceinline__ float &A(int L, int q) const { return reinterpret_cast<float *>(w)[2 * lv + 16 * L + q]; }
    __device__ __forceinline__ float &D(int L, int q) const { return reinterpret_cast<float *>(w)[18 * lv + 16 * L + q]; }
    __device__ __forceinline__ unsigned &dm(int L, int q) const { return w[34 * lv + 16 * L + q]; }
    __device__ __forceinline__ float &T(int L, int lane) const { return reinterpret_cast<float *>(w)[50 * lv + 64 * L + lane]; }
    __device__ __forceinline__ int &cw(int L, int k) const { return reinterpret_cast<int *>(w)[114 * lv + 4 * L + k]; }
    __device__ __forceinline__ int &node(int L) const { return reinterpret_cast<int *>(w)[118 * lv + L]; }
    __device__ __forceinline__ int &cur(int L) const { return reinterpret_cast<int *>(w)[119 * lv + L]; }
    __device__ __forceinline__ int &s_node(int i) const { return reinterpret_cast<int *>(w)[120 * lv + i]; }
    __device__ __forceinline__ int &s_par(int i) const { return reinterpret_cast<int *>(w)[123 * lv + 4 + i]; }
};
__host__ __device__ constexpr int mesh_frames_words(int lv) { return 126 * lv + 8; }      // (even: the next wavefront's `set` stays 8-byte aligned)
__device__ __forceinline__ int mesh_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <int K> __device__ __forceinline__ float quad_bcast(float v) {      // lane K of every quad -> its four lanes
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), K * 0x55, 0xf, 0xf, false));
}
template <int K> __device__ __forceinline__ int quad_bcast(int v) { return __builtin_amdgcn_mov_dpp(v, K * 0x55, 0xf, 0xf, false); }
__device__ __forceinline__ float quad_bcast_dyn(float v, int k) {            // k wave-uniform
    return k == 0 ? quad_bcast<0>(v) : (k == 1 ? quad_bcast<1>(v) : (k == 2 ? quad_bcast<2>(v) : quad_bcast<3>(v)));
}
__device__ __forceinline__ double quad_min(double v) {
    v = fmin(v, dpp_mov_f64<0xB1>(v));      // quad_perm [1,0,3,2]
    return fmin(v, dpp_mov_f64<0x4E>(v));   // quad_perm [2,3,0,1]
}

// pre() of `node` for the quads that are `on`: lane l forms child l's term.  A = the node's far-field sum, dm = the quad's
// descend mask (bit l: child l has to be opened), T = the solid angle of child l when that child is a TRIANGLE to be opened,
// cw = the quad's four child words (cw[k] on every lane).  Called by all 64 lanes.
__device__ __forceinline__ void mesh_quad_pre(const float *boxq, int node, bool on, int l, int qsh,
                                              float qx, float qy, float qz, float acc2, float &A, unsigned &dm, float &T, int (&cw)[4]) {
#pragma clang fp contract(off)
    const float4 *r4 = reinterpret_cast<const float4 *>(boxq + ((size_t)4 * node + l) * MESH_Q_REC);
    // everything this lane can need from this node, requested together (one trip)
    const float4 a0 = r4[0], a1 = r4[1], a2 = r4[2], a3 = r4[3], a4 = r4[4], a5 = r4[5], t0 = r4[6], t1 = r4[7], t2 = r4[8];
    const int ci = __builtin_bit_cast(int, a5.w);
    float om = 0.f;
    bool desc = false;
    if (on && ci != -1) {
        const float rc[24] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w,
                              a3.x, a3.y, a3.z, a3.w, a4.x, a4.y, a4.z, a4.w, a5.x, a5.y, a5.z, 0.f};
#define RC(i) rc[(i) >> 2]                  /* r[i + l] of the node-major record (mesh_node_pre_t) */
        float x = qx - RC(4), y = qy - RC(8), z = qz - RC(12);
        const float ql2 = x * x + y * y + z * z;
        bool use = false;
        if (!(ql2 <= RC(0) * acc2)) {
            const float m2 = 1.0f / ql2, m1 = sqrtf(m2);
            x *= m1; y *= m1; z *= m1;
            om = -m2 * (x * RC(16) + y * RC(20) + z * RC(24));
            const float x2 = x * x, y2 = y * y, z2 = z * z;
            const float m3 = m2 * m1;
            om += m3 * (RC(28) + RC(32) + RC(36)
                        - 3.0f * ((x2 * RC(28) + y2 * RC(32) + z2 * RC(36)) + x * y * RC(40) + x * z * RC(48) + y * z * RC(44)));
            const float x3 = x2 * x, y3 = y2 * y, z3 = z2 * z;
            const float m4 = m2 * m2;
            const float t0x = RC(80) + RC(84), t0y = RC(88) + RC(68), t0z = RC(72) + RC(76);
            const float t1x = y * RC(68) + z * RC(72), t1y = z * RC(76) + x * RC(80), t1z = x * RC(84) + y * RC(88);
            om += m4 * (1.5f * (x * (3.0f * RC(52) + t0x) + y * (3.0f * RC(56) + t0y) + z * (3.0f * RC(60) + t0z))
                        - 7.5f * ((x3 * RC(52) + y3 * RC(56) + z3 * RC(60)) + x * y * z * RC(64) + (x2 * t1x + y2 * t1y + z2 * t1z)));
            use = isfinite(om);
        }
#undef RC
        if (!use) { om = 0.f; desc = true; }
    }
    // the library's lane-order sum over the PRESENT children (empty children are last; an empty node sums to 0)
    const float o0 = quad_bcast<0>(om), o1 = quad_bcast<1>(om), o2 = quad_bcast<2>(om), o3 = quad_bcast<3>(om);
    cw[0] = quad_bcast<0>(ci); cw[1] = quad_bcast<1>(ci); cw[2] = quad_bcast<2>(ci); cw[3] = quad_bcast<3>(ci);
    float sum = o0;
    if (cw[1] != -1) { sum = sum + o1; if (cw[2] != -1) { sum = sum + o2; if (cw[3] != -1) sum = sum + o3; } }
    A = sum;
    dm = (unsigned)((__ballot(desc) >> qsh) & 0xFull);
    T = 0.f;
    if (desc && ci >= 0) {
        const float u[9] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w, t2.x};
        T = mesh_tri_solid_angle(u, qx, qy, qz);
    }
}

// UT_SolidAngle::computeSolidAngle for 16 queries: the four lanes of a quad pass the same (qx, qy, qz, act) and get the same
// result.  Called by all 64 lanes; F = this wavefront's frames in LDS.
__device__ inline float mesh_solid_angle_quad(const DevMesh *m, MeshFrames F, float qx, float qy, float qz, bool act) {
#pragma clang fp contract(off)
    const float acc2 = MESH_FWN_ACCURACY * MESH_FWN_ACCURACY;
    const int lane = (int)__lane_id(), l = lane & 3, qsh = lane & 60, qd = lane >> 2;
    const mesh_cptr<DevMesh> mu = mesh_const_uni(m);
    const float *boxq = mu->fwn_boxq;
    int L = 0, node = 0, cur = 0;
    unsigned long long S = __ballot(act);
    if (!S) return 0.f;
    float A, D = 0.f, T;
    unsigned dm;
    int cw[4];
    mesh_quad_pre(boxq, 0, act, l, qsh, qx, qy, qz, acc2, A, dm, T, cw);
    for (;;) {
        const bool inS = (S >> lane) & 1ull;
        const unsigned mydm = inS ? dm : 0u;
        unsigned long long U = 0ull;
        for (; cur < 4; cur++) { U = __ballot((mydm >> cur) & 1u); if (U) break; }
        if (cur >= 4) {                                           // every child some query opens has been visited: post()
            const float v = A + D;
            if (L == 0) return v;
            --L;
            node = mesh_uni(F.node(L)); cur = mesh_uni(F.cur(L)); S = F.set(L);
            dm = F.dm(L, qd); A = F.A(L, qd); T = F.T(L, lane);
            const float Dp = F.D(L, qd);
            D = inS ? Dp + v : Dp;                                 // (the quads of the child's set are a subset of the parent's)
            cw[0] = F.cw(L, 0); cw[1] = F.cw(L, 1); cw[2] = F.cw(L, 2); cw[3] = F.cw(L, 3);
            continue;
        }
        const int c = cur++;
        const bool inU = (U >> lane) & 1ull;
        const int ci = mesh_uni(c == 0 ? cw[0] : (c == 1 ? cw[1] : (c == 2 ? cw[2] : cw[3])));
        if (ci == -1) { cur = 4; continue; }                      // (empty children are last; never marked by pre)
        if (ci >= 0) {
            const float t = quad_bcast_dyn(T, c);                  // the triangle's solid angle sits on lane c of the quad
            if (inU) D = D + t;
        } else {
            const int child = ci & 0x7fffffff;
            float A2, T2;
            unsigned d2;
            int cw2[4];
            mesh_quad_pre(boxq, child, inU, l, qsh, qx, qy, qz, acc2, A2, d2, T2, cw2);
            if (!__ballot(inU && d2 != 0u)) { if (inU) D = D + A2; }      // nobody goes deeper: the child's value is its pre()
            else {
                if (lane == 0) { F.node(L) = node; F.cur(L) = cur; F.set(L) = S; F.cw(L, 0) = cw[0]; F.cw(L, 1) = cw[1]; F.cw(L, 2) = cw[2]; F.cw(L, 3) = cw[3]; }
                if (l == 0) { F.dm(L, qd) = dm; F.A(L, qd) = A; F.D(L, qd) = D; }
                F.T(L, lane) = T;
                ++L;
                node = child; cur = 0; S = U; dm = d2; A = A2; D = 0.f; T = T2;
                cw[0] = cw2[0]; cw[1] = cw2[1]; cw[2] = cw2[2]; cw[3] = cw2[3];
            }
        }
    }
}

// (Round 4, measured and dropped: SEEDING a walk with the triangle that was closest to the quad's previous query - a valid upper
// bound from the first node on.  The nearest-first descent finds as good a bound within its first few nodes; the seed's own
// triangle test in front of the walk cost more: C5-shaped blob 13.6 -> 15.3 ms, Trefoil.obj 11.8 -> 13.0 ms.)
// Closest point for 16 queries.  `best` in: the query's cutoff (squared), out: min(exact squared distance, cutoff); `closest`
// is written when a triangle beat the cutoff.  The four lanes of a quad pass the same arguments and get the same results.
// `floor2`: a query whose distance (squared) is proven to be AT MOST this has its answer (an inside query that only asks "deeper
// than ...?") and leaves the walk; the value returned for it is then an upper bound <= floor2, not the distance.
__device__ inline double mesh_closest_quad(const DevMesh *m, MeshFrames F, d3 p, bool act, d3 &closest, double best, const double floor2 = -1.0) {
    const int lane = (int)__lane_id(), l = lane & 3, qsh = lane & 60;
    const mesh_cptr<DevMesh> mu = mesh_const_uni(m);
    const float *boxq = mu->fwn_boxq;
    const double *triq = mu->fwn_triq;
    const double cutoff = best;
    double mine = best;                    // the best of the triangles THIS lane has evaluated
    int myf = -1;                          // ... and its face
    d3 myc = mk3(0, 0, 0);
    int sp = __ballot(act) ? 1 : 0;
    if (lane == 0) { F.s_node(0) = 0; F.s_par(0) = -1; }
    while (sp > 0) {
        sp = mesh_uni(sp) - 1;
        const int node = mesh_uni(F.s_node(sp)), par = mesh_uni(F.s_par(sp));
        // this lane's child: bound, child word, fp64 triangle - and the bound of the node itself (from its parent's record), all
        // requested together
        const float *rq = boxq + ((size_t)4 * node + l) * MESH_Q_REC;
        const float2 a0 = *reinterpret_cast<const float2 *>(rq + 34);    // the child's bounding box: lo.x lo.y | lo.z hi.x hi.y hi.z
        const float4 a1 = *reinterpret_cast<const float4 *>(rq + 36);
        const int ci = __builtin_bit_cast(int, rq[23]);
        const double *t = triq + ((size_t)4 * node + l) * MESH_Q_TRI;
        const double2 u0 = *reinterpret_cast<const double2 *>(t), u1 = *reinterpret_cast<const double2 *>(t + 2), u2 = *reinterpret_cast<const double2 *>(t + 4),
                      u3 = *reinterpret_cast<const double2 *>(t + 6), u4 = *reinterpret_cast<const double2 *>(t + 8);
        float2 p0 = a0; float4 p1 = a1;
        if (par >= 0) {                                                   // par = 4 * parent + child index
            p0 = *reinterpret_cast<const float2 *>(boxq + (size_t)par * MESH_Q_REC + 34);
            p1 = *reinterpret_cast<const float4 *>(boxq + (size_t)par * MESH_Q_REC + 36);
        }
        // squared distance from p to the box (round 4: boxes instead of the winding number's bounding spheres - a sphere around a
        // flat patch of the surface contains every query within the patch's radius of it, the box only those within its thickness;
        // no square roots either).  The host rounds the boxes outwards; the factor covers the rounding of this sum.
        auto lb2_of = [&](const float2 &b0, const float4 &b1) {
            const double ex = fmax(fmax((double)b0.x - p.x, p.x - (double)b1.y), 0.0);
            const double ey = fmax(fmax((double)b0.y - p.y, p.y - (double)b1.z), 0.0);
            const double ez = fmax(fmax((double)b1.x - p.z, p.z - (double)b1.w), 0.0);
            return (ex * ex + ey * ey + ez * ez) * 0.999999999999;
        };
        if (par >= 0) {                                           // the bests may have improved since this node was pushed
            if (!__ballot(act && lb2_of(p0, p1) < best)) continue;
        }
        double lb2 = 1.7976931348623157e308;
        if (act && ci != -1) lb2 = lb2_of(a0, a1);
        if (ci >= 0 && lb2 < best) {                              // my child is a triangle that can still beat the query's best
            const d3 c = closest_on_triangle(p, mk3(u0.x, u0.y, u1.x), mk3(u1.y, u2.x, u2.y), mk3(u3.x, u3.y, u4.x));
            const double d = mesh_dist2(p, c);
            if (d < mine || (d == mine && myf >= 0 && ci < myf)) { mine = d; myc = c; myf = ci; }
        }
        best = quad_min(mine);
        if (best <= floor2) act = false;
        // internal children some query still wants, farthest first on the stack (the nearest ends on top); order by the bound of
        // the first wanting query (the queries of a wavefront are neighbours)
        const unsigned long long W = __ballot(ci < -1 && lb2 < best);
        if (!W) continue;
        int c0 = -1, c1 = -1, c2 = -1, c3 = -1;
        int r0 = -1, r1 = -1, r2 = -1;                           // (the bounds as float BITS: non-negative floats order like integers - scalar compares)
        const int lbf = __builtin_bit_cast(int, (float)lb2);      // the visiting order needs no more than a float of the bound
        for (int k = 0; k < 4; k++) {
            const unsigned long long Uk = W & (0x1111111111111111ull << k);
            if (!Uk) continue;
            const int fl = __ffsll((long long)Uk) - 1;
            const int rep = __builtin_amdgcn_readlane(lbf, fl);
            const int cwk = ((__builtin_amdgcn_readlane(ci, fl) & 0x7fffffff) << 2) | k;
            if (rep > r0) { c3 = c2; c2 = c1; r2 = r1; c1 = c0; r1 = r0; c0 = cwk; r0 = rep; }
            else if (rep > r1) { c3 = c2; c2 = c1; r2 = r1; c1 = cwk; r1 = rep; }
            else if (rep > r2) { c3 = c2; c2 = cwk; r2 = rep; }
            else c3 = cwk;
        }
        if (lane == 0) {
            int q = sp;
            if (c0 >= 0) { F.s_node(q) = c0 >> 2; F.s_par(q) = (node << 2) | (c0 & 3); ++q; }
            if (c1 >= 0) { F.s_node(q) = c1 >> 2; F.s_par(q) = (node << 2) | (c1 & 3); ++q; }
            if (c2 >= 0) { F.s_node(q) = c2 >> 2; F.s_par(q) = (node << 2) | (c2 & 3); ++q; }
            if (c3 >= 0) { F.s_node(q) = c3 >> 2; F.s_par(q) = (node << 2) | (c3 & 3); ++q; }
        }
        sp += (c0 >= 0) + (c1 >= 0) + (c2 >= 0) + (c3 >= 0);
    }
    // the quad's winner: the lowest lane holding the minimum hands its point to the other three
    best = quad_min(mine);
    int fkey = (mine == best && myf >= 0) ? myf : 0x7fffffff;                      // equal distances: the lower face
    fkey = min(fkey, __builtin_amdgcn_mov_dpp(fkey, 0xB1, 0xf, 0xf, false));
    fkey = min(fkey, __builtin_amdgcn_mov_dpp(fkey, 0x4E, 0xf, 0xf, false));
    const unsigned hit = (unsigned)((__ballot(mine == best && myf == fkey) >> qsh) & 0xFull);
    const int src = qsh + (hit ? __ffs((int)hit) - 1 : 0);
    const double wx = __shfl(myc.x, src, 64), wy = __shfl(myc.y, src, 64), wz = __shfl(myc.z, src, 64);
    if (best < cutoff) closest = mk3(wx, wy, wz);
    return best;
}

// "sdf(p) < cap?" answered NO from the distance lattice alone (DevMesh::dl: distances at the nodes, rounded down; h = the farthest a
// point is from its nearest node), without a walk.  sdf = (1 - 2 w) * dist with |1 - 2 w| <= 1.05 (what the approximate winding
// number of a closed mesh may add, as in the tile sweep's band) and dist 1-Lipschitz:
//   cap > 0: a node outside, farther from the surface than h, has p on its side of it and dist(p) >= d_node - h;
//   cap < 0: "deeper than |cap|?" needs dist(p) > |cap| / 1.05, and dist(p) <= d_node + h wherever the node lies.
// false = not decided (no lattice, p outside it, an uncapped query).
__device__ __forceinline__ bool mesh_lattice_not_below(const DevMesh *m, d3 p, double cap) {
    if (!(cap < 1.0e300) || cap == 0.0) return false;
    const mesh_cptr<DevMesh> ml = mesh_const_uni(m);
    const float *dl = ml->dl;
    if (!dl) return false;
    const int n0 = ml->dln[0], n1 = ml->dln[1], n2 = ml->dln[2];
    const float inv = ml->dl_inv, fx = ((float)p.x - ml->dl_min[0]) * inv, fy = ((float)p.y - ml->dl_min[1]) * inv, fz = ((float)p.z - ml->dl_min[2]) * inv;
    if (!(fx > -0.5f && fx < (float)n0 - 0.5f && fy > -0.5f && fy < (float)n1 - 0.5f && fz > -0.5f && fz < (float)n2 - 0.5f)) return false;
    const float v = dl[((size_t)(int)rintf(fx) * n1 + (int)rintf(fy)) * n2 + (int)rintf(fz)];
    const double h = (double)ml->dl_h;
    const bool node_outside = __float_as_int(v) >= 0;                                   // (the sign bit: inside)
    // (a mesh with defect pockets - DevMesh::dl_tau, dl_slack: within dl_tau of the surface a point may sit in a pocket where |sdf| <= dl_slack
    // whatever the winding number does there.  cap > 0: the point must be farther from the surface than any pocket reaches, or the
    // pocket's values must lie above the cap as well - they need not; cap < 0: a pocket's value is no deeper than -dl_slack.)
    if (cap > 0.0) return node_outside && ((double)v - h) * 0.952 >= cap * (1.0 + 1e-9) && ((double)v - h) > (double)ml->dl_tau;
    const double dn = (double)fabsf(v) * (1.0 + 4.0e-6);                                // an upper bound of the node's distance
    return 1.05 * (dn + h) <= -cap * (1.0 - 1e-9) && (double)ml->dl_slack <= -cap * (1.0 - 1e-9);
}

// lb <= sdf(p) <= ub from the lattice under the same two assumptions (false: nothing known - no lattice, or p outside it)
__device__ __forceinline__ bool mesh_lattice_bounds(const DevMesh *m, d3 p, double &lb, double &ub) {
    lb = -1.7976931348623157e308; ub = 1.7976931348623157e308;
    const mesh_cptr<DevMesh> ml = mesh_const_uni(m);
    const float *dl = ml->dl;
    if (!dl) return false;
    const int n0 = ml->dln[0], n1 = ml->dln[1], n2 = ml->dln[2];
    const float inv = ml->dl_inv, fx = ((float)p.x - ml->dl_min[0]) * inv, fy = ((float)p.y - ml->dl_min[1]) * inv, fz = ((float)p.z - ml->dl_min[2]) * inv;
    if (!(fx > -0.5f && fx < (float)n0 - 0.5f && fy > -0.5f && fy < (float)n1 - 0.5f && fz > -0.5f && fz < (float)n2 - 0.5f)) return false;
    const float v = dl[((size_t)(int)rintf(fx) * n1 + (int)rintf(fy)) * n2 + (int)rintf(fz)];
    const double h = (double)ml->dl_h;
    const bool node_outside = __float_as_int(v) >= 0;                                   // (the sign bit: inside)
    const double dlo = (double)fabsf(v), dhi = dlo * (1.0 + 4.0e-6);                    // the node's distance, from below / above
    const double far = fmax(dlo - h, 0.0), near = dhi + h;                                            // dist(p) lies in [far, near]
    if (dlo > h) {                      // p on the node's side of the surface
        if (node_outside) { lb = 0.952 * far; ub = 1.05 * near; }
        else { lb = -1.05 * near; ub = -0.952 * far; }
    } else { lb = -1.05 * near; ub = 1.05 * near; }
    // defect pockets (DevMesh::dl_tau, dl_slack): a point within dl_tau of the surface may sit in one, with |sdf| <= dl_slack
    if (!(far > (double)ml->dl_tau)) { lb = fmin(lb, -(double)ml->dl_slack); ub = fmax(ub, (double)ml->dl_slack); }
    return true;
}

// mesh_sdf_grad / mesh_sdf_grad_below for 16 queries (all 64 lanes call; a quad passes the same arguments); cap = +infinity
// gives the plain query.  Returns false when sdf >= cap is proven (or the query is not `act`).
__device__ inline bool mesh_sdf_grad_below_quad(const DevMesh *m, MeshFrames F, d3 p, bool act, double cap, double &sdf, d3 &g) {
    float sa = mesh_solid_angle_quad(m, F, (float)p.x, (float)p.y, (float)p.z, act);
#if ISDF_MESH_DUP == 1
    { float sb = mesh_solid_angle_quad(m, F, (float)p.x + 1e-30f, (float)p.y, (float)p.z, act); asm volatile("" : "+v"(sb)); if (sb == 12345.f) sa = sb; }
#endif
    const double w = (double)sa / (4.0 * 3.14159265358979323846);
    const double s = 1. - 2. * w;
    double cutoff2 = 1.7976931348623157e308, floor2 = -1.0;
    if (s > 0.0 && cap < 1.0e300) {
        if (!(cap > 0.0)) act = false;                  // outside and sdf >= 0 >= cap
        else { const double dc = cap / s; cutoff2 = dc * dc * (1.0 + 1e-12); }
    } else if (s < 0.0 && cap < 0.0) {
        // inside, and the question is "deeper than cap?": sdf = s * dist < cap <=> dist > cap / s - a triangle within that distance
        // answers no (the swept-volume scans ask it of every sample against the running minimum); a hair below: errs towards "evaluate"
        const double dc = cap / s;
        floor2 = dc * dc * (1.0 - 1e-12);
    }
    d3 c = mk3(0, 0, 0);
    double d2 = mesh_closest_quad(m, F, p, act, c, cutoff2, floor2);
#if ISDF_MESH_DUP == 2
    { d3 c2 = mk3(0, 0, 0); double e2 = mesh_closest_quad(m, F, mk3(p.x + 1e-300, p.y, p.z), act, c2, cutoff2, floor2); asm volatile("" : "+v"(e2)); if (e2 == -1.0) d2 = e2; }
#endif
    if (!act || d2 >= cutoff2 || d2 <= floor2) return false;
    sdf = s * sqrt(d2);
    g = mesh_unit(s, p - c);
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------------
// FLAT evaluation (round 6) for SMALL meshes - the reference's own robots are 12- to 52-face meshes (mybox.obj, Lthick.obj, drone.obj).
// A walk pays per NODE VISIT: ~770 vector + ~480 scalar instructions of wave-uniform control around the four children of one node,
// whatever the mesh - 12 nodes for the 20 faces of Lthick.obj, 9 000 instructions per round.  A hierarchy this small fits a table: its
// (node, child) slots in level order (DevMesh::flat, built by isdf_set_shape).  Lanes 4q .. 4q+3 still work on query q, but sweep a whole
// LEVEL's slots side by side, four at a time - no frames, no stack, no scalar control per node:
//   winding number: level by level, every reachable slot's far-field term / descend test (and its triangle's solid angle when a
//                   triangle is opened); a query's opened slots are one 64-bit mask in a register.  The values are then combined bottom
//                   up in the hierarchy's own order - A = ((om0 + om1) + om2) + om3 over a node's present children (0 for an opened one),
//                   D = 0 + value(child) ... in child order over the opened ones, value = opened children ? A + D : A - so w is the
//                   library's bit for bit, like every other form (mesh_solid_angle).
//   closest point : every triangle whose box can still beat the query's best, exact in any order, equal distances to the lower face.
// om: this wavefront's [16 queries][MESH_FLAT_SLOTS] floats in LDS (a quad's lanes exchange slot values through it).
__device__ __forceinline__ void flat_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// one slot's term: the far-field approximation of the (node, child) whose record is rq, seen from q, or - when the child has to be opened - 0,
// respectively its solid angle if it is a triangle; returns "has to be opened".  The arithmetic of mesh_quad_pre, contraction off.
__device__ __forceinline__ bool mesh_slot_term(const float *rq, bool tri, float qx, float qy, float qz, float acc2, float &omv) {
#pragma clang fp contract(off)
    const float4 *r4 = reinterpret_cast<const float4 *>(rq);
    const float4 a0 = r4[0], a1 = r4[1], a2 = r4[2], a3 = r4[3], a4 = r4[4], a5 = r4[5];
    const float rc[24] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w,
                          a3.x, a3.y, a3.z, a3.w, a4.x, a4.y, a4.z, a4.w, a5.x, a5.y, a5.z, 0.f};
#define RC(i) rc[(i) >> 2]                  /* r[i + l] of the node-major record (mesh_node_pre_t) */
    float x = qx - RC(4), y = qy - RC(8), z = qz - RC(12);
    const float ql2 = x * x + y * y + z * z;
    bool use = false;
    omv = 0.f;
    if (!(ql2 <= RC(0) * acc2)) {
        const float m2 = 1.0f / ql2, m1 = sqrtf(m2);
        x *= m1; y *= m1; z *= m1;
        omv = -m2 * (x * RC(16) + y * RC(20) + z * RC(24));
        const float x2 = x * x, y2 = y * y, z2 = z * z;
        const float m3 = m2 * m1;
        omv += m3 * (RC(28) + RC(32) + RC(36)
                     - 3.0f * ((x2 * RC(28) + y2 * RC(32) + z2 * RC(36)) + x * y * RC(40) + x * z * RC(48) + y * z * RC(44)));
        const float x3 = x2 * x, y3 = y2 * y, z3 = z2 * z;
        const float m4 = m2 * m2;
        const float t0x = RC(80) + RC(84), t0y = RC(88) + RC(68), t0z = RC(72) + RC(76);
        const float t1x = y * RC(68) + z * RC(72), t1y = z * RC(76) + x * RC(80), t1z = x * RC(84) + y * RC(88);
        omv += m4 * (1.5f * (x * (3.0f * RC(52) + t0x) + y * (3.0f * RC(56) + t0y) + z * (3.0f * RC(60) + t0z))
                     - 7.5f * ((x3 * RC(52) + y3 * RC(56) + z3 * RC(60)) + x * y * z * RC(64) + (x2 * t1x + y2 * t1y + z2 * t1z)));
        use = isfinite(omv);
    }
#undef RC
    if (use) return false;
    omv = 0.f;
    if (tri) {                                  // an opened TRIANGLE: its solid angle is the slot's value
        const float4 t0 = r4[6], t1 = r4[7];
        const float u[9] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w, rq[32]};
        omv = mesh_tri_solid_angle(u, qx, qy, qz);
    }
    return true;
}

// F: the workgroup's copy of DevMesh::flat in LDS (tables, records, fp64 triangles) and this wavefront's [16][nsp] slot values
struct FlatLds { const int *tab; const float *rec; const double *trec; float *om; int n_slots, n_levels, nsp; };
// (the blob describes itself: header ints [19] slots, [20] levels, [22] / [23] where the records / the fp64 triangles begin)
__device__ __forceinline__ FlatLds flat_lds_view(int *blob, float *om) {
    const int ns = mesh_uni(blob[19]);
    return FlatLds{blob, reinterpret_cast<const float *>(blob + mesh_uni(blob[22])), reinterpret_cast<const double *>(blob + mesh_uni(blob[23])), om, ns, mesh_uni(blob[20]), (ns + 3) & ~3};
}
__host__ __device__ constexpr int flat_om_floats(int n_slots) { return 16 * ((n_slots + 3) & ~3); }
// all threads of the workgroup: the blob into LDS (16 bytes per thread and round)
__device__ __forceinline__ void flat_lds_stage(const DevMesh *m, int *blob) {
    const mesh_cptr<DevMesh> mu = mesh_const_uni(m);
    const int4 *src = reinterpret_cast<const int4 *>(mu->flat);
    const int n4 = mu->flat_words >> 2;
    for (int i = threadIdx.x; i < n4; i += blockDim.x) reinterpret_cast<int4 *>(blob)[i] = src[i];
    __syncthreads();
}
__device__ inline bool mesh_sdf_grad_below_flat(const FlatLds &F, d3 p, bool act, double cap, double &sdf, d3 &g) {
    const float acc2 = MESH_FWN_ACCURACY * MESH_FWN_ACCURACY;
    const int lane = (int)__lane_id(), j = lane & 3, qsh = lane & 60, qd = lane >> 2;
    if (!__ballot(act)) return false;
    const int *fl = F.tab;
    const int n_slots = F.n_slots, n_levels = F.n_levels, n_tris = fl[18];
    const int *slots = fl + 24, *tris = slots + 4 * n_slots, *nodes = tris + n_tris;
    float *const om = F.om + qd * F.nsp;
    const float qx = (float)p.x, qy = (float)p.y, qz = (float)p.z;
    // ---- far-field terms, level by level
    unsigned long long opened = 0ull;              // bit s: slot s is reachable AND has to be opened (the same in the four lanes of a quad)
    for (int l = 0; l < n_levels; l++) {
        const int b = fl[l], e = fl[l + 1];
        for (int s0 = b; s0 < e; s0 += 4) {
            const int sl = s0 + j;
            const bool valid = sl < e;
            const int4 si = *reinterpret_cast<const int4 *>(slots + 4 * (valid ? sl : b));      // record, parent slot, triangle, level
            const bool reach = act && valid && (si.y < 0 || ((opened >> si.y) & 1ull));
            bool desc = false;
            if (__ballot(reach)) {
                if (reach) {
                    float omv;
                    desc = mesh_slot_term(F.rec + sl * MESH_Q_REC, si.z >= 0, qx, qy, qz, acc2, omv);
                    om[sl] = omv;
                }
            }
            opened |= ((__ballot(desc) >> qsh) & 0xFull) << s0;
        }
        flat_lds_sync();
    }
    // ---- the values, bottom up (a step = the nodes of one level; lane j of the quad takes every fourth of them)
    float w_root = 0.f;
    for (int st = 0; st < n_levels; st++) {
        const int b = fl[9 + st], e = fl[9 + st + 1];
        for (int n0 = b; n0 < e; n0 += 4) {
            const int nd = n0 + j;
            const bool valid = nd < e;
            const int *ni = nodes + 5 * (valid ? nd : b);
            const int own = ni[0];
            if (act && valid && (own < 0 || ((opened >> own) & 1ull))) {
                float A = 0.f, D = 0.f;
                bool any = false;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const int cs = ni[1 + c];
                    if (cs < 0) break;                                    // (empty children are last)
                    const bool dsc = (opened >> cs) & 1ull;
                    const float v = om[cs];
                    const float a = dsc ? 0.f : v;
                    A = c == 0 ? a : A + a;
                    if (dsc) { D = D + v; any = true; }
                }
                const float val = any ? A + D : A;
                if (own >= 0) om[own] = val; else w_root = val;
            }
        }
        flat_lds_sync();
    }
    // (the root is the only node of the last step: lane 0 of the quad holds its value)
    const float sa = quad_bcast<0>(w_root);
    const double w = (double)sa / (4.0 * 3.14159265358979323846);
    const double s = 1. - 2. * w;
    double cutoff2 = 1.7976931348623157e308, floor2 = -1.0;
    if (s > 0.0 && cap < 1.0e300) {
        if (!(cap > 0.0)) act = false;                  // outside and sdf >= 0 >= cap
        else { const double dc = cap / s; cutoff2 = dc * dc * (1.0 + 1e-12); }
    } else if (s < 0.0 && cap < 0.0) {
        const double dc = cap / s;
        floor2 = dc * dc * (1.0 - 1e-12);
    }
    // ---- closest point: lane j takes every fourth triangle
    double mine = cutoff2, best = cutoff2;
    int myf = -1;
    d3 myc = mk3(0, 0, 0);
    if (__ballot(act)) {
        for (int k0 = 0; k0 < n_tris; k0 += 4) {
            const int k = k0 + j;
            const bool valid = act && k < n_tris;
            const int sl = tris[valid ? k : 0];
            const int4 si = *reinterpret_cast<const int4 *>(slots + 4 * sl);
            const float *rq = F.rec + sl * MESH_Q_REC;
            const float2 b0 = *reinterpret_cast<const float2 *>(rq + 34);
            const float4 b1 = *reinterpret_cast<const float4 *>(rq + 36);
            const double ex = fmax(fmax((double)b0.x - p.x, p.x - (double)b1.y), 0.0);
            const double ey = fmax(fmax((double)b0.y - p.y, p.y - (double)b1.z), 0.0);
            const double ez = fmax(fmax((double)b1.x - p.z, p.z - (double)b1.w), 0.0);
            const double lb2 = (ex * ex + ey * ey + ez * ez) * 0.999999999999;
            if (valid && lb2 < best) {
                const double *t = F.trec + sl * MESH_Q_TRI;
                const d3 c = closest_on_triangle(p, mk3(t[0], t[1], t[2]), mk3(t[3], t[4], t[5]), mk3(t[6], t[7], t[8]));
                const double d = mesh_dist2(p, c);
                if (d < mine || (d == mine && myf >= 0 && si.z < myf)) { mine = d; myc = c; myf = si.z; }
            }
            best = quad_min(mine);
        }
    }
    best = quad_min(mine);
    int fkey = (mine == best && myf >= 0) ? myf : 0x7fffffff;                      // equal distances: the lower face
    fkey = min(fkey, __builtin_amdgcn_mov_dpp(fkey, 0xB1, 0xf, 0xf, false));
    fkey = min(fkey, __builtin_amdgcn_mov_dpp(fkey, 0x4E, 0xf, 0xf, false));
    const unsigned hit = (unsigned)((__ballot(mine == best && myf == fkey) >> qsh) & 0xFull);
    const int src = qsh + (hit ? __ffs((int)hit) - 1 : 0);
    const double wx = __shfl(myc.x, src, 64), wy = __shfl(myc.y, src, 64), wz = __shfl(myc.z, src, 64);
    const d3 c = mk3(wx, wy, wz);
    if (!act || best >= cutoff2 || best <= floor2) return false;
    sdf = s * sqrt(best);
    g = mesh_unit(s, p - c);
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------------
// WAVE-COOPERATIVE queries (round 3; small robots and hierarchies deeper than MESH_Q_LEVELS keep them).  The per-lane walks above make every lane chase its own node: 64 different 368-byte records per
// step of the loop and a private stack per lane - measured on the swept-volume sweep with a 720-face robot: 1.5e8 vector memory
// instructions per launch, 53-70 % of the wave cycles waiting, VALU 15 % busy (profiles/r3b_v1mesh_pmc_*).  But the lanes of a
// wavefront ask about NEIGHBOURING points (one obstacle point seen from poses a few centimetres apart in the descent; the
// voxels around one pose in the tile sweep), so the wavefront walks the hierarchy ONCE: the node and its stack are wave-uniform
// (records and triangles arrive through scalar loads), a node is visited when ANY lane needs it, and every lane keeps doing
// exactly its own arithmetic -
//   winding number: a lane that is far enough from a child takes that child's far-field term, the others descend; the lane's
//                   partial sums live per LEVEL (the tree-shaped sum of the library, see mesh_solid_angle), so w stays the
//                   library's bit for bit;
//   closest point : branch and bound with per-lane best distances; a subtree is skipped when NO lane can improve in it, so
//                   every lane still sees every triangle that could be its closest (the minimum is exact either way).
// All lanes that execute the call take part (ballot): it may sit inside divergent code, as long as those lanes reach it together.

__device__ inline float mesh_solid_angle_wave(const DevMesh *m, float qx, float qy, float qz) {
#pragma clang fp contract(off)
    const float acc2 = MESH_FWN_ACCURACY * MESH_FWN_ACCURACY;
    // frames of the levels ABOVE the current one (written on the way down, read back on the way up; the index is wave-uniform,
    // so these private arrays are one coalesced access each); the current level lives in registers
    int f_node[MESH_FRAMES], f_cur[MESH_FRAMES];
    unsigned long long f_set[MESH_FRAMES];
    unsigned f_dm[MESH_FRAMES];
    float f_A[MESH_FRAMES], f_D[MESH_FRAMES];
    const int lane = (int)__lane_id();
    // the hierarchy's arrays, fetched ONCE (m itself lives in global memory: every m->field is a load)
    const mesh_cptr<DevMesh> mu = mesh_const_uni(m);
    const mesh_cptr<float> box = mesh_const_uni(mu->fwn_box), trif = mesh_const_uni(mu->trif);
    const mesh_cptr<int> chl = mesh_const_uni(mu->fwn_child);
    int L = 0, node = 0, cur = 0;
    unsigned long long S = __ballot(1);
    float A, D = 0.f;
    unsigned dm = mesh_node_pre_t(box, chl, qx, qy, qz, acc2, A);
    for (;;) {
        const bool inS = (S >> lane) & 1ull;
        const unsigned mydm = inS ? dm : 0u;
        unsigned long long U = 0ull;
        for (; cur < 4; cur++) { U = __ballot((mydm >> cur) & 1u); if (U) break; }
        if (cur >= 4) {                                           // every child some lane descends into has been visited: post()
            const float v = A + D;
            if (L == 0) return v;
            --L;
            node = mesh_uni(f_node[L]); cur = mesh_uni(f_cur[L]); S = f_set[L]; dm = f_dm[L]; A = f_A[L];
            D = inS ? f_D[L] + v : f_D[L];                         // (the lanes of the child's set are a subset of the parent's)
            continue;
        }
        const int l = cur++;
        const bool inU = (U >> lane) & 1ull;
        const int ci = mesh_uni(chl[4 * node + l]);
        if (ci == -1) { cur = 4; continue; }                      // (empty children are last; never marked by pre)
        if (ci >= 0) {
            if (inU) D = D + mesh_tri_solid_angle(trif + (size_t)9 * ci, qx, qy, qz);
        } else {
            const int child = ci & 0x7fffffff;
            float A2 = 0.f;
            unsigned d2 = 0u;
            if (inU) d2 = mesh_node_pre_t(box + (size_t)92 * child, chl + 4 * child, qx, qy, qz, acc2, A2);
            if (!__ballot(d2 != 0u)) { if (inU) D = D + A2; }      // nobody goes deeper: the child's value is its pre()
            else {
                f_node[L] = node; f_cur[L] = cur; f_set[L] = S; f_dm[L] = dm; f_A[L] = A; f_D[L] = D;
                ++L;
                node = child; cur = 0; S = U; dm = d2; A = A2; D = 0.f;
            }
        }
    }
}

// per-lane lower bound (squared) of the distance from p to child l of the node whose record is r
template <typename FP>
__device__ __forceinline__ double mesh_child_lb2(FP r, int l, d3 p) {
    const double px = r[4 + l], py = r[8 + l], pz = r[12 + l];
    const double dx = p.x - px, dy = p.y - py, dz = p.z - pz;
    const double rad = (double)sqrtf(r[l]) * 1.00001 + 1e-6 * (fabs(px) + fabs(py) + fabs(pz) + 1.0);
    const double lb = fmax(sqrt(dx * dx + dy * dy + dz * dz) - rad, 0.0);
    return lb * lb;
}
// `best` in: the lane's cutoff (squared), out: min(exact squared distance, cutoff); `closest` written when a triangle beat it
__device__ inline double mesh_closest_wave(const DevMesh *m, d3 p, bool act, d3 &closest, double best) {
    int s_node[MESH_STACK], s_par[MESH_STACK];
    const mesh_cptr<DevMesh> mu = mesh_const_uni(m);
    const mesh_cptr<float> boxq = mesh_const_uni(mu->fwn_boxq);
    const mesh_cptr<int> chl = mesh_const_uni(mu->fwn_child);
    const mesh_cptr<double> tri = mesh_const_uni(mu->tri);
    int bestf = -1;
    int sp = 1;
    s_node[0] = 0; s_par[0] = -1;
    while (sp > 0) {
        sp = mesh_uni(sp) - 1;
        const int node = mesh_uni(s_node[sp]), par = mesh_uni(s_par[sp]);
        if (par >= 0) {                                           // the bests may have improved since this node was pushed
            const double lb2 = mesh_child_lb2_box(boxq, par >> 2, par & 3, p);
            if (!__ballot(act && lb2 < best)) continue;
        }
        const mesh_cptr<int> ch = chl + 4 * node;
        // (up to four internal children, kept sorted by decreasing bound in NAMED scalars: a dynamically indexed private
        // array would live in scratch memory)
        int c0 = -1, c1 = -1, c2 = -1, c3 = -1;
        int r0 = -1, r1 = -1, r2 = -1;
        for (int l = 0; l < 4; l++) {
            const int ci = mesh_uni(ch[l]);
            if (ci == -1) break;
            const double lb2 = mesh_child_lb2_box(boxq, node, l, p);
            const bool want = act && lb2 < best;
            const unsigned long long U = __ballot(want);
            if (!U) continue;
            if (ci >= 0) {
                if (want) {
                    const mesh_cptr<double> t = tri + (size_t)9 * ci;
                    const d3 c = closest_on_triangle(p, mk3(t[0], t[1], t[2]), mk3(t[3], t[4], t[5]), mk3(t[6], t[7], t[8]));
                    const double d = mesh_dist2(p, c);
                    if (d < best || (d == best && bestf >= 0 && ci < bestf)) { best = d; closest = c; bestf = ci; }      // equal distances: the lower face (every walk form, whatever its order)
                }
            } else {
                // visiting order: by the bound of the first lane that wants the child (the lanes are neighbours); rep >= 0
                const int fl = __ffsll((long long)U) - 1;
                const int rep = __builtin_amdgcn_readlane(__builtin_bit_cast(int, (float)lb2), fl);      // float bits of a non-negative bound: ordered like integers
                const int cw = ((ci & 0x7fffffff) << 2) | l;
                if (rep > r0) { c3 = c2; c2 = c1; r2 = r1; c1 = c0; r1 = r0; c0 = cw; r0 = rep; }
                else if (rep > r1) { c3 = c2; c2 = c1; r2 = r1; c1 = cw; r1 = rep; }
                else if (rep > r2) { c3 = c2; c2 = cw; r2 = rep; }
                else c3 = cw;
            }
        }
        if (c0 >= 0) { s_node[sp] = c0 >> 2; s_par[sp] = (node << 2) | (c0 & 3); ++sp; }      // farthest first: the nearest child ends on top
        if (c1 >= 0) { s_node[sp] = c1 >> 2; s_par[sp] = (node << 2) | (c1 & 3); ++sp; }
        if (c2 >= 0) { s_node[sp] = c2 >> 2; s_par[sp] = (node << 2) | (c2 & 3); ++sp; }
        if (c3 >= 0) { s_node[sp] = c3 >> 2; s_par[sp] = (node << 2) | (c3 & 3); ++sp; }
    }
    return best;
}

// mesh_sdf_grad / mesh_sdf_grad_below for all executing lanes together
__device__ inline double mesh_sdf_grad_wave(const DevMesh *m, d3 p, d3 &g) {
    const float sa = mesh_solid_angle_wave(m, (float)p.x, (float)p.y, (float)p.z);
    const double w = (double)sa / (4.0 * 3.14159265358979323846);
    const double s = 1. - 2. * w;
    d3 c = mk3(0, 0, 0);
    const double d2 = mesh_closest_wave(m, p, true, c, 1.7976931348623157e308);
    g = mesh_unit(s, p - c);
    return s * sqrt(d2);
}
__device__ inline bool mesh_sdf_grad_below_wave(const DevMesh *m, d3 p, double cap, double &sdf, d3 &g) {
    if (mesh_lattice_not_below(m, p, cap)) return false;       // (answered by the distance lattice: this lane stays out of the walks)
    const float sa = mesh_solid_angle_wave(m, (float)p.x, (float)p.y, (float)p.z);
    const double w = (double)sa / (4.0 * 3.14159265358979323846);
    const double s = 1. - 2. * w;
    double cutoff2 = 1.7976931348623157e308;
    bool act = true;
    if (s > 0.0) {
        if (!(cap > 0.0)) act = false;                  // outside and sdf >= 0 >= cap
        else { const double dc = cap / s; cutoff2 = dc * dc * (1.0 + 1e-12); }
    }
    d3 c = mk3(0, 0, 0);
    const double d2 = mesh_closest_wave(m, p, act, c, cutoff2);
    if (!act || d2 >= cutoff2) return false;
    sdf = s * sqrt(d2);
    g = mesh_unit(s, p - c);
    return true;
}

} // namespace isdf
