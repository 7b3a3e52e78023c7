// The robot-shape plugin by itself: BasicShape::getonlySDF / getonlyGrad1 / getSDFwithGrad1 (Shape.hpp:32-88 for the
// analytic classes, Shape.cpp:105-151 for the mesh Generalshape) on a batch of BODY-FRAME points, one lane per point.
// The sweeps inline the same device functions (dev_shapes.hpp / dev_mesh.hpp); this entry point exists so a caller (and
// the parity tests) can query the installed shape directly, like the reference's planners and visualisers do.
#include "isdf_ctx.hpp"
#include "dev_mesh.hpp"
#include <cstring>

namespace isdf {

template <int KIND>
__global__ __launch_bounds__(256) void shape_eval_kernel(DevShape S, const double *__restrict__ pts, int n, double *__restrict__ sdf_out, double *__restrict__ grad_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const d3 p = mk3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
    if (sdf_out) sdf_out[i] = shape_sdf<KIND>(S, p);
    if (grad_out) {
        const d3 g = shape_grad<KIND>(S, p);
        grad_out[3 * i] = g.x; grad_out[3 * i + 1] = g.y; grad_out[3 * i + 2] = g.z;
    }
}

// ---- What the distance lattice's users assume about a mesh, CHECKED before one is built (Shape.cpp:105-151).
// sdf = (1 - 2 w) * dist tracks the distance only where the winding number of the surface is 0 outside and 1 inside.  That is a
// property of the surface, tested here with the EXACT winding number (the sum of all triangles' solid angles in fp64, no hierarchy):
// a hair off every face, on both sides, at the centroid and (small meshes) next to the three corners, w must be 0 or 1 (+- 0.01).
// Every region of space the surface bounds touches a face, so a region where w is 2 (a closed sheet inside another one with the
// same orientation, overlapping bodies), -1 (an inverted body) or a fraction (an open or torn surface) shows up at the faces that
// bound it - whatever its thickness; a thin-walled hollow shell (outer sheet outwards, inner sheet inwards) passes, as it should.
// Real meshes carry DEFECTS, though: the reference's own Trefoil.obj has 18 folded sliver faces (of 12 000) with a pocket of w = -1
// a few micrometres thick behind them.  A sample that fails is therefore probed further out along the normal, doubling the offset: if
// w is 0 / 1 again within tau_limit (a tenth of the lattice's node-to-point reach) the pocket is a defect of known thickness - the
// largest such offset comes back in out[1] (float bits), the largest |1 - 2 w| seen inside a pocket in out[2] - and the lattice's
// users widen their brackets by what a pocket that thin can do (dev_mesh.hpp: dl_tau, dl_slack).  If it is not, out[0] bit 0 is
// raised: the mesh gets no lattice.
__device__ __forceinline__ d3 cross3(d3 a, d3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ double tri_solid_angle_f64(d3 a, d3 b, d3 c) {      // Van Oosterom - Strackee, about the origin
    const double la = sqrt(dot3(a, a)), lb = sqrt(dot3(b, b)), lc = sqrt(dot3(c, c));
    const double det = a.x * (b.y * c.z - b.z * c.y) - a.y * (b.x * c.z - b.z * c.x) + a.z * (b.x * c.y - b.y * c.x);
    const double den = la * lb * lc + dot3(a, b) * lc + dot3(a, c) * lb + dot3(b, c) * la;
    return 2.0 * atan2(det, den);
}
__device__ inline double mesh_exact_w(const double *__restrict__ tri, int nF, d3 p) {
    double sum = 0.0;
    for (int g = 0; g < nF; g++) {
        const double *u = tri + (size_t)9 * g;
        sum += tri_solid_angle_f64(mk3(u[0], u[1], u[2]) - p, mk3(u[3], u[4], u[5]) - p, mk3(u[6], u[7], u[8]) - p);
    }
    return sum / (4.0 * 3.14159265358979323846);
}
__global__ __launch_bounds__(256) void mesh_validity_kernel(const double *__restrict__ tri, int nF, int samples, int face_stride, double eps, double tau_limit, unsigned *out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int n_tested = (nF + face_stride - 1) / face_stride;
    if (i >= (long long)n_tested * samples) return;
    const int f = (int)(i / samples) * face_stride, k = (int)(i % samples);
    const double *t = tri + (size_t)9 * f;
    const d3 a = mk3(t[0], t[1], t[2]), b = mk3(t[3], t[4], t[5]), c = mk3(t[6], t[7], t[8]);
    // barycentric sample: the centroid, then points two thirds of the way to each corner
    const double wa = k == 0 ? 1.0 / 3 : (k == 1 ? 7.0 / 9 : 1.0 / 9), wb = k == 0 ? 1.0 / 3 : (k == 2 ? 7.0 / 9 : 1.0 / 9), wc = 1.0 - wa - wb;
    const d3 q = wa * a + wb * b + wc * c;
    const d3 nr = cross3(b - a, c - a);
    const double nl = sqrt(dot3(nr, nr));
    if (!(nl > 0.0)) return;                                       // a degenerate face bounds nothing
    const d3 n = (1.0 / nl) * nr;
    // (the offset also stays well inside the face's own extent: a sliver is sampled closer to its plane)
    const double off = fmin(eps, 1.0e-3 * sqrt(nl));
    auto is01 = [](double v) { return fabs(v) <= 0.01 || fabs(v - 1.0) <= 0.01; };
    for (int side = 0; side < 2; side++) {
        double tt = off;
        for (int step = 0;; step++, tt *= 2.0) {
            const double w = mesh_exact_w(tri, nF, q + (side ? -tt : tt) * n);
            if (is01(w)) {
                if (step > 0) atomicMax(&out[1], __float_as_uint((float)tt * 1.0000001f));      // a pocket: w is 0 / 1 again tt off the face
                break;
            }
            atomicMax(&out[2], __float_as_uint((float)fabs(1.0 - 2.0 * w) * 1.0000001f));
            if (tt > tau_limit) { atomicOr(&out[0], 1u); break; }                               // not a defect: a region
        }
    }
}

// the mesh kind's distance lattice (DevMesh::dl): one thread per node, the per-lane walks.  The same launch MEASURES what the
// lattice's users assume about the reference's APPROXIMATE winding number (the order-2 far-field expansion moves it by a few 1e-3):
// s = 1 - 2 w is sampled at every node, at the centre of the node's cell and at the midpoints of its three edges (the half-spacing
// lattice but for the face centres: 5 samples per node).  flag bit 0 is raised - and the mesh gets NO lattice - when
//   |s| > 1.04 at ANY sample (on the surface itself s = 0: no distance condition needed), or
//   |s| < 0.96 at a sample farther than `clear` from the surface (node: its own distance; the others: the node's distance - clear,
//   they lie within `clear` of it);
// the users then work with [0.952, 1.05].  range[0] / range[1]: the smallest / largest |s| seen among the lower-checked samples.
__global__ __launch_bounds__(256) void mesh_lattice_kernel(const DevMesh *m, float *out, int nx, int ny, int nz, float mx, float my, float mz, float cell, float clear, unsigned *flag,
                                                           unsigned *range) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)nx * ny * nz) return;
    const int iz = (int)(i % nz), iy = (int)((i / nz) % ny), ix = (int)(i / ((long long)nz * ny));
    // (the node's position exactly as the filter forms it: float arithmetic)
    const d3 p = mk3((double)(mx + (float)ix * cell), (double)(my + (float)iy * cell), (double)(mz + (float)iz * cell));
    d3 c;
    double s;
    const double d = sqrt(mesh_query(m, p, c, s)) * (1.0 - 1.0e-6);
    float f = (float)d;
    if ((double)f > d) f = __int_as_float(__float_as_int(f) - 1);      // (f > d >= 0: the next float towards zero)
    out[i] = s > 0.0 ? f : -f;                   // (the sign BIT says inside: a node on the surface is +0 or -0)
    bool bad = fabs(s) > 1.04 || (d > (double)clear && fabs(s) < 0.96);
    float lo = 2.0f, hi = 0.0f;
    if (d > (double)clear) { lo = fminf(lo, (float)fabs(s)); hi = fmaxf(hi, (float)fabs(s)); }
    const bool lower = d > 2.0 * (double)clear;
    const double hc = 0.5 * (double)cell;
    for (int k = 0; k < 4; k++) {                // cell centre, then the midpoints of the edges towards +x, +y, +z
        const d3 q = mk3(p.x + ((k == 0 || k == 1) ? hc : 0.0), p.y + ((k == 0 || k == 2) ? hc : 0.0), p.z + ((k == 0 || k == 3) ? hc : 0.0));
        const double sk = 1.0 - 2.0 * ((double)mesh_solid_angle(m, (float)q.x, (float)q.y, (float)q.z) / (4.0 * 3.14159265358979323846));
        bad = bad || fabs(sk) > 1.04 || (lower && fabs(sk) < 0.96);
        if (lower) { lo = fminf(lo, (float)fabs(sk)); hi = fmaxf(hi, (float)fabs(sk)); }
    }
    if (bad) atomicOr(flag, 1u);
    if (hi > 0.0f) { atomicMin(&range[0], __float_as_uint(lo)); atomicMax(&range[1], __float_as_uint(hi)); }     // (non-negative floats order like their bits)
}

}  // namespace isdf

// Does the surface bound a solid the way the lattice's users need (mesh_validity_kernel)?  valid_out: 1 yes, 0 no; defect_out: the
// thickness of the thickest defect pocket found (0: none) and the largest |1 - 2 w| inside one
int isdf_mesh_surface_valid(isdf_ctx *c, const double *d_tri, int nF, double extent, double tau_limit, int *valid_out, float defect_out[2]) {
    *valid_out = 0; defect_out[0] = defect_out[1] = 0.f;
    unsigned *d_flag = nullptr;
    HIPCHK(c, hipMalloc(&d_flag, 3 * sizeof(unsigned)));
    HIPCHK(c, hipMemsetAsync(d_flag, 0, 3 * sizeof(unsigned), c->stream));
    const int samples = nF <= 4096 ? 4 : 1;            // (the test is quadratic in the faces: the reference's 12 000-face Trefoil.obj takes the centroids)
    // (... and beyond 32 768 faces every k-th face only, so that isdf_set_shape stays within ~0.1 s: 1e9 solid angles)
    const int face_stride = nF <= 32768 ? 1 : (int)(((long long)nF * nF + (1ll << 30) - 1) >> 30);
    const long long total = (long long)((nF + face_stride - 1) / face_stride) * samples;
    hipLaunchKernelGGL(isdf::mesh_validity_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, d_tri, nF, samples, face_stride, 1.0e-5 * extent, tau_limit, d_flag);
    unsigned back[3] = {1u, 0u, 0u};
    const hipError_t e1 = hipGetLastError();
    const hipError_t e2 = hipMemcpyAsync(back, d_flag, sizeof(back), hipMemcpyDeviceToHost, c->stream);
    const hipError_t e3 = hipStreamSynchronize(c->stream);
    (void)hipFree(d_flag);
    HIPCHK(c, e1); HIPCHK(c, e2); HIPCHK(c, e3);
    *valid_out = (back[0] & 1u) ? 0 : 1;
    std::memcpy(&defect_out[0], &back[1], 4); std::memcpy(&defect_out[1], &back[2], 4);
    return ISDF_OK;
}

// builds the lattice over `lo .. hi` (body frame) with `n` cells along the longest axis; fills the DevMesh fields (host copy `hm`);
// s_range_out: the measured range of |1 - 2 w| away from the surface (0, 0: nothing measured)
int isdf_mesh_lattice_build(isdf_ctx *c, isdf::DevMesh *hm, const double lo[3], const double hi[3], int n, float s_range_out[2]) {
    s_range_out[0] = s_range_out[1] = 0.f;
    double ext = 0.0;
    for (int a = 0; a < 3; a++) ext = std::max(ext, hi[a] - lo[a]);
    if (!(ext > 0.0) || n < 2) return ISDF_OK;
    const float cell = (float)(ext / n);
    int dims[3];
    for (int a = 0; a < 3; a++) dims[a] = (int)std::ceil((hi[a] - lo[a]) / cell) + 1;
    const size_t total = (size_t)dims[0] * dims[1] * dims[2];
    HIPCHK(c, hipMalloc(&c->d_mesh_dl, (total + 3) * sizeof(float)));      // (+ the flag word and the two range words)
    unsigned *d_flag = reinterpret_cast<unsigned *>(c->d_mesh_dl + total);
    const unsigned init[3] = {0u, 0x7F7FFFFFu, 0u};
    HIPCHK(c, hipMemcpyAsync(d_flag, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
    const float h = 0.5f * 1.7320508f * cell * 1.001f + 1.0e-5f * (float)ext;      // half a cell diagonal, + the float rounding of positions of this size
    hipLaunchKernelGGL(isdf::mesh_lattice_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, c->d_mesh, c->d_mesh_dl, dims[0], dims[1], dims[2],
                       (float)lo[0], (float)lo[1], (float)lo[2], cell, h, d_flag, d_flag + 1);
    HIPCHK(c, hipGetLastError());
    unsigned back[3] = {1u, 0u, 0u};
    HIPCHK(c, hipMemcpyAsync(back, d_flag, sizeof(back), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (back[2] != 0u) { std::memcpy(&s_range_out[0], &back[1], 4); std::memcpy(&s_range_out[1], &back[2], 4); }
    if (back[0]) { (void)hipFree(c->d_mesh_dl); c->d_mesh_dl = nullptr; return ISDF_OK; }      // not a mesh the lattice's users can reason about
    hm->dl = c->d_mesh_dl;
    for (int a = 0; a < 3; a++) { hm->dln[a] = dims[a]; hm->dl_min[a] = (float)lo[a]; }
    hm->dl_inv = 1.0f / cell;
    hm->dl_h = h;
    return ISDF_OK;
}

extern "C" int isdf_shape_eval(isdf_ctx *c, const double *p_rel, int n, double *sdf_out, double *grad_out) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    if (n < 0 || (n > 0 && !p_rel)) return isdf_fail(c, ISDF_ERR_INVALID_ARG, "bad points");
    if (!c->have_shape) return isdf_fail(c, ISDF_ERR_STATE, "isdf_set_shape has not been called");
    if (n == 0 || (!sdf_out && !grad_out)) return ISDF_OK;
    HIPCHK(c, hipSetDevice(c->device));
    double *d = nullptr;
    HIPCHK(c, hipMalloc(&d, (size_t)n * 7 * sizeof(double)));
    double *d_p = d, *d_s = d + (size_t)3 * n, *d_g = d + (size_t)4 * n;
    int rc = ISDF_OK;
    auto chk = [&](hipError_t e, const char *what) { if (e != hipSuccess && rc == ISDF_OK) { c->err = std::string(what) + ": " + hipGetErrorString(e); rc = ISDF_ERR_HIP; } };
    chk(hipMemcpyAsync(d_p, p_rel, (size_t)3 * n * sizeof(double), hipMemcpyHostToDevice, c->stream), "upload points");
    if (rc == ISDF_OK) {
        const dim3 grid((n + 255) / 256), block(256);
        if (c->shape.kind == ISDF_SHAPE_MESH) hipLaunchKernelGGL(isdf::shape_eval_kernel<ISDF_SHAPE_MESH>, grid, block, 0, c->stream, c->shape, d_p, n, sdf_out ? d_s : nullptr, grad_out ? d_g : nullptr);
        else hipLaunchKernelGGL(isdf::shape_eval_kernel<-1>, grid, block, 0, c->stream, c->shape, d_p, n, sdf_out ? d_s : nullptr, grad_out ? d_g : nullptr);
        chk(hipGetLastError(), "shape_eval_kernel");
    }
    if (rc == ISDF_OK && sdf_out) chk(hipMemcpyAsync(sdf_out, d_s, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream), "download sdf");
    if (rc == ISDF_OK && grad_out) chk(hipMemcpyAsync(grad_out, d_g, (size_t)3 * n * sizeof(double), hipMemcpyDeviceToHost, c->stream), "download grad");
    chk(hipStreamSynchronize(c->stream), "shape_eval sync");
    (void)hipFree(d);
    return rc;
}

// Host evaluation of the float atan2 the device's winding number uses (dev_mesh.hpp: the C library's algorithm restated) - so the
// CPU tests can hold it against libm bit for bit; n pairs, out[i] = atan2f(y[i], x[i]).
extern "C" int isdf_mesh_atan2f(const float *y, const float *x, long long n, float *out) {
    if (n < 0 || (n > 0 && (!y || !x || !out))) return ISDF_ERR_INVALID_ARG;
    for (long long i = 0; i < n; i++) out[i] = isdf::glibc_atan2f(y[i], x[i]);
    return ISDF_OK;
}
