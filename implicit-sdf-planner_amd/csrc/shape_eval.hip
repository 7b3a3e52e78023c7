// The robot-shape plugin by itself: BasicShape::getonlySDF / getonlyGrad1 / getSDFwithGrad1 (Shape.hpp:32-88 for the
// analytic classes, Shape.cpp:105-151 for the mesh Generalshape) on a batch of BODY-FRAME points, one lane per point.
// The sweeps inline the same device functions (dev_shapes.hpp / dev_mesh.hpp); this entry point exists so a caller (and
// the parity tests) can query the installed shape directly, like the reference's planners and visualisers do.
#include "isdf_ctx.hpp"
#include "dev_mesh.hpp"

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

// the mesh kind's distance lattice (DevMesh::dl): one thread per node, the per-lane walks
// (flag: raised when a node farther than `clear` from the surface has |1 - 2 w| outside [0.96, 1.04] - a surface covered twice, an
// inverted one: the assumption the lattice's users make about the winding number does not hold for this mesh, and it gets no lattice)
__global__ __launch_bounds__(256) void mesh_lattice_kernel(const DevMesh *m, float *out, int nx, int ny, int nz, float mx, float my, float mz, float cell, float clear, unsigned *flag) {
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
    if (d > (double)clear && !(fabs(s) >= 0.96 && fabs(s) <= 1.04)) atomicOr(flag, 1u);
}

}  // namespace isdf

// builds the lattice over `lo .. hi` (body frame) with `n` cells along the longest axis; fills the DevMesh fields (host copy `hm`)
int isdf_mesh_lattice_build(isdf_ctx *c, isdf::DevMesh *hm, const double lo[3], const double hi[3], int n) {
    double ext = 0.0;
    for (int a = 0; a < 3; a++) ext = std::max(ext, hi[a] - lo[a]);
    if (!(ext > 0.0) || n < 2) return ISDF_OK;
    const float cell = (float)(ext / n);
    int dims[3];
    for (int a = 0; a < 3; a++) dims[a] = (int)std::ceil((hi[a] - lo[a]) / cell) + 1;
    const size_t total = (size_t)dims[0] * dims[1] * dims[2];
    HIPCHK(c, hipMalloc(&c->d_mesh_dl, (total + 1) * sizeof(float)));      // (+ the flag word)
    unsigned *d_flag = reinterpret_cast<unsigned *>(c->d_mesh_dl + total);
    HIPCHK(c, hipMemsetAsync(d_flag, 0, sizeof(unsigned), c->stream));
    const float h = 0.5f * 1.7320508f * cell * 1.001f + 1.0e-5f * (float)ext;      // half a cell diagonal, + the float rounding of positions of this size
    hipLaunchKernelGGL(isdf::mesh_lattice_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, c->d_mesh, c->d_mesh_dl, dims[0], dims[1], dims[2],
                       (float)lo[0], (float)lo[1], (float)lo[2], cell, h, d_flag);
    HIPCHK(c, hipGetLastError());
    unsigned flag = 0;
    HIPCHK(c, hipMemcpyAsync(&flag, d_flag, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (flag) { (void)hipFree(c->d_mesh_dl); c->d_mesh_dl = nullptr; return ISDF_OK; }      // not a mesh the lattice's users can reason about
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
