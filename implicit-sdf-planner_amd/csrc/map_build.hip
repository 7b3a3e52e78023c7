// Once-per-plan map products on the device:
//   isdf_set_pointcloud   point cloud -> occupancy grid          PCSmapManager::rcvGlobalMapHandler, PCSmap_manager.cpp:87-200
//                                                                 (createGridMap Gridmap3D.cpp:25-39, getGridIndex :135-175)
//   isdf_generate_esdf    occupancy -> unsigned ESDF (metres)     GridMap3D::generateESDF3d, Gridmap3D.cpp:361-414 (+ fillESDF :497-533)
//   isdf_gather_points    occupied voxel centres around waypoints plan_manager.cpp:232-254 with getPointsInAABBOutOfLastOne,
//                                                                 PCSmap_manager.h:182-216
// All integer / byte work, HBM-bound.  The ESDF is the exact Euclidean distance transform: the reference's three
// passes of Felzenszwalb's 1-D lower envelope produce, for every voxel, the exact squared voxel distance to the nearest
// occupied voxel, and that integer is unique - so the passes here compute the same integer a different way
// (z: nearest set bit of the occupancy bit-row; y, x: bounded outward scan, exact because a candidate at offset r
// cannot beat the current best once r^2 >= best) and res * sqrt(d2) is bit-identical to the reference's double.
#include "isdf_ctx.hpp"
#include <cmath>
#include <cstring>
#include <limits>

namespace isdf {

constexpr int EDT_INF = 0x3fffffff;        // "no occupied voxel on this line / plane yet"

// ---------------------------------------------------------------------------------------------------------------
// point cloud -> counts -> occupancy
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pc_count_kernel(const float *__restrict__ xyz, long long n, DevGrid G, unsigned *__restrict__ counts) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const double x = (double)xyz[3 * i], y = (double)xyz[3 * i + 1], z = (double)xyz[3 * i + 2];
        int ix = 0, iy = 0, iz = 0;      // getGridIndex: a point outside the map lands in voxel (0,0,0) (:137-140)
        const bool in = !(x < G.bmin[0] || y < G.bmin[1] || z < G.bmin[2] || x > G.bmax[0] || y > G.bmax[1] || z > G.bmax[2]);
        if (in) {
            ix = (int)floor((x - G.bmin[0]) / G.res); iy = (int)floor((y - G.bmin[1]) / G.res); iz = (int)floor((z - G.bmin[2]) / G.res);
            if (ix >= G.X) ix = G.X - 1;       // the lower clamps of :149-168 cannot trigger inside the map
            if (iy >= G.Y) iy = G.Y - 1;
            if (iz >= G.Z) iz = G.Z - 1;
        }
        atomicAdd(&counts[((size_t)ix * G.Y + iy) * G.Z + iz], 1u);
    }
}
__global__ __launch_bounds__(256) void pc_threshold_kernel(const unsigned *__restrict__ counts, size_t n, unsigned thr, uint8_t *__restrict__ occ) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        occ[i] = counts[i] >= thr ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------------------
// exact Euclidean distance transform
// ---------------------------------------------------------------------------------------------------------------
// z pass: one lane per voxel of a z-row chunk; the row's occupancy bits (ballot) give the nearest set bit on both sides.
__global__ __launch_bounds__(256) void edt_z_kernel(DevGrid G, const unsigned *__restrict__ bits, int *__restrict__ d2) {
    const size_t n_rows = (size_t)G.X * G.Y;
    const size_t row0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    for (size_t row = row0; row < n_rows; row += n_waves) {
        const unsigned *rb = bits + row * G.ZW;
        for (int z = lane; z < G.Z; z += 64) {
            // nearest set bit at or below z, and above z
            int best = EDT_INF;
            int w = z >> 5;
            unsigned m = rb[w] & (0xFFFFFFFFu >> (31 - (z & 31)));
            int ww = w;
            while (!m && ww > 0) m = rb[--ww];
            if (m) { const int p = (ww << 5) + 31 - __clz(m); best = (z - p) * (z - p); }
            m = (z & 31) == 31 ? 0u : (rb[w] & (0xFFFFFFFFu << ((z & 31) + 1)));
            ww = w;
            while (!m && ww + 1 < G.ZW) m = rb[++ww];
            if (m) { const int p = (ww << 5) + __ffs(m) - 1; if (p < G.Z) best = min(best, (p - z) * (p - z)); }
            d2[row * G.Z + z] = best;
        }
    }
}
// y / x pass: thread per voxel, lanes along z (coalesced); out(q) = min_p (q - p)^2 + in(p) over the line through q.
// AXIS 1: line along y (stride Z); AXIS 0: line along x (stride Y*Z) and the final res * sqrt conversion.
template <int AXIS>
__global__ __launch_bounds__(256) void edt_line_kernel(DevGrid G, const int *__restrict__ in, int *__restrict__ out_i, float *__restrict__ out_f) {
    const size_t total = (size_t)G.X * G.Y * G.Z;
    const int n = AXIS == 1 ? G.Y : G.X;
    const size_t stride = AXIS == 1 ? (size_t)G.Z : (size_t)G.Y * G.Z;
    for (size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x; a < total; a += (size_t)gridDim.x * blockDim.x) {
        const int q = AXIS == 1 ? (int)((a / G.Z) % G.Y) : (int)(a / ((size_t)G.Y * G.Z));
        int best = in[a];
        for (int r = 1; r < n; r++) {
            const int rr = r * r;
            if (rr >= best) break;                       // farther samples cost at least r^2
            if (q - r >= 0) best = min(best, rr + in[a - (size_t)r * stride]);      // INF + rr stays < 2^31
            if (q + r < n) best = min(best, rr + in[a + (size_t)r * stride]);
        }
        if (AXIS == 1) out_i[a] = best;
        else {
            // grid_esdf = grid_resolution * std::sqrt(val) (Gridmap3D.cpp:409); no occupied voxel -> sqrt(DBL_MAX)
            const double v = best >= EDT_INF ? 1.7976931348623157e308 : (double)best;
            out_f[a] = (float)(G.res * sqrt(v));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// constraint-point gather
// ---------------------------------------------------------------------------------------------------------------
struct GatherBox { int lo[3], hi[3], llo[3], lhi[3]; };    // this waypoint's index box and the previous waypoint's
__global__ __launch_bounds__(256) void gather_mark_kernel(DevGrid G, const GatherBox *__restrict__ boxes, const unsigned *__restrict__ occ_bits,
                                                          unsigned *__restrict__ marks) {
    const GatherBox b = boxes[blockIdx.x];
    const int nx = b.hi[0] - b.lo[0] + 1, ny = b.hi[1] - b.lo[1] + 1, nz = b.hi[2] - b.lo[2] + 1;
    const long long nvox = (long long)nx * ny * nz;
    for (long long t = threadIdx.x; t < nvox; t += blockDim.x) {
        const int k = b.lo[2] + (int)(t % nz), j = b.lo[1] + (int)((t / nz) % ny), i = b.lo[0] + (int)(t / ((long long)nz * ny));
        const bool outside_last = i > b.lhi[0] || i < b.llo[0] || j > b.lhi[1] || j < b.llo[1] || k > b.lhi[2] || k < b.llo[2];
        if (!outside_last) continue;
        const size_t w = ((size_t)i * G.Y + j) * G.ZW + (k >> 5);
        const unsigned bit = 1u << (k & 31);
        if (occ_bits[w] & bit) atomicOr(&marks[w], bit);
    }
}
constexpr int GATHER_CHUNK = 1024;     // words per block in the count / emit kernels
__global__ __launch_bounds__(256) void gather_count_kernel(const unsigned *__restrict__ marks, size_t n_words, unsigned *__restrict__ chunk_cnt) {
    __shared__ unsigned s[256];
    const size_t base = (size_t)blockIdx.x * GATHER_CHUNK;
    unsigned c = 0;
    for (int t = threadIdx.x; t < GATHER_CHUNK; t += 256) if (base + t < n_words) c += __popc(marks[base + t]);
    s[threadIdx.x] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) chunk_cnt[blockIdx.x] = s[0];
}
__global__ __launch_bounds__(256) void gather_scan_kernel(unsigned *chunk_cnt, int n_chunks, unsigned *total) {   // one block
    __shared__ unsigned s_carry;
    __shared__ unsigned s[256];
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < n_chunks; b0 += 256) {
        const int i = b0 + threadIdx.x;
        const unsigned v = i < n_chunks ? chunk_cnt[i] : 0u;
        s[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            const unsigned add = (int)threadIdx.x >= o ? s[threadIdx.x - o] : 0u;
            __syncthreads();
            s[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < n_chunks) chunk_cnt[i] = s_carry + s[threadIdx.x] - v;      // exclusive offset
        __syncthreads();
        if (threadIdx.x == 255) s_carry += s[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = s_carry;
}
__global__ __launch_bounds__(64) void gather_emit_kernel(DevGrid G, const unsigned *__restrict__ marks, size_t n_words,
                                                         const unsigned *__restrict__ chunk_off, double *__restrict__ pts, unsigned cap) {
    // one wave per chunk: words in ascending order -> points sorted by (ix, iy, iz)
    const size_t base = (size_t)blockIdx.x * GATHER_CHUNK;
    unsigned off = chunk_off[blockIdx.x];
    const int lane = threadIdx.x;
    for (int t0 = 0; t0 < GATHER_CHUNK; t0 += 64) {
        const size_t w = base + t0 + lane;
        unsigned m = w < n_words ? marks[w] : 0u;
        const int c = __popc(m);
        int incl = c;                                        // inclusive scan over the wave
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
        unsigned pos = off + (unsigned)(incl - c);
        const size_t row = w / G.ZW;
        const int zw = (int)(w - row * G.ZW);
        const int ix = (int)(row / G.Y), iy = (int)(row - (size_t)ix * G.Y);
        while (m) {
            const int z = (zw << 5) + __ffs(m) - 1;
            m &= m - 1u;
            if (pos < cap) {     // getGridCubeCenter (Gridmap3D.cpp:182-194)
                pts[3 * (size_t)pos] = (ix + 0.5) * G.res + G.bmin[0];
                pts[3 * (size_t)pos + 1] = (iy + 0.5) * G.res + G.bmin[1];
                pts[3 * (size_t)pos + 2] = (z + 0.5) * G.res + G.bmin[2];
            }
            pos++;
        }
        off += (unsigned)__shfl(incl, 63, 64);
    }
}

void launch_build_bits(const DevGrid &G, int use_esdf, double thresh, unsigned *out, hipStream_t stream);   // tile_sweep.hip

} // namespace isdf

using namespace isdf;

static size_t n_bit_words(const DevGrid &G) { return (size_t)G.X * G.Y * ((G.Z + 31) / 32); }

// occupancy bit-grid (padding bits of the last word of a row are zero)
static int occupancy_bits(isdf_ctx *c, unsigned **out_bits) {
    DevGrid G = c->grid;
    G.ZW = (G.Z + 31) / 32;
    unsigned *b = nullptr;
    HIPCHK(c, hipMalloc(&b, n_bit_words(G) * sizeof(unsigned)));
    G.occ = c->d_occ;
    launch_build_bits(G, 0, 0.0, b, c->stream);
    *out_bits = b;
    return ISDF_OK;
}

extern "C" int isdf_set_pointcloud(isdf_ctx *c, const float *xyz, long long n_points, const double *bmin_opt, const double *bmax_opt,
                                   double resolution, int sta_threshold, int dims_out[3]) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    if (!xyz || n_points < 1 || !(resolution > 0) || sta_threshold < 0) return isdf_fail(c, ISDF_ERR_INVALID_ARG, "bad point cloud arguments");
    if ((bmin_opt == nullptr) != (bmax_opt == nullptr)) return isdf_fail(c, ISDF_ERR_INVALID_ARG, "give both boundaries or neither");
    HIPCHK(c, hipSetDevice(c->device));
    double bmin[3], bmax[3];
    if (bmin_opt) { for (int a = 0; a < 3; a++) { bmin[a] = bmin_opt[a]; bmax[a] = bmax_opt[a]; } }
    else {   // "measure boundary": the tight box of the cloud (PCSmap_manager.cpp:110-141)
        for (int a = 0; a < 3; a++) { bmin[a] = 1.0e300; bmax[a] = -1.0e300; }
        for (long long i = 0; i < n_points; i++)
            for (int a = 0; a < 3; a++) {
                const double v = (double)xyz[3 * i + a];
                if (v > bmax[a]) bmax[a] = v;
                if (v < bmin[a]) bmin[a] = v;
            }
    }
    long long dim[3];
    for (int a = 0; a < 3; a++) {
        dim[a] = (long long)std::ceil((bmax[a] - bmin[a]) / resolution);      // createGridMap (Gridmap3D.cpp:29-31)
        if (dim[a] < 1 || dim[a] > 4096) return isdf_fail(c, ISDF_ERR_INVALID_ARG, "grid dimension out of range [1, 4096]");
    }
    const size_t n = (size_t)dim[0] * dim[1] * dim[2];
    if (c->d_esdf) { (void)hipFree(c->d_esdf); c->d_esdf = nullptr; }
    isdf_frontend_release(c);
    if (c->d_occ) { (void)hipFree(c->d_occ); c->d_occ = nullptr; }
    c->grid.X = (int)dim[0]; c->grid.Y = (int)dim[1]; c->grid.Z = (int)dim[2]; c->grid.res = resolution;
    for (int a = 0; a < 3; a++) { c->grid.bmin[a] = bmin[a]; c->grid.bmax[a] = bmax[a]; }
    c->have_geom = true;
    float *d_xyz = nullptr; unsigned *d_cnt = nullptr;
    HIPCHK(c, hipMalloc(&d_xyz, (size_t)n_points * 3 * sizeof(float)));
    HIPCHK(c, hipMalloc(&d_cnt, n * sizeof(unsigned)));
    HIPCHK(c, hipMalloc(&c->d_occ, n));
    HIPCHK(c, hipMemcpyAsync(d_xyz, xyz, (size_t)n_points * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(d_cnt, 0, n * sizeof(unsigned), c->stream));
    hipLaunchKernelGGL(pc_count_kernel, dim3(2048), dim3(256), 0, c->stream, d_xyz, n_points, c->grid, d_cnt);
    hipLaunchKernelGGL(pc_threshold_kernel, dim3(4096), dim3(256), 0, c->stream, d_cnt, n, (unsigned)sta_threshold, c->d_occ);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    (void)hipFree(d_xyz); (void)hipFree(d_cnt);
    c->grid.esdf = nullptr; c->grid.occ = c->d_occ;
    c->bits_dirty = true;
    if (dims_out) { dims_out[0] = (int)dim[0]; dims_out[1] = (int)dim[1]; dims_out[2] = (int)dim[2]; }
    ISDF_REPLICATE(c, isdf_set_pointcloud(p_, xyz, n_points, bmin_opt, bmax_opt, resolution, sta_threshold, nullptr));
    return ISDF_OK;
}

extern "C" int isdf_generate_esdf(isdf_ctx *c) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    if (!c->have_geom || !c->d_occ) return isdf_fail(c, ISDF_ERR_STATE, "generate_esdf needs an occupancy grid");
    HIPCHK(c, hipSetDevice(c->device));
    DevGrid G = c->grid;
    G.ZW = (G.Z + 31) / 32;
    const size_t n = (size_t)G.X * G.Y * G.Z;
    unsigned *bits = nullptr;
    int rc = occupancy_bits(c, &bits);
    if (rc) return rc;
    int *a = nullptr, *b = nullptr;
    HIPCHK(c, hipMalloc(&a, n * sizeof(int)));
    HIPCHK(c, hipMalloc(&b, n * sizeof(int)));
    if (c->d_esdf) { (void)hipFree(c->d_esdf); c->d_esdf = nullptr; }
    HIPCHK(c, hipMalloc(&c->d_esdf, n * sizeof(float)));
    hipLaunchKernelGGL(edt_z_kernel, dim3(4096), dim3(256), 0, c->stream, G, bits, a);
    hipLaunchKernelGGL(edt_line_kernel<1>, dim3(8192), dim3(256), 0, c->stream, G, a, b, (float *)nullptr);
    hipLaunchKernelGGL(edt_line_kernel<0>, dim3(8192), dim3(256), 0, c->stream, G, b, (int *)nullptr, c->d_esdf);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    (void)hipFree(a); (void)hipFree(b); (void)hipFree(bits);
    c->grid.esdf = c->d_esdf;
    c->bits_dirty = true;
    c->bricks_stale = true;
    ISDF_REPLICATE(c, isdf_generate_esdf(p_));
    return ISDF_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// ESDF point sampler: GridMap3D::getSDFValue / getSDFValueWithGrad (GridMap3D.h:114-146, :155-193) at n world points
// ---------------------------------------------------------------------------------------------------------------
// One thread per point: base cell of pos - res/2 (getGridIndex: a position outside the map gives cell (0,0,0), Gridmap3D.cpp:137-140),
// the eight corner values (an invalid corner reads 0, :535-542; the isInMap test of the loop is overwritten, GridMap3D.h:128-131),
// the reference's interpolation order and its ANALYTIC gradient of the interpolant (:185-191) - same operations, no contraction:
// bit for bit the reference's doubles on a float32 ESDF.  The one operation of this path that is a pure gather: 24 B in, 8 random
// 4-byte reads (four 32-byte sectors: the two z corners of a column are neighbours), 8-32 B out per point - HBM / L2 bound.
namespace isdf {
// base cell of a world point (getGridIndex of pos - res/2, Gridmap3D.cpp:135-168) and its fractional offsets
__device__ __forceinline__ void esdf_base_cell(const DevGrid &G, double px, double py, double pz, int &ix, int &iy, int &iz, double &dx, double &dy, double &dz, double &inv) {
#pragma clang fp contract(off)
    const double hx = px - 0.5 * G.res * 1.0, hy = py - 0.5 * G.res * 1.0, hz = pz - 0.5 * G.res * 1.0;   // pos_w - 0.5 * res * Ones()
    ix = 0; iy = 0; iz = 0;
    const bool in = !(hx < G.bmin[0] || hy < G.bmin[1] || hz < G.bmin[2] || hx > G.bmax[0] || hy > G.bmax[1] || hz > G.bmax[2]);
    if (in) {
        ix = (int)floor((hx - G.bmin[0]) / G.res); iy = (int)floor((hy - G.bmin[1]) / G.res); iz = (int)floor((hz - G.bmin[2]) / G.res);
        // (the reference's "iy < 0 -> ix = 0" typos cannot fire: the differences are >= 0 inside the map)
        if (ix < 0) ix = 0;
        if (ix >= G.X) ix = G.X - 1;
        if (iy >= G.Y) iy = G.Y - 1;
        if (iz >= G.Z) iz = G.Z - 1;
    }
    const double cx = (ix + 0.5) * G.res + G.bmin[0], cy = (iy + 0.5) * G.res + G.bmin[1], cz = (iz + 0.5) * G.res + G.bmin[2];
    inv = 1.0 / G.res;
    dx = (px - cx) * inv; dy = (py - cy) * inv; dz = (pz - cz) * inv;
}
// the reference's interpolation order and its analytic gradient of the interpolant (GridMap3D.h:133-145, :185-191)
__device__ __forceinline__ void esdf_interpolate(const double (&v)[2][2][2], double dx, double dy, double dz, double inv, double *val, double *grad, long long i) {
#pragma clang fp contract(off)
    const double v00 = (1 - dx) * v[0][0][0] + dx * v[1][0][0];
    const double v01 = (1 - dx) * v[0][0][1] + dx * v[1][0][1];
    const double v10 = (1 - dx) * v[0][1][0] + dx * v[1][1][0];
    const double v11 = (1 - dx) * v[0][1][1] + dx * v[1][1][1];
    const double v0 = (1 - dy) * v00 + dy * v10;
    const double v1 = (1 - dy) * v01 + dy * v11;
    if (val) val[i] = (1 - dz) * v0 + dz * v1;
    if (grad) {
        const double gz = (v1 - v0) * inv;
        const double gy = ((1 - dz) * (v10 - v00) + dz * (v11 - v01)) * inv;
        double gx = (1 - dz) * (1 - dy) * (v[1][0][0] - v[0][0][0]);
        gx += (1 - dz) * dy * (v[1][1][0] - v[0][1][0]);
        gx += dz * (1 - dy) * (v[1][0][1] - v[0][0][1]);
        gx += dz * dy * (v[1][1][1] - v[0][1][1]);
        gx *= inv;
        grad[3 * i] = gx; grad[3 * i + 1] = gy; grad[3 * i + 2] = gz;
    }
}
__global__ __launch_bounds__(256) void esdf_sample_kernel(DevGrid G, const double *__restrict__ xyz, long long n, double *__restrict__ val, double *__restrict__ grad) {
#pragma clang fp contract(off)
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        int ix, iy, iz;
        double dx, dy, dz, inv;
        esdf_base_cell(G, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], ix, iy, iz, dx, dy, dz, inv);
        double v[2][2][2];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    const int x = ix + a, y = iy + b, z = iz + c;
                    const bool ok = x < G.X && y < G.Y && z < G.Z;          // (lower bounds hold by construction)
                    v[a][b][c] = ok ? (double)__builtin_nontemporal_load(G.esdf + (((size_t)x * G.Y + y) * G.Z + z)) : 0.0;
                }
        esdf_interpolate(v, dx, dy, dz, inv, val, grad, i);
    }
}

// ---- SCATTERED points.  In the z-fastest grid the eight corners of a cell lie in four different z-rows: a point that shares no
// cache line with its neighbours costs four 128-byte lines (measured: ~1 KB of HBM traffic per point, 5.7e9 points/s, the
// memory system already moving 5.8 TB/s - profiles/r4_esdf_sample_pmc.txt).  The BRICKED copy stores the grid as 2 x 2 x 2-cell
// bricks WITH their one-voxel apron - 3 x 3 x 3 = 27 floats, padded to 32 = exactly one 128-byte line - so that the eight
// corners of ANY base cell sit in ONE line: brick (ix >> 1, iy >> 1, iz >> 1), entry ((ix & 1) + a) * 9 + ((iy & 1) + b) * 3 +
// (iz & 1) + c.  Entries beyond the map hold the 0 the reference reads there.  Four times the grid's memory (256 MiB at 256^3),
// built once per map on first use; the same arithmetic on the same values: bitwise the z-fastest path's results.
constexpr int BRICK_FLOATS = 32;
__global__ __launch_bounds__(256) void esdf_brick_build_kernel(DevGrid G, int BX, int BY, int BZ, float *__restrict__ bricks) {
    const size_t n = (size_t)BX * BY * BZ * BRICK_FLOATS;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(t & (BRICK_FLOATS - 1));
        size_t b = t / BRICK_FLOATS;
        const int bz = (int)(b % BZ); b /= BZ;
        const int by = (int)(b % BY); const int bx = (int)(b / BY);
        float v = 0.f;
        if (e < 27) {
            const int x = 2 * bx + e / 9, y = 2 * by + (e / 3) % 3, z = 2 * bz + e % 3;
            if (x < G.X && y < G.Y && z < G.Z) v = G.esdf[((size_t)x * G.Y + y) * G.Z + z];
        }
        bricks[t] = v;
    }
}
__global__ __launch_bounds__(256) void esdf_sample_bricked_kernel(DevGrid G, int BY, int BZ, const float *__restrict__ bricks, const double *__restrict__ xyz, long long n,
                                                                  double *__restrict__ val, double *__restrict__ grad) {
#pragma clang fp contract(off)
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        int ix, iy, iz;
        double dx, dy, dz, inv;
        esdf_base_cell(G, __builtin_nontemporal_load(xyz + 3 * i), __builtin_nontemporal_load(xyz + 3 * i + 1), __builtin_nontemporal_load(xyz + 3 * i + 2), ix, iy, iz, dx, dy, dz, inv);
        const float *br = bricks + (((size_t)(ix >> 1) * BY + (iy >> 1)) * BZ + (iz >> 1)) * BRICK_FLOATS + (ix & 1) * 9 + (iy & 1) * 3 + (iz & 1);
        double v[2][2][2];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++)
#pragma unroll
                for (int c = 0; c < 2; c++) v[a][b][c] = (double)__builtin_nontemporal_load(br + a * 9 + b * 3 + c);
        esdf_interpolate(v, dx, dy, dz, inv, val, grad, i);
    }
}

}  // namespace isdf

extern "C" int isdf_esdf_sample_device(isdf_ctx *c, const double *d_xyz, long long n, double *d_value, double *d_grad, void *stream) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    if (n < 0 || (n > 0 && !d_xyz)) return isdf_fail(c, ISDF_ERR_INVALID_ARG, "bad points");
    if (!c->have_geom || !c->d_esdf) return isdf_fail(c, ISDF_ERR_STATE, "no ESDF grid (isdf_set_grid with ISDF_GRID_ESDF, or isdf_generate_esdf)");
    if (n == 0 || (!d_value && !d_grad)) return ISDF_OK;
    HIPCHK(c, hipSetDevice(c->device));
    DevGrid G = c->grid;
    G.esdf = c->d_esdf;
    const long long blocks = std::min<long long>((n + 255) / 256, 256ll * 64);
    hipLaunchKernelGGL(esdf_sample_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, G, d_xyz, n, d_value, d_grad);
    HIPCHK(c, hipGetLastError());
    return ISDF_OK;
}

// the bricked copy, built (and rebuilt after the ESDF changed) on first use, on the caller's stream
static int esdf_bricks_ready(isdf_ctx *c, hipStream_t st) {
    const int BX = (c->grid.X + 1) / 2, BY = (c->grid.Y + 1) / 2, BZ = (c->grid.Z + 1) / 2;
    const size_t need = (size_t)BX * BY * BZ * isdf::BRICK_FLOATS;
    if (c->d_esdf_bricks && !c->bricks_stale && c->bricks_cap == need) return ISDF_OK;
    if (c->bricks_cap != need) {
        if (c->d_esdf_bricks) { HIPCHK(c, hipDeviceSynchronize()); (void)hipFree(c->d_esdf_bricks); }
        c->d_esdf_bricks = nullptr; c->bricks_cap = 0;
        HIPCHK(c, hipMalloc(&c->d_esdf_bricks, need * sizeof(float)));
        c->bricks_cap = need;
    }
    DevGrid G = c->grid;
    G.esdf = c->d_esdf;
    hipLaunchKernelGGL(isdf::esdf_brick_build_kernel, dim3(256 * 32), dim3(256), 0, st, G, BX, BY, BZ, c->d_esdf_bricks);
    HIPCHK(c, hipGetLastError());
    c->bricks_stale = false;
    return ISDF_OK;
}
extern "C" int isdf_esdf_sample_scattered_device(isdf_ctx *c, const double *d_xyz, long long n, double *d_value, double *d_grad, void *stream) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    if (n < 0 || (n > 0 && !d_xyz)) return isdf_fail(c, ISDF_ERR_INVALID_ARG, "bad points");
    if (!c->have_geom || !c->d_esdf) return isdf_fail(c, ISDF_ERR_STATE, "no ESDF grid (isdf_set_grid with ISDF_GRID_ESDF, or isdf_generate_esdf)");
    if (n == 0 || (!d_value && !d_grad)) return ISDF_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const int rc = esdf_bricks_ready(c, (hipStream_t)stream);
    if (rc) return rc;
    DevGrid G = c->grid;
    const long long blocks = std::min<long long>((n + 255) / 256, 256ll * 64);
    hipLaunchKernelGGL(isdf::esdf_sample_bricked_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, G, (c->grid.Y + 1) / 2, (c->grid.Z + 1) / 2,
                       c->d_esdf_bricks, d_xyz, n, d_value, d_grad);
    HIPCHK(c, hipGetLastError());
    return ISDF_OK;
}

static int esdf_sample_host(isdf_ctx *c, const double *xyz, long long n, double *value_out, double *grad_out, bool scattered);
extern "C" int isdf_esdf_sample(isdf_ctx *c, const double *xyz, long long n, double *value_out, double *grad_out) { return esdf_sample_host(c, xyz, n, value_out, grad_out, false); }
extern "C" int isdf_esdf_sample_scattered(isdf_ctx *c, const double *xyz, long long n, double *value_out, double *grad_out) { return esdf_sample_host(c, xyz, n, value_out, grad_out, true); }
static int esdf_sample_host(isdf_ctx *c, const double *xyz, long long n, double *value_out, double *grad_out, bool scattered) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    if (n < 0 || (n > 0 && !xyz)) return isdf_fail(c, ISDF_ERR_INVALID_ARG, "bad points");
    if (!c->have_geom || !c->d_esdf) return isdf_fail(c, ISDF_ERR_STATE, "no ESDF grid (isdf_set_grid with ISDF_GRID_ESDF, or isdf_generate_esdf)");
    if (n == 0 || (!value_out && !grad_out)) return ISDF_OK;
    HIPCHK(c, hipSetDevice(c->device));
    // staging in the ctx (grows only): a caller that samples every optimizer step pays no allocation and no implicit device
    // synchronisation (hipFree) per call
    { const int rc0 = isdf_ensure_doubles(c, &c->d_esdf_stage, &c->esdf_stage_cap, (size_t)n * 7); if (rc0) return rc0; }
    double *const d = c->d_esdf_stage;
    double *d_p = d, *d_v = d + (size_t)3 * n, *d_g = d + (size_t)4 * n;
    int rc = ISDF_OK;
    auto chk = [&](hipError_t e, const char *what) { if (e != hipSuccess && rc == ISDF_OK) { c->err = std::string(what) + ": " + hipGetErrorString(e); rc = ISDF_ERR_HIP; } };
    chk(hipMemcpyAsync(d_p, xyz, (size_t)3 * n * sizeof(double), hipMemcpyHostToDevice, c->stream), "upload points");
    if (rc == ISDF_OK) rc = scattered ? isdf_esdf_sample_scattered_device(c, d_p, n, value_out ? d_v : nullptr, grad_out ? d_g : nullptr, c->stream)
                                      : isdf_esdf_sample_device(c, d_p, n, value_out ? d_v : nullptr, grad_out ? d_g : nullptr, c->stream);
    if (rc == ISDF_OK && value_out) chk(hipMemcpyAsync(value_out, d_v, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream), "download values");
    if (rc == ISDF_OK && grad_out) chk(hipMemcpyAsync(grad_out, d_g, (size_t)3 * n * sizeof(double), hipMemcpyDeviceToHost, c->stream), "download gradients");
    chk(hipStreamSynchronize(c->stream), "esdf_sample sync");
    return rc;
}

extern "C" int isdf_get_grid(isdf_ctx *c, int grid_kind, void *out, int dtype, int dims_out[3], double origin_out[3], double bmax_out[3]) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    if (!c->have_geom) return isdf_fail(c, ISDF_ERR_STATE, "no grid set");
    HIPCHK(c, hipSetDevice(c->device));
    if (dims_out) { dims_out[0] = c->grid.X; dims_out[1] = c->grid.Y; dims_out[2] = c->grid.Z; }
    for (int a = 0; a < 3; a++) { if (origin_out) origin_out[a] = c->grid.bmin[a]; if (bmax_out) bmax_out[a] = c->grid.bmax[a]; }
    if (!out) return ISDF_OK;
    const size_t n = (size_t)c->grid.X * c->grid.Y * c->grid.Z;
    if (grid_kind == ISDF_GRID_ESDF) {
        if (!c->d_esdf) return isdf_fail(c, ISDF_ERR_STATE, "no ESDF grid");
        if (dtype != ISDF_F32) return isdf_fail(c, ISDF_ERR_UNSUPPORTED, "the ESDF is stored as float32");
        HIPCHK(c, hipMemcpy(out, c->d_esdf, n * sizeof(float), hipMemcpyDeviceToHost));
    } else if (grid_kind == ISDF_GRID_OCCUPANCY) {
        if (!c->d_occ) return isdf_fail(c, ISDF_ERR_STATE, "no occupancy grid");
        if (dtype != ISDF_U8) return isdf_fail(c, ISDF_ERR_UNSUPPORTED, "the occupancy grid is stored as uint8");
        HIPCHK(c, hipMemcpy(out, c->d_occ, n, hipMemcpyDeviceToHost));
    } else return isdf_fail(c, ISDF_ERR_INVALID_ARG, "bad grid kind");
    return ISDF_OK;
}

// projInMap + getGridIndex of a box corner (PCSmap_manager.h:130-137, Gridmap3D.cpp:135-175)
static void corner_index(const DevGrid &G, const double p[3], int idx[3]) {
    const int dim[3] = {G.X, G.Y, G.Z};
    for (int a = 0; a < 3; a++) {
        double v = p[a];
        if (v < G.bmin[a]) v = G.bmin[a];
        if (v > G.bmax[a]) v = G.bmax[a];
        int i = (int)std::floor((v - G.bmin[a]) / G.res);
        if (i < 0) i = 0;
        if (i >= dim[a]) i = dim[a] - 1;
        idx[a] = i;
    }
}

extern "C" int isdf_gather_points(isdf_ctx *c, const double *waypoints, int n_waypoints, const double half[3], const double *offset_opt,
                                  int *M_out) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    if (n_waypoints < 0 || (n_waypoints > 0 && !waypoints) || !half) return isdf_fail(c, ISDF_ERR_INVALID_ARG, "bad waypoint arguments");
    if (!c->have_geom || !c->d_occ) return isdf_fail(c, ISDF_ERR_STATE, "gather_points needs an occupancy grid");
    HIPCHK(c, hipSetDevice(c->device));
    DevGrid G = c->grid;
    G.ZW = (G.Z + 31) / 32;
    const double off[3] = {offset_opt ? offset_opt[0] : 0.0, offset_opt ? offset_opt[1] : 0.0, offset_opt ? offset_opt[2] : 0.0};
    std::vector<GatherBox> boxes(n_waypoints);
    double last[3] = {999, 999, 999};             // tmp_pos of plan_manager.cpp:228
    for (int w = 0; w < n_waypoints; w++) {
        const double *p = waypoints + 3 * w;
        double c1[3], c2[3], l1[3], l2[3];
        for (int a = 0; a < 3; a++) {
            c1[a] = p[a] - half[a] + off[a]; c2[a] = p[a] + half[a] + off[a];     // the offset shifts this box only (:184-185 vs :191-192)
            l1[a] = last[a] - half[a]; l2[a] = last[a] + half[a];
        }
        corner_index(G, c1, boxes[w].lo); corner_index(G, c2, boxes[w].hi);
        corner_index(G, l1, boxes[w].llo); corner_index(G, l2, boxes[w].lhi);
        for (int a = 0; a < 3; a++) last[a] = p[a];
    }
    int M = 0;
    if (c->d_points) { (void)hipFree(c->d_points); c->d_points = nullptr; }
    if (c->d_tstar) { (void)hipFree(c->d_tstar); c->d_tstar = nullptr; }
    c->M = 0;
    if (n_waypoints > 0) {
        const size_t nw = n_bit_words(G);
        const int n_chunks = (int)((nw + GATHER_CHUNK - 1) / GATHER_CHUNK);
        unsigned *bits = nullptr, *marks = nullptr, *chunk = nullptr;
        GatherBox *d_boxes = nullptr;
        int rc = occupancy_bits(c, &bits);
        if (rc) return rc;
        HIPCHK(c, hipMalloc(&marks, nw * sizeof(unsigned)));
        HIPCHK(c, hipMalloc(&chunk, ((size_t)n_chunks + 1) * sizeof(unsigned)));
        HIPCHK(c, hipMalloc(&d_boxes, boxes.size() * sizeof(GatherBox)));
        HIPCHK(c, hipMemsetAsync(marks, 0, nw * sizeof(unsigned), c->stream));
        HIPCHK(c, hipMemcpyAsync(d_boxes, boxes.data(), boxes.size() * sizeof(GatherBox), hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(gather_mark_kernel, dim3(n_waypoints), dim3(256), 0, c->stream, G, d_boxes, bits, marks);
        hipLaunchKernelGGL(gather_count_kernel, dim3(n_chunks), dim3(256), 0, c->stream, marks, nw, chunk);
        hipLaunchKernelGGL(gather_scan_kernel, dim3(1), dim3(256), 0, c->stream, chunk, n_chunks, chunk + n_chunks);
        unsigned total = 0;
        HIPCHK(c, hipMemcpyAsync(&total, chunk + n_chunks, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        M = (int)total;
        if (M > 0) {
            HIPCHK(c, hipMalloc(&c->d_points, (size_t)3 * M * sizeof(double)));
            HIPCHK(c, hipMalloc(&c->d_tstar, (size_t)M * sizeof(double)));
            HIPCHK(c, hipMemsetAsync(c->d_tstar, 0, (size_t)M * sizeof(double), c->stream));     // lastTstar = 0 (plan_manager.cpp:254)
            hipLaunchKernelGGL(gather_emit_kernel, dim3(n_chunks), dim3(64), 0, c->stream, G, marks, nw, chunk, c->d_points, total);
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        (void)hipFree(bits); (void)hipFree(marks); (void)hipFree(chunk); (void)hipFree(d_boxes);
    }
    c->M = M;
    c->points_epoch++;
    if (M_out) *M_out = M;
    ISDF_REPLICATE(c, isdf_gather_points(p_, waypoints, n_waypoints, half, offset_opt, nullptr));      // the same ordered set on every device
    return ISDF_OK;
}

extern "C" int isdf_get_points(isdf_ctx *c, double *xyz_out, int capacity) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    if (capacity < 0 || (capacity > 0 && !xyz_out)) return isdf_fail(c, ISDF_ERR_INVALID_ARG, "bad output buffer");
    HIPCHK(c, hipSetDevice(c->device));
    const int n = c->M < capacity ? c->M : capacity;
    if (n > 0) HIPCHK(c, hipMemcpy(xyz_out, c->d_points, (size_t)3 * n * sizeof(double), hipMemcpyDeviceToHost));
    return c->M;
}
