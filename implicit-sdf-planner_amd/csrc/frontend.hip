// Front end: pose feasibility by kernel convolution.  Integer / bit work on the device:
//   fe_shape_rows_kernel   the robot voxelised at every (roll, pitch) attitude - BasicShape::initShape kernel part
//                          (Shape.hpp:406-462): voxel (a, b, c) is set when getonlySDF(pos, Rx * Ry) <= margin; one lane per
//                          (attitude, a, b) row writes the row's k bits as one dword (bit c = voxel c);
//   fe_map_bits_kernel     the occupancy map inflated by (k - 1) / 2 voxels of free space on every side and bit-packed along z
//                          (PCSmapManager::generateMapKernel, PCSmap_manager.h:46-78), 32 voxels per dword;
//   fe_check_kernel        one WAVEFRONT per query of AstarGetSucc's per-neighbour test (front_end_Astar.hpp:214-217):
//                          every lane keeps the map bits of its <= ceil(k^2 / 64) tile rows in registers (loaded once), then
//                          attitudes are tried in the reference's breadth-first order (a host-built table per parent
//                          attitude, visit_kernels_by_distance sw_manager.hpp:850-909): row dword AND map bits, wave ballot,
//                          stop at the first attitude with no overlap (kernelConv<true>, :813-847).
//   fe_cspace_kernel       the whole configuration space in one pass: 128 bits per voxel and 128 attitudes, bit a = attitude a fits;
//   isdf_frontend_astar_*  the SE(3) A* itself (front_end_Astar.hpp:172-403) on the calling thread over that table: the search
//                          is a chain of dependent pops, the 27 x (up to 121 attitudes x k^2 byte-ANDs) behind every pop is
//                          what the device has already done for every voxel at once.
// The reference's byte layouts (bit z in byte z / 8 under mask 0x80 >> (z % 8)) are produced for read-back only.
// This file is compiled with -ffp-contract=off: the voxelisation compares an SDF with a margin, and robots aligned with
// the grid put voxel centres exactly ON that margin - the comparison must see the reference's roundings.
#include "isdf_ctx.hpp"
#include "dev_mesh.hpp"
#include <hip/hip_ext.h>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <map>
#include <memory>
#include <new>
#include <queue>
#include <vector>

namespace isdf {

struct FeParams {
    int k, n_att, side;          // kernel_size, attitudes, (k - 1) / 2
    double res, margin;
    int iX, iY, iZW;             // inflated map: X + 2h, Y + 2h, dwords per z-row (one spare dword at the end of every row)
    int X, Y, Z;
};

template <int KIND>
__global__ __launch_bounds__(256) void fe_shape_rows_kernel(DevShape S, FeParams F, const double *__restrict__ rot, unsigned *__restrict__ rows) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int kk = F.k * F.k;
    if (t >= F.n_att * kk) return;
    const int att = t / kk, ab = t - att * kk, a = ab / F.k, b = ab - a * F.k;
    const double *R = rot + (size_t)9 * att;
    const double x = F.res * a - F.side * F.res, y = F.res * b - F.side * F.res;       // Shape.hpp:444-446
    unsigned w = 0;
    for (int c = 0; c < F.k; c++) {
        const double z = F.res * c - F.side * F.res;
        if (shape_sdf_rotated<KIND>(S, mk3(x, y, z), R) <= F.margin) w |= 1u << c;
    }
    rows[t] = w;
}

__global__ __launch_bounds__(256) void fe_map_bits_kernel(FeParams F, const uint8_t *__restrict__ occ, unsigned *__restrict__ bits) {
    const size_t n = (size_t)F.iX * F.iY * F.iZW;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        const int w = (int)(t % F.iZW);
        const size_t xy = t / F.iZW;
        const int fy = (int)(xy % F.iY), fx = (int)(xy / F.iY);
        const int x = fx - F.side, y = fy - F.side;
        unsigned v = 0;
        if (x >= 0 && x < F.X && y >= 0 && y < F.Y) {
            const uint8_t *row = occ + ((size_t)x * F.Y + y) * F.Z;
            for (int bit = 0; bit < 32; bit++) {
                const int z = (w << 5) + bit - F.side;
                if (z >= 0 && z < F.Z && row[z] == 1) v |= 1u << bit;
            }
        }
        bits[t] = v;
    }
}

// reference byte layouts for read-back
__global__ __launch_bounds__(256) void fe_pack_shape_bytes_kernel(FeParams F, const unsigned *__restrict__ rows, uint8_t *__restrict__ out) {
    const int bl = (F.k + 7) / 8;
    const int n = F.n_att * F.k * F.k * bl;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int q = t % bl, row = t / bl;
    const unsigned w = rows[row] >> (q * 8);
    unsigned byte = 0;
    for (int bit = 0; bit < 8; bit++) if ((w >> bit) & 1u) byte |= 0x80u >> bit;
    out[t] = (uint8_t)byte;
}
__global__ __launch_bounds__(256) void fe_pack_map_bytes_kernel(FeParams F, const unsigned *__restrict__ bits, uint8_t *__restrict__ out, int zb) {
    const size_t n = (size_t)F.iX * F.iY * zb;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(t % zb);
        const size_t xy = t / zb;
        const unsigned w = bits[xy * F.iZW + (q >> 2)] >> ((q & 3) * 8);
        unsigned byte = 0;
        for (int bit = 0; bit < 8; bit++) if ((w >> bit) & 1u) byte |= 0x80u >> bit;
        out[t] = (uint8_t)byte;
    }
}

constexpr int FE_MAX_ROWS_PER_LANE = 16;     // k <= 31: 961 rows / 64 lanes

struct FeQuery {
    int n, yk, xk;
    double max_roll, max_pitch, ang_res;
    const int *index; const double *fr, *fp;
    uint8_t *ok; double *cr, *cp; int *kidx;
    const unsigned short *seq; const int *seq_len; int seq_stride;
};

__global__ __launch_bounds__(256) void fe_check_kernel(FeParams F, FeQuery Q, const uint8_t *__restrict__ occ, const unsigned *__restrict__ rows,
                                                        const unsigned *__restrict__ bits) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (q >= Q.n) return;
    const int ix = Q.index[3 * q], iy = Q.index[3 * q + 1], iz = Q.index[3 * q + 2];
    const double fr = Q.fr[q], fp = Q.fp[q];
    const int fi = (int)((fr + Q.max_roll) / Q.ang_res), fj = (int)((fp + Q.max_pitch) / Q.ang_res);      // sw_manager.hpp:914-915
    bool good = ix >= 0 && ix < F.X && iy >= 0 && iy < F.Y && iz >= 0 && iz < F.Z;                           // isIndexValid
    good = good && fi >= 0 && fi < Q.xk && fj >= 0 && fj < Q.yk;                                             // the reference indexes visited[][] with these
    if (good) good = occ[((size_t)ix * F.Y + iy) * F.Z + iz] != 1;                                           // !isIndexOccupiedFlate(vi, 0)
    int found = -1;
    if (good) {
        // the tile's map bits: inflated voxel (ix + i, iy + j, iz .. iz + k - 1) for this lane's rows
        const int kk = F.k * F.k;
        const unsigned kmask = (F.k >= 32) ? 0xFFFFFFFFu : ((1u << F.k) - 1u);
        unsigned mb[FE_MAX_ROWS_PER_LANE];
        const int w0 = iz >> 5, sh = iz & 31;
#pragma unroll
        for (int r = 0; r < FE_MAX_ROWS_PER_LANE; r++) {
            const int row = lane + (r << 6);
            unsigned v = 0;
            if (row < kk) {
                const int i = row / F.k, j = row - i * F.k;
                const unsigned *p = bits + ((size_t)(ix + i) * F.iY + (iy + j)) * F.iZW + w0;
                const unsigned b0 = p[0], b1 = p[1];
                v = (sh ? ((b0 >> sh) | (b1 << (32 - sh))) : b0) & kmask;
            }
            mb[r] = v;
        }
        const int start = fi * Q.yk + fj;
        const unsigned short *seq = Q.seq + (size_t)start * Q.seq_stride;
        const int len = Q.seq_len[start];
        for (int t = 0; t < len; t++) {
            const int att = seq[t];
            const unsigned *rw = rows + (size_t)att * kk;
            unsigned hit = 0;
#pragma unroll
            for (int r = 0; r < FE_MAX_ROWS_PER_LANE; r++) {
                const int row = lane + (r << 6);
                if (row < kk) hit |= rw[row] & mb[r];
            }
            if (!__ballot(hit != 0u)) { found = att; break; }
        }
    }
    if (lane == 0) {
        Q.ok[q] = found >= 0 ? 1 : 0;
        if (found >= 0) {
            const int ri = found / Q.yk, rj = found - ri * Q.yk;
            Q.cr[q] = fr + (ri - fi) * Q.ang_res;                                                            // :931-932
            Q.cp[q] = fp + (rj - fj) * Q.ang_res;
        }
        if (Q.kidx) Q.kidx[q] = found;
    }
}

// Whole configuration space at once: for every voxel of the map, which attitudes fit (kernelConv<true> for all of them).
// One lane per voxel, 64 consecutive z per wavefront, so neighbouring lanes read the same map dwords; the robot's rows are
// wave-uniform (scalar loads) and only the NON-EMPTY rows of an attitude are visited (host-built lists: the robot fills a
// small part of its k^3 box).  Output: 4 dwords per voxel and 128 attitudes, bit a = attitude a is collision-free; occupied voxels get 0
// (AstarGetSucc never asks about them).
struct FeRow { unsigned off, word; };      // off = (i * iY + j) * iZW: where the tile row starts relative to the voxel's own row

__global__ __launch_bounds__(256) void fe_cspace_kernel(FeParams F, const uint8_t *__restrict__ occ, const unsigned *__restrict__ bits,
                                                         const FeRow *__restrict__ rows, const int *__restrict__ row_ptr, uint4 *__restrict__ out) {
    const int zblocks = (F.Z + 63) >> 6;
    const long long wv = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long n_wv = (long long)F.X * F.Y * zblocks;
    if (wv >= n_wv) return;
    const int lane = threadIdx.x & 63;
    const int zb = (int)(wv % zblocks);
    const long long xy = wv / zblocks;
    const int y = (int)(xy % F.Y), x = (int)(xy / F.Y);
    const int z = (zb << 6) + lane;
    const bool valid = z < F.Z;
    const int zc = valid ? z : F.Z - 1;
    const bool is_occ = occ[((size_t)x * F.Y + y) * F.Z + zc] == 1;
    const unsigned *base = bits + ((size_t)x * F.iY + y) * F.iZW + (zc >> 5);
    const int sh = zc & 31;
    const unsigned kmask = (F.k >= 32) ? 0xFFFFFFFFu : ((1u << F.k) - 1u);
    const bool work = valid && !is_occ;
    const int nq = (F.n_att + 127) >> 7;                  // 128-attitude groups = uint4 words per voxel
    for (int q = 0; q < nq; q++) {
        unsigned m[4] = {0u, 0u, 0u, 0u};
        const int a_end = min(F.n_att, (q + 1) << 7);
        for (int a = q << 7; a < a_end; a++) {
            const int r0 = row_ptr[a], r1 = row_ptr[a + 1];
            unsigned hit = work ? 0u : 1u;
            for (int r = r0; r < r1; r++) {
                if ((r & 7) == 0 && __ballot(hit == 0u) == 0ull) break;          // every lane has already collided (or has no work)
                const FeRow e = rows[r];                                           // wave-uniform
                const unsigned *p = base + e.off;
                const unsigned b0 = p[0], b1 = p[1];
                const unsigned mb = (sh ? ((b0 >> sh) | (b1 << (32 - sh))) : b0) & kmask;
                hit |= mb & e.word;
            }
            if (hit == 0u) m[(a >> 5) & 3] |= 1u << (a & 31);
        }
        if (valid) out[(((size_t)x * F.Y + y) * F.Z + z) * nq + q] = make_uint4(m[0], m[1], m[2], m[3]);
    }
}

}  // namespace isdf

using namespace isdf;

namespace {

// Eigen::AngleAxisd(rx, UnitX) * Eigen::AngleAxisd(ry, UnitY) -> Matrix3d (Shape.hpp:429): quaternion product, then
// QuaternionBase::toRotationMatrix.  Row-major.
void attitude_rotation(double rx, double ry, double *R) {
    const double aw = std::cos(0.5 * rx), ax = std::sin(0.5 * rx);
    const double bw = std::cos(0.5 * ry), by = std::sin(0.5 * ry);
    const double w = aw * bw - ax * 0.0 - 0.0 * by - 0.0 * 0.0;
    const double x = aw * 0.0 + ax * bw + 0.0 * 0.0 - 0.0 * by;
    const double y = aw * by + 0.0 * bw + 0.0 * 0.0 - ax * 0.0;
    const double z = aw * 0.0 + 0.0 * bw + ax * by - 0.0 * 0.0;
    const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1.0 - (txx + tyy);
}

// the order in which visit_kernels_by_distance (sw_manager.hpp:850-909) tests attitudes when the parent's is (sx, sy):
// the level attitude first, then breadth-first from the parent (neighbours pushed as (0,+1) (0,-1) (+1,0) (-1,0)), the level
// attitude skipped when it comes up again, at most maxdeepth + 1 = 801 queue pops
void bfs_order(int xk, int yk, int sx, int sy, std::vector<unsigned short> &out) {
    out.clear();
    const int zi = (xk - 1) / 2, zj = (yk - 1) / 2;
    out.push_back((unsigned short)(zi * yk + zj));
    std::vector<char> visited((size_t)xk * yk, 0);
    std::queue<std::pair<int, int>> q;
    q.push({sx, sy});
    visited[(size_t)sx * yk + sy] = 1;
    const int dirs[4][2] = {{0, 1}, {0, -1}, {1, 0}, {-1, 0}};
    int deep = 0;
    while (!q.empty()) {
        deep++;
        const int x = q.front().first, y = q.front().second;
        q.pop();
        if (x != zi || y != zj) out.push_back((unsigned short)(x * yk + y));
        for (const auto &d : dirs) {
            const int nx = x + d[0], ny = y + d[1];
            if (nx < 0 || nx >= xk || ny < 0 || ny >= yk || visited[(size_t)nx * yk + ny]) continue;
            visited[(size_t)nx * yk + ny] = 1;
            q.push({nx, ny});
        }
        if (deep > 800) break;
    }
}

void fe_free(isdf_ctx *c) {
    if (c->fe.h_cspace) { if (c->fe.h_cspace_pinned) (void)hipHostFree(c->fe.h_cspace); else std::free(c->fe.h_cspace); }
    void *ptrs[] = {c->fe.d_rows, c->fe.d_bits, c->fe.d_rot, c->fe.d_seq, c->fe.d_seq_len, c->fe.d_row_list, c->fe.d_row_ptr, c->fe.d_cspace};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    c->fe = isdf_ctx::FrontEnd{};
}

FeParams fe_params(const isdf_ctx *c) {
    FeParams F{};
    F.k = c->fe.cfg.kernel_size; F.n_att = c->fe.xk * c->fe.yk; F.side = (F.k - 1) / 2;
    F.res = c->grid.res; F.margin = c->fe.margin;
    F.X = c->grid.X; F.Y = c->grid.Y; F.Z = c->grid.Z;
    F.iX = F.X + 2 * F.side; F.iY = F.Y + 2 * F.side; F.iZW = (F.Z + 2 * F.side + 31) / 32 + 1;
    return F;
}

}  // namespace

void isdf_frontend_release(isdf_ctx *c) { fe_free(c); }

extern "C" int isdf_frontend_build(isdf_ctx *c, const isdf_frontend_config *cfg) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    if (!cfg) return isdf_fail(c, ISDF_ERR_INVALID_ARG, "null front-end config");
    if (cfg->kernel_size < 1 || cfg->kernel_size > 31 || (cfg->kernel_size & 1) == 0) return isdf_fail(c, ISDF_ERR_INVALID_ARG, "kernel_size must be odd and <= 31");
    if (!(cfg->kernel_ang_res > 0.0) || cfg->kernel_max_roll < 0.0 || cfg->kernel_max_pitch < 0.0) return isdf_fail(c, ISDF_ERR_INVALID_ARG, "bad attitude grid");
    if (!c->have_shape) return isdf_fail(c, ISDF_ERR_STATE, "isdf_set_shape has not been called");
    if (!c->have_geom || !c->d_occ) return isdf_fail(c, ISDF_ERR_STATE, "the front end needs an occupancy grid");
    HIPCHK(c, hipSetDevice(c->device));
    fe_free(c);
    isdf_ctx::FrontEnd &fe = c->fe;
    fe.cfg = *cfg;
    fe.xk = (int)std::floor(2 * cfg->kernel_max_roll / cfg->kernel_ang_res) + 1;        // Shape.hpp:297-298
    fe.yk = (int)std::floor(2 * cfg->kernel_max_pitch / cfg->kernel_ang_res) + 1;
    if ((long long)fe.xk * fe.yk > 65535) return isdf_fail(c, ISDF_ERR_UNSUPPORTED, "more than 65535 attitudes");
    fe.margin = std::max(cfg->front_end_safeh, c->grid.res / 2);                          // Shape.hpp:415
    const int n_att = fe.xk * fe.yk;
    // attitudes in the order of the reference's accumulating loops (:416-421); slots its loops never reach stay empty
    std::vector<double> rot((size_t)n_att * 9, 0.0);
    std::vector<char> reached((size_t)n_att, 0);
    int i = 0;
    for (double roll = -cfg->kernel_max_roll; roll <= cfg->kernel_max_roll; roll += cfg->kernel_ang_res, i++) {
        int j = 0;
        for (double pitch = -cfg->kernel_max_pitch; pitch <= cfg->kernel_max_pitch; pitch += cfg->kernel_ang_res, j++) {
            if (i >= fe.xk || j >= fe.yk) continue;
            attitude_rotation(roll * M_PI / 180.0, pitch * M_PI / 180.0, rot.data() + (size_t)(i * fe.yk + j) * 9);
            reached[(size_t)i * fe.yk + j] = 1;
        }
    }
    const FeParams F = fe_params(c);
    HIPCHK(c, hipMalloc(&fe.d_rot, rot.size() * sizeof(double)));
    HIPCHK(c, hipMemcpy(fe.d_rot, rot.data(), rot.size() * sizeof(double), hipMemcpyHostToDevice));
    const size_t n_rows = (size_t)n_att * F.k * F.k;
    HIPCHK(c, hipMalloc(&fe.d_rows, n_rows * sizeof(unsigned)));
    {
        const dim3 grid((unsigned)((n_rows + 255) / 256)), block(256);
        if (c->shape.kind == ISDF_SHAPE_MESH) hipLaunchKernelGGL(fe_shape_rows_kernel<ISDF_SHAPE_MESH>, grid, block, 0, c->stream, c->shape, F, fe.d_rot, fe.d_rows);
        else hipLaunchKernelGGL(fe_shape_rows_kernel<-1>, grid, block, 0, c->stream, c->shape, F, fe.d_rot, fe.d_rows);
        HIPCHK(c, hipGetLastError());
    }
    // attitudes the reference's loops never initialise hold no voxels (an all-zero kernel never collides)
    for (int a = 0; a < n_att; a++)
        if (!reached[a]) HIPCHK(c, hipMemsetAsync(fe.d_rows + (size_t)a * F.k * F.k, 0, (size_t)F.k * F.k * sizeof(unsigned), c->stream));
    const size_t n_bits = (size_t)F.iX * F.iY * F.iZW;
    HIPCHK(c, hipMalloc(&fe.d_bits, (n_bits + 2) * sizeof(unsigned)));
    HIPCHK(c, hipMemsetAsync(fe.d_bits + n_bits, 0, 2 * sizeof(unsigned), c->stream));
    hipLaunchKernelGGL(fe_map_bits_kernel, dim3(2048), dim3(256), 0, c->stream, F, c->d_occ, fe.d_bits);
    HIPCHK(c, hipGetLastError());
    // breadth-first order tables
    std::vector<std::vector<unsigned short>> orders(n_att);
    size_t stride = 1;
    for (int sx = 0; sx < fe.xk; sx++)
        for (int sy = 0; sy < fe.yk; sy++) { bfs_order(fe.xk, fe.yk, sx, sy, orders[(size_t)sx * fe.yk + sy]); stride = std::max(stride, orders[(size_t)sx * fe.yk + sy].size()); }
    std::vector<unsigned short> seq((size_t)n_att * stride, 0);
    std::vector<int> seq_len(n_att);
    for (int a = 0; a < n_att; a++) { std::copy(orders[a].begin(), orders[a].end(), seq.begin() + (size_t)a * stride); seq_len[a] = (int)orders[a].size(); }
    fe.seq_stride = (int)stride;
    fe.h_seq = seq; fe.h_seq_len = seq_len;
    HIPCHK(c, hipMalloc(&fe.d_seq, seq.size() * sizeof(unsigned short)));
    HIPCHK(c, hipMalloc(&fe.d_seq_len, seq_len.size() * sizeof(int)));
    HIPCHK(c, hipMemcpyAsync(fe.d_seq, seq.data(), seq.size() * sizeof(unsigned short), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(fe.d_seq_len, seq_len.data(), seq_len.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // non-empty rows of every attitude (for the whole-map pass)
    {
        std::vector<unsigned> rows_h(n_rows);
        HIPCHK(c, hipMemcpy(rows_h.data(), fe.d_rows, n_rows * sizeof(unsigned), hipMemcpyDeviceToHost));
        std::vector<FeRow> list;
        std::vector<int> ptr(n_att + 1, 0);
        for (int a = 0; a < n_att; a++) {
            for (int ii = 0; ii < F.k; ii++)
                for (int jj = 0; jj < F.k; jj++) {
                    const unsigned w = rows_h[((size_t)a * F.k + ii) * F.k + jj];
                    if (w) list.push_back(FeRow{(unsigned)(((size_t)ii * F.iY + jj) * F.iZW), w});
                }
            ptr[a + 1] = (int)list.size();
        }
        fe.n_row_list = (int)list.size();
        HIPCHK(c, hipMalloc(&fe.d_row_list, std::max<size_t>(1, list.size()) * sizeof(FeRow)));
        HIPCHK(c, hipMalloc(&fe.d_row_ptr, ptr.size() * sizeof(int)));
        if (!list.empty()) HIPCHK(c, hipMemcpy(fe.d_row_list, list.data(), list.size() * sizeof(FeRow), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(fe.d_row_ptr, ptr.data(), ptr.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    fe.built = true;
    return ISDF_OK;
}

extern "C" int isdf_frontend_cspace(isdf_ctx *c, uint32_t *free_mask_out, double *kernel_ms_out) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    if (!c->fe.built) return isdf_fail(c, ISDF_ERR_STATE, "isdf_frontend_build has not been called");
    const FeParams F = fe_params(c);
    const size_t nw = 4 * (size_t)((F.n_att + 127) / 128);             // dwords per voxel
    HIPCHK(c, hipSetDevice(c->device));
    const size_t n = (size_t)F.X * F.Y * F.Z;
    if (!c->fe.d_cspace) HIPCHK(c, hipMalloc(&c->fe.d_cspace, n * nw * sizeof(unsigned)));
    hipEvent_t e0, e1;
    HIPCHK(c, hipEventCreate(&e0)); HIPCHK(c, hipEventCreate(&e1));
    const long long n_wv = (long long)F.X * F.Y * ((F.Z + 63) >> 6);
    hipExtLaunchKernelGGL(fe_cspace_kernel, dim3((unsigned)((n_wv + 3) / 4)), dim3(256), 0, c->stream, e0, e1, 0, F, c->d_occ, c->fe.d_bits,
                          (const FeRow *)c->fe.d_row_list, c->fe.d_row_ptr, (uint4 *)c->fe.d_cspace);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && free_mask_out) e = hipMemcpyAsync(free_mask_out, c->fe.d_cspace, n * nw * sizeof(unsigned), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    HIPCHK(c, e);
    if (kernel_ms_out) *kernel_ms_out = ms;
    return ISDF_OK;
}

extern "C" int isdf_frontend_get_shape_kernels(isdf_ctx *c, uint8_t *out, int dims_out[3]) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    if (!c->fe.built) return isdf_fail(c, ISDF_ERR_STATE, "isdf_frontend_build has not been called");
    const FeParams F = fe_params(c);
    const int bl = (F.k + 7) / 8, per = F.k * F.k * bl;
    if (dims_out) { dims_out[0] = c->fe.xk; dims_out[1] = c->fe.yk; dims_out[2] = per; }
    if (!out) return ISDF_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t n = (size_t)F.n_att * per;
    uint8_t *d = nullptr;
    HIPCHK(c, hipMalloc(&d, n));
    hipLaunchKernelGGL(fe_pack_shape_bytes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, F, c->fe.d_rows, d);
    hipError_t e = hipMemcpyAsync(out, d, n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    HIPCHK(c, e);
    return ISDF_OK;
}

extern "C" int isdf_frontend_get_map_kernel(isdf_ctx *c, uint8_t *out, int dims_out[3]) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    if (!c->fe.built) return isdf_fail(c, ISDF_ERR_STATE, "isdf_frontend_build has not been called");
    const FeParams F = fe_params(c);
    const int zb = (F.Z + 2 * F.side + 7) / 8;
    if (dims_out) { dims_out[0] = F.iX; dims_out[1] = F.iY; dims_out[2] = zb; }
    if (!out) return ISDF_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t n = (size_t)F.iX * F.iY * zb;
    uint8_t *d = nullptr;
    HIPCHK(c, hipMalloc(&d, n));
    hipLaunchKernelGGL(fe_pack_map_bytes_kernel, dim3(2048), dim3(256), 0, c->stream, F, c->fe.d_bits, d, zb);
    hipError_t e = hipMemcpyAsync(out, d, n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    HIPCHK(c, e);
    return ISDF_OK;
}

extern "C" int isdf_frontend_check(isdf_ctx *c, int n, const int32_t *index, const double *father_roll, const double *father_pitch,
                                   uint8_t *ok, double *child_roll, double *child_pitch, int32_t *kernel_index_out) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    if (n < 0 || (n > 0 && (!index || !father_roll || !father_pitch || !ok || !child_roll || !child_pitch))) return isdf_fail(c, ISDF_ERR_INVALID_ARG, "bad front-end query");
    if (!c->fe.built) return isdf_fail(c, ISDF_ERR_STATE, "isdf_frontend_build has not been called");
    if (n == 0) return ISDF_OK;
    HIPCHK(c, hipSetDevice(c->device));
    const FeParams F = fe_params(c);
    // one allocation: [index 3n int | fr n | fp n | cr n | cp n | kidx n int | ok n]
    const size_t bytes = (size_t)n * (3 * 4 + 8 * 4 + 4 + 1) + 64;
    char *d = nullptr;
    HIPCHK(c, hipMalloc(&d, bytes));
    double *d_fr = (double *)d, *d_fp = d_fr + n, *d_cr = d_fp + n, *d_cp = d_cr + n;
    int *d_idx = (int *)(d_cp + n), *d_kidx = d_idx + (size_t)3 * n;
    uint8_t *d_ok = (uint8_t *)(d_kidx + n);
    int rc = ISDF_OK;
    auto chk = [&](hipError_t e, const char *what) { if (e != hipSuccess && rc == ISDF_OK) { c->err = std::string(what) + ": " + hipGetErrorString(e); rc = ISDF_ERR_HIP; } };
    chk(hipMemcpyAsync(d_idx, index, (size_t)3 * n * sizeof(int), hipMemcpyHostToDevice, c->stream), "upload index");
    chk(hipMemcpyAsync(d_fr, father_roll, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream), "upload roll");
    chk(hipMemcpyAsync(d_fp, father_pitch, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream), "upload pitch");
    if (rc == ISDF_OK) {
        FeQuery Q{};
        Q.n = n; Q.xk = c->fe.xk; Q.yk = c->fe.yk;
        Q.max_roll = c->fe.cfg.kernel_max_roll; Q.max_pitch = c->fe.cfg.kernel_max_pitch; Q.ang_res = c->fe.cfg.kernel_ang_res;
        Q.index = d_idx; Q.fr = d_fr; Q.fp = d_fp; Q.ok = d_ok; Q.cr = d_cr; Q.cp = d_cp; Q.kidx = d_kidx;
        Q.seq = c->fe.d_seq; Q.seq_len = c->fe.d_seq_len; Q.seq_stride = c->fe.seq_stride;
        hipLaunchKernelGGL(fe_check_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, c->stream, F, Q, c->d_occ, c->fe.d_rows, c->fe.d_bits);
        chk(hipGetLastError(), "fe_check_kernel");
    }
    if (rc == ISDF_OK) chk(hipMemcpyAsync(ok, d_ok, (size_t)n, hipMemcpyDeviceToHost, c->stream), "download ok");
    if (rc == ISDF_OK) chk(hipMemcpyAsync(child_roll, d_cr, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream), "download roll");
    if (rc == ISDF_OK) chk(hipMemcpyAsync(child_pitch, d_cp, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream), "download pitch");
    if (rc == ISDF_OK && kernel_index_out) chk(hipMemcpyAsync(kernel_index_out, d_kidx, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, c->stream), "download index");
    chk(hipStreamSynchronize(c->stream), "front-end sync");
    (void)hipFree(d);
    return rc;
}

// ---- the SE(3) A* (AstarPathSearcher, front_end_Astar.hpp:172-403) -------------------------------------------------------
namespace {

// GridNode (:12-57) without what the voxel index already says and without fScore (the reference reads a node's fScore only where it
// inserts the node into the open set - the key is passed along instead).  All-zero = a fresh node: `father1` is the father's voxel
// index + 1, A_START1 the start node object, 0 none.
struct ANode { double g, roll, pitch; int father1; int id; };
constexpr int A_START = -2, A_START1 = -1;
constexpr int A_PAGE_BITS = 10;

// GridNodeMap (:127-152) allocates a node for every voxel of the map and reset() (:160-170) walks all of them after every
// search; here a page of 1 024 nodes appears (zeroed) when the search first touches it, and a fresh table is the reset
struct ANodes {
    std::vector<ANode *> pages;
    explicit ANodes(size_t n) : pages((n >> A_PAGE_BITS) + 1, nullptr) {}
    ~ANodes() { for (ANode *p : pages) std::free(p); }
    ANodes(const ANodes &) = delete;
    ANodes &operator=(const ANodes &) = delete;
    ANode &at(size_t i) {
        ANode *&p = pages[i >> A_PAGE_BITS];
        if (!p) { p = (ANode *)std::calloc((size_t)1 << A_PAGE_BITS, sizeof(ANode)); if (!p) throw std::bad_alloc(); }
        return p[i & (((size_t)1 << A_PAGE_BITS) - 1)];
    }
};

// The open set: the reference's std::multimap<double, GridNode *> pops its begin() - the smallest key, and among equal keys the one
// inserted first (equal keys are inserted at the end of their range).  A binary heap ordered by (key, insertion number) pops in
// exactly that order without a tree node per entry.
struct AOpen {
    struct E { double f; unsigned long long seq; int idx; };
    std::vector<E> h;
    unsigned long long n_ins = 0;
    static bool before(const E &a, const E &b) { return a.f < b.f || (a.f == b.f && a.seq < b.seq); }
    bool empty() const { return h.empty(); }
    void push(double f, int idx) {
        h.push_back(E{f, n_ins++, idx});
        size_t i = h.size() - 1;
        while (i > 0) { const size_t p = (i - 1) >> 1; if (!before(h[i], h[p])) break; std::swap(h[i], h[p]); i = p; }
    }
    int pop() {
        const int idx = h[0].idx;
        h[0] = h.back(); h.pop_back();
        const size_t n = h.size();
        for (size_t i = 0;;) {
            size_t l = 2 * i + 1, r = l + 1, m = i;
            if (l < n && before(h[l], h[m])) m = l;
            if (r < n && before(h[r], h[m])) m = r;
            if (m == i) break;
            std::swap(h[i], h[m]); i = m;
        }
        return idx;
    }
};

// getHeu (:172-188)
inline double astar_heu(const int a[3], const int b[3]) {
    const double p = 1.0 / 1000;
    const int dx = std::abs(a[0] - b[0]), dy = std::abs(a[1] - b[1]), dz = std::abs(a[2] - b[2]);
    const int dmin = std::min(dx, std::min(dy, dz)), dmax = std::max(dx, std::max(dy, dz));
    const int dmid = dx + dy + dz - dmin - dmax;
    const double h = std::sqrt(3.0) * dmin + std::sqrt(2.0) * (dmid - dmin) + (dmax - dmid);
    return h * (1 + p);
}

// GridMap3D::isInMap / getGridIndex (Gridmap3D.cpp:41-69,135-175; an in-map position cannot reach the branches with the typos)
inline bool astar_in_map(const DevGrid &G, const double p[3]) {
    for (int a = 0; a < 3; a++) if (p[a] < G.bmin[a]) return false;
    for (int a = 0; a < 3; a++) if (p[a] > G.bmax[a]) return false;
    return true;
}
inline void astar_index(const DevGrid &G, const double p[3], int idx[3]) {
    const int dims[3] = {G.X, G.Y, G.Z};
    for (int a = 0; a < 3; a++) {
        const double d = p[a] - G.bmin[a];
        int i = (int)std::floor(d / G.res);
        if (i < 0) i = 0;
        if (i >= dims[a]) i = dims[a] - 1;
        idx[a] = i;
    }
}

}  // namespace

extern "C" int isdf_frontend_astar_search(isdf_ctx *c, const double start[3], const double goal[3], isdf_astar_result *result) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    if (!start || !goal || !result) return isdf_fail(c, ISDF_ERR_INVALID_ARG, "null start / goal / result");
    if (!c->fe.built) return isdf_fail(c, ISDF_ERR_STATE, "isdf_frontend_build has not been called");
    isdf_ctx::FrontEnd &fe = c->fe;
    const DevGrid &G = c->grid;
    const size_t n_vox = (size_t)G.X * G.Y * G.Z;
    if (n_vox > (size_t)0x7FFFFFF0) return isdf_fail(c, ISDF_ERR_UNSUPPORTED, "the A* indexes voxels with 31 bits");
    *result = isdf_astar_result{};
    fe.path_xyz.clear(); fe.path_rp.clear();
    const size_t nw = 4 * (size_t)((fe.xk * fe.yk + 127) / 128);       // dwords of a voxel's word in the table
    typedef std::chrono::steady_clock clk;
    if (!fe.h_cspace_valid) {
        const clk::time_point t0 = clk::now();
        HIPCHK(c, hipSetDevice(c->device));
        if (!fe.h_cspace) {
            // 16 bytes per voxel and 128 attitudes (268 MB at 256^3, 2 GiB at 512^3): pinned if the host grants it, pageable otherwise (the copy is then staged by the runtime)
            if (hipHostMalloc((void **)&fe.h_cspace, n_vox * nw * sizeof(unsigned), hipHostMallocDefault) == hipSuccess) fe.h_cspace_pinned = true;
            else {
                (void)hipGetLastError();
                fe.h_cspace = (unsigned *)std::malloc(n_vox * nw * sizeof(unsigned));
                fe.h_cspace_pinned = false;
                if (!fe.h_cspace) return isdf_fail(c, ISDF_ERR_HIP, "no host memory for the configuration-space table");
            }
        }
        double ms = 0.0;
        const int rc = isdf_frontend_cspace(c, fe.h_cspace, &ms);
        if (rc != ISDF_OK) return rc;
        fe.h_cspace_valid = true;
        result->cspace_ms = ms;
        result->table_ms = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
    }
    const clk::time_point t1 = clk::now();
    if (!astar_in_map(G, start) || !astar_in_map(G, goal)) return ISDF_OK;                 // :244-249: success_flag = false
    int si[3], gi[3];
    astar_index(G, start, si);
    astar_index(G, goal, gi);
    const int xk = fe.xk, yk = fe.yk;
    const double max_roll = fe.cfg.kernel_max_roll, max_pitch = fe.cfg.kernel_max_pitch, ang_res = fe.cfg.kernel_ang_res;
    const unsigned *table = fe.h_cspace;
    const unsigned short *seq = fe.h_seq.data();
    const int *seq_len = fe.h_seq_len.data();
    const size_t seq_stride = (size_t)fe.seq_stride;
    const size_t YZ = (size_t)G.Y * G.Z;
    double edge[4];                                                                         // sqrt(i*i + j*j + k*k), :230
    for (int q = 0; q < 4; q++) edge[q] = std::sqrt((double)q);

    try {
    ANodes nodes(n_vox);
    // the start node is an object of its own (`new GridNode`, :260): IT sits in the open set and is the first successors' father;
    // the map's node of the start cell is only marked open (:276-278) and stays that way
    ANode start_node{};
    start_node.g = 0.0; start_node.id = 1;
    AOpen open;
    open.push(astar_heu(si, gi), A_START);
    {
        ANode &m = nodes.at((size_t)si[0] * YZ + (size_t)si[1] * G.Z + si[2]);
        m.id = 1; m.g = start_node.g;
    }
    int terminate = -1;
    bool found = false;
    long long expansions = 0, checks = 0;
    while (!open.empty()) {
        const int cur = open.pop();
        ANode &cn = cur == A_START ? start_node : nodes.at((size_t)cur);
        cn.id = -1;
        expansions++;
        int ci[3];
        if (cur == A_START) { ci[0] = si[0]; ci[1] = si[1]; ci[2] = si[2]; }
        else { ci[2] = (int)((size_t)cur % G.Z); ci[1] = (int)(((size_t)cur / G.Z) % G.Y); ci[0] = (int)((size_t)cur / YZ); }
        if (ci[0] == gi[0] && ci[1] == gi[1] && ci[2] == gi[2]) { terminate = cur; found = true; break; }          // :289-297
        // AstarGetSucc (:197-236): the parent's attitude picks the breadth-first order (checkKernelValue, sw_manager.hpp:911-942)
        const double fr = cn.roll, fp = cn.pitch, gcur = cn.g;
        const int fi = (int)((fr + max_roll) / ang_res), fj = (int)((fp + max_pitch) / ang_res);
        const bool parent_on_grid = fi >= 0 && fi < xk && fj >= 0 && fj < yk;             // (the reference indexes visited[][] with these)
        const unsigned short *order = seq + (size_t)(parent_on_grid ? fi * yk + fj : 0) * seq_stride;
        const int order_len = parent_on_grid ? seq_len[fi * yk + fj] : 0;
        int succ[27]; double succ_ec[27];
        int ns = 0;
        for (int i = -1; i < 2; i++)
            for (int j = -1; j < 2; j++)
                for (int k = -1; k < 2; k++) {
                    checks++;
                    const int vx = ci[0] + i, vy = ci[1] + j, vz = ci[2] + k;
                    if (vx < 0 || vx >= G.X || vy < 0 || vy >= G.Y || vz < 0 || vz >= G.Z) continue;               // isIndexValid
                    const size_t v = (size_t)vx * YZ + (size_t)vy * G.Z + vz;
                    const unsigned *m = table + nw * v;                                                               // occupied voxels hold 0
                    unsigned any = m[0] | m[1] | m[2] | m[3];
                    for (size_t w = 4; w < nw; w++) any |= m[w];
                    if (any == 0u) continue;
                    int att = -1;
                    for (int t = 0; t < order_len; t++) {
                        const int a = order[t];
                        if ((m[a >> 5] >> (a & 31)) & 1u) { att = a; break; }
                    }
                    if (att < 0) continue;
                    const int ri = att / yk, rj = att - ri * yk;
                    ANode &p = nodes.at(v);
                    p.roll = fr + (ri - fi) * ang_res;                                                                // sw_manager.hpp:931-932; Astar :227-228:
                    p.pitch = fp + (rj - fj) * ang_res;                                                               // written whatever the node's state
                    succ[ns] = (int)v; succ_ec[ns] = edge[i * i + j * j + k * k]; ns++;
                }
        for (int s2 = 0; s2 < ns; s2++) {
            ANode &nb = nodes.at((size_t)succ[s2]);
            const double tg = succ_ec[s2] + gcur;
            if (nb.id == 1 && !(tg < nb.g)) continue;
            if (nb.id == -1 && !(tg < nb.g)) continue;
            const int ni[3] = {(int)((size_t)succ[s2] / YZ), (int)(((size_t)succ[s2] / G.Z) % G.Y), (int)((size_t)succ[s2] % G.Z)};
            nb.father1 = cur == A_START ? A_START1 : cur + 1; nb.g = tg;
            if (nb.id != 1) {                                       // new (:306-317) or closed and reopened (:330-343); an open node keeps its OLD key (:319-328)
                nb.id = 1;
                open.push(tg + astar_heu(ni, gi) + 0.0, succ[s2]);                                                   // + getCustomCost (:192-196) = 0
            }
        }
    }
    result->expansions = expansions; result->checks = checks;
    if (found) {
        // getPath (:359-376) and the (roll, pitch) chain of getastarSE3Path (:378-403): fathers back to the start node, reversed
        std::vector<int> chain;
        for (int p = terminate;;) {
            chain.push_back(p);
            const int f1 = p == A_START ? 0 : nodes.at((size_t)p).father1;          // (the start node object has no father)
            if (f1 == 0) break;
            p = f1 == A_START1 ? A_START : f1 - 1;
        }
        for (size_t q = chain.size(); q-- > 0;) {
            int ix, iy, iz;
            const ANode *n;
            if (chain[q] == A_START) { ix = si[0]; iy = si[1]; iz = si[2]; n = &start_node; }
            else { const size_t v = (size_t)chain[q]; ix = (int)(v / YZ); iy = (int)((v / G.Z) % G.Y); iz = (int)(v % G.Z); n = &nodes.at(v); }
            fe.path_xyz.push_back((ix + 0.5) * G.res + G.bmin[0]);                                                   // getGridCubeCenter, Gridmap3D.cpp:182-194
            fe.path_xyz.push_back((iy + 0.5) * G.res + G.bmin[1]);
            fe.path_xyz.push_back((iz + 0.5) * G.res + G.bmin[2]);
            fe.path_rp.push_back(n->roll); fe.path_rp.push_back(n->pitch);
        }
        result->success = 1;
        result->n_path = (int)chain.size();
    }
    } catch (const std::bad_alloc &) {
        fe.path_xyz.clear(); fe.path_rp.clear();
        *result = isdf_astar_result{};
        return isdf_fail(c, ISDF_ERR_HIP, "the A* ran out of host memory");
    }
    result->search_ms = std::chrono::duration<double, std::milli>(clk::now() - t1).count();
    return ISDF_OK;
}

extern "C" int isdf_frontend_astar_path(isdf_ctx *c, int capacity, double *xyz, double *roll_pitch, double *rot) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    if (capacity < 0) return isdf_fail(c, ISDF_ERR_INVALID_ARG, "negative capacity");
    const int n = (int)(c->fe.path_rp.size() / 2);
    for (int i = 0; i < n && i < capacity; i++) {
        if (xyz) for (int k = 0; k < 3; k++) xyz[3 * i + k] = c->fe.path_xyz[3 * (size_t)i + k];
        const double roll = c->fe.path_rp[2 * (size_t)i], pitch = c->fe.path_rp[2 * (size_t)i + 1];
        if (roll_pitch) { roll_pitch[2 * i] = roll; roll_pitch[2 * i + 1] = pitch; }
        if (rot) attitude_rotation(roll * M_PI / 180.0, pitch * M_PI / 180.0, rot + 9 * (size_t)i);                   // :386-388
    }
    return n;
}
