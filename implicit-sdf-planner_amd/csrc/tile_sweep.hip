// V2 / V3 integral sweep on gfx950: the P x (K+1) constraint-sample loop of
//   addTimeIntPenaltyParallel   (back_end_optimizer.hpp:432-554)
// with the per-pose tile term grad_cost_p (:766-824) plugged in before the flatness back-prop (:536), the voxel
// qualifier being occupancy (V2) or esdf <= occ_thresh (V3).
//
// Mapping (DESIGN.md "tile kernel"):
//   * one WAVEFRONT per constraint sample (piece i, sample j); a 256-thread workgroup = 4 consecutive samples.
//   * phase 0: every wave computes its pose (fp64) and its tile index box exactly as getPointsInAABB does.
//   * phase 1 (staging): the workgroup walks the UNION of its samples' tiles once with coalesced z-row reads of
//     the grid and stages the tile into LDS as a compacted list of qualifying voxels (ballot/popcount compaction,
//     per-wave segments => deterministic order, no atomics).  Consecutive samples' tiles overlap >90 %, so the tile
//     is fetched once per workgroup instead of once per sample.
//   * phase 2: each wave streams the LDS list for its own pose: body-frame transform, |p_rel| cull, robot SDF;
//     the few voxels with an active penalty are compacted again (per-wave LDS list) and get the 6 extra SDF
//     evaluations of the central-difference gradient in dense rounds.
//   * phase 3: wave butterfly reduction of (cost, dL/dpos, dL/dquat), dynamics penalties, flatness reverse mode,
//     20-double partial record per sample; tile_reduce_kernel sums the records per piece in a fixed order.
// Everything on the data path is fp64 except the stored ESDF (fp32).  No MFMA: gather/interpolate work.
#include "isdf_internal.hpp"
#include "dev_mesh.hpp"

namespace isdf {

constexpr int TS_WAVES = 4;
constexpr int TS_THREADS = 64 * TS_WAVES;
constexpr int TS_CHUNK = 4096;                 // union-tile voxels staged per round
constexpr int TS_SEG = TS_CHUNK / TS_WAVES;    // per-wave list segment (capacity == voxels it stages)
constexpr int TS_ROUNDS = TS_SEG / 64;
constexpr int TS_NEAR_CAP = 128;

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ bool in_map(const DevGrid &G, double x, double y, double z) {
    if (x < G.bmin[0] || y < G.bmin[1] || z < G.bmin[2]) return false;
    if (x > G.bmax[0] || y > G.bmax[1] || z > G.bmax[2]) return false;
    return true;
}
// GridMap3D::getGridIndex (Gridmap3D.cpp:135-175) for a point already known to be inside the map.
__device__ __forceinline__ int axis_index(double v, double bmin, double res, int n) {
    int i = (int)floor((v - bmin) / res);
    if (i < 0) i = 0;
    if (i >= n) i = n - 1;
    return i;
}
__device__ __forceinline__ double clampd(double v, double lo, double hi) {   // projInMap, PCSmap_manager.h:130-137
    if (v < lo) v = lo;
    if (v > hi) v = hi;
    return v;
}

// trilinear ESDF value at pos (GridMap3D.h:155-193; invalid corner -> 0, Gridmap3D.cpp:535-542)
__device__ __forceinline__ double esdf_trilinear(const DevGrid &G, d3 pos) {
    const double hx = pos.x - 0.5 * G.res, hy = pos.y - 0.5 * G.res, hz = pos.z - 0.5 * G.res;
    int ix = 0, iy = 0, iz = 0;
    if (in_map(G, hx, hy, hz)) {
        ix = axis_index(hx, G.bmin[0], G.res, G.X);
        iy = axis_index(hy, G.bmin[1], G.res, G.Y);
        iz = axis_index(hz, G.bmin[2], G.res, G.Z);
    }
    const double cx = (ix + 0.5) * G.res + G.bmin[0], cy = (iy + 0.5) * G.res + G.bmin[1], cz = (iz + 0.5) * G.res + G.bmin[2];
    const double inv = 1.0 / G.res;
    const double dx = (pos.x - cx) * inv, dy = (pos.y - cy) * inv, dz = (pos.z - cz) * inv;
    double v[2][2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const int x = ix + a, y = iy + b, z = iz + c;
                const bool ok = x < G.X && y < G.Y && z < G.Z;   // lower bounds hold by construction
                v[a][b][c] = ok ? (double)G.esdf[((size_t)x * G.Y + y) * G.Z + z] : 0.0;
            }
    const double v00 = (1 - dx) * v[0][0][0] + dx * v[1][0][0];
    const double v01 = (1 - dx) * v[0][0][1] + dx * v[1][0][1];
    const double v10 = (1 - dx) * v[0][1][0] + dx * v[1][1][0];
    const double v11 = (1 - dx) * v[0][1][1] + dx * v[1][1][1];
    const double v0 = (1 - dy) * v00 + dy * v10;
    const double v1 = (1 - dy) * v01 + dy * v11;
    return (1 - dz) * v0 + dz * v1;
}

// exact floor(v / d) for v < 2^24 (float reciprocal + one correction step)
__device__ __forceinline__ unsigned fast_div(unsigned v, unsigned d, float inv, bool small) {
    if (!small) return v / d;
    unsigned q = (unsigned)((float)v * inv);
    const int r = (int)(v - q * d);
    if (r < 0) q -= 1;
    else if (r >= (int)d) q += 1;
    return q;
}

struct Acc { double costp, gpx, gpy, gpz, gqw, gqx, gqy, gqz; };

template <int KIND>
__device__ __forceinline__ void near_process(const SweepParams &P, const unsigned *near, int count, int lane, const int *ulo,
                                             d3 pos, const m3 &R, d4 quat, Acc &acc, unsigned &n_grad) {
    for (int base = 0; base < count; base += 64) {
        const int e = base + lane;
        if (e < count) {
            const unsigned ent = near[e];
            const int gx = ulo[0] + (int)(ent & 1023u), gy = ulo[1] + (int)((ent >> 10) & 1023u), gz = ulo[2] + (int)(ent >> 20);
            const d3 c = mk3((gx + 0.5) * P.grid.res + P.grid.bmin[0], (gy + 0.5) * P.grid.res + P.grid.bmin[1],
                             (gz + 0.5) * P.grid.res + P.grid.bmin[2]);
            const d3 pm = c - pos;
            const d3 prel = mat_tmul(R, pm);
            const double sdf = shape_sdf<KIND>(P.shape, prel);
            double f = 0.0, df = 0.0;
            smoothed_l1(P.safety_hor - sdf, P.mu, f, df);
            if (f > 0.0) {
                const d3 g = shape_grad<KIND>(P.shape, prel);
                n_grad++;
                acc.costp += f;
                const d3 wg = mat_mul(R, g);
                acc.gpx += df * wg.x; acc.gpy += df * wg.y; acc.gpz += df * wg.z;
                const d4 sq = quat_step_grad(quat, g, pm);
                acc.gqw += -df * sq.w; acc.gqx += -df * sq.x; acc.gqy += -df * sq.y; acc.gqz += -df * sq.z;
            }
        }
    }
}

template <int KIND>
__global__ __launch_bounds__(TS_THREADS) void tile_sweep_kernel(const SweepParams P) {
    __shared__ int s_box[TS_WAVES][8];
    __shared__ unsigned s_list[TS_CHUNK];
    __shared__ int s_cnt[TS_WAVES];
    __shared__ unsigned s_near[TS_WAVES][TS_NEAR_CAP];
    __shared__ double s_out[TS_WAVES][PARTIAL_STRIDE];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int K1 = P.K + 1;
    const long long s_begin = (long long)P.piece_begin * K1, s_end = (long long)P.piece_end * K1;
    const long long sidx = s_begin + (long long)blockIdx.x * TS_WAVES + wave;
    const bool valid = sidx < s_end;
    const DevGrid &G = P.grid;

    // ---------------- phase 0: pose of this wave's sample ----------------
    int j = 0, pi = 0, tb = 0;
    double step = 0.0;
    d3 pos = mk3(0, 0, 0), vel = pos, acc3 = pos, jer = pos, sna = pos, omg = pos;
    d4 quat; quat.w = 1; quat.x = quat.y = quat.z = 0;
    m3 R;
    Basis B;
    int lo[3] = {0, 0, 0}, hi[3] = {-1, -1, -1};
    bool active = false, culled = false;
    const double *cpiece = nullptr;
    int ld = 6 * P.N;
    if (valid) {
        const int gp = (int)(sidx / K1);
        j = (int)(sidx - (long long)gp * K1);
        tb = gp / P.N;
        pi = gp - tb * P.N;
        cpiece = P.coeffs + (size_t)tb * 18 * P.N + 6 * pi;
        const double Ti = P.T[(size_t)tb * P.N + pi];
        const double integral_frac = 1.0 / P.K;
        step = Ti * integral_frac;
        const double s1 = j * step;
        poly_basis(s1, B);
        pos = ct_beta(cpiece, ld, B.b0);
        vel = ct_beta(cpiece, ld, B.b1);
        acc3 = ct_beta(cpiece, ld, B.b2);
        jer = ct_beta(cpiece, ld, B.b3);
        sna = ct_beta(cpiece, ld, B.b4);
        FlatS fs; FlatS2 ft;
        flat_core(P.flat, vel, acc3, fs);
        flat_core2(P.flat, vel, acc3, jer, fs, ft);
        quat = flat_quat(fs);
        omg = flat_omg(fs, ft);
        R = quat_to_rot(quat);
        if (P.enable_pos) {
            active = true;
            if (P.cull_threshold > 0.0 && esdf_trilinear(G, pos) > P.cull_threshold) { active = false; culled = true; }
            if (active) {
                // getPointsInAABB(pos, bd/2): corners projected into the map, floor-indexed (PCSmap_manager.h:150-155)
                const double c1x = clampd(pos.x - P.bd_half, G.bmin[0], G.bmax[0]), c2x = clampd(pos.x + P.bd_half, G.bmin[0], G.bmax[0]);
                const double c1y = clampd(pos.y - P.bd_half, G.bmin[1], G.bmax[1]), c2y = clampd(pos.y + P.bd_half, G.bmin[1], G.bmax[1]);
                const double c1z = clampd(pos.z - P.bd_half, G.bmin[2], G.bmax[2]), c2z = clampd(pos.z + P.bd_half, G.bmin[2], G.bmax[2]);
                lo[0] = axis_index(c1x, G.bmin[0], G.res, G.X); hi[0] = axis_index(c2x, G.bmin[0], G.res, G.X);
                lo[1] = axis_index(c1y, G.bmin[1], G.res, G.Y); hi[1] = axis_index(c2y, G.bmin[1], G.res, G.Y);
                lo[2] = axis_index(c1z, G.bmin[2], G.res, G.Z); hi[2] = axis_index(c2z, G.bmin[2], G.res, G.Z);
                // NaN poses make every comparison false: keep the box sane
                if (!(hi[0] >= lo[0] && hi[1] >= lo[1] && hi[2] >= lo[2])) active = false;
            }
        }
    }
    if (lane == 0) {
        s_box[wave][0] = lo[0]; s_box[wave][1] = lo[1]; s_box[wave][2] = lo[2];
        s_box[wave][3] = hi[0]; s_box[wave][4] = hi[1]; s_box[wave][5] = hi[2];
        s_box[wave][6] = active ? 1 : 0;
    }
    __syncthreads();

    // ---------------- grouping: stage the union tile once, or each tile alone when they are far apart ----------------
    int ulo[3] = {1 << 30, 1 << 30, 1 << 30}, uhi[3] = {-1, -1, -1};
    long long sum_vol = 0;
    int n_active = 0;
#pragma unroll
    for (int w = 0; w < TS_WAVES; w++) {
        if (s_box[w][6]) {
            n_active++;
            long long v = 1;
#pragma unroll
            for (int a = 0; a < 3; a++) {
                ulo[a] = min(ulo[a], s_box[w][a]);
                uhi[a] = max(uhi[a], s_box[w][3 + a]);
                v *= (s_box[w][3 + a] - s_box[w][a] + 1);
            }
            sum_vol += v;
        }
    }
    Acc acc = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned n_pairs = 0, n_grad = 0;
    int near_cnt = 0;
    int near_ulo[3] = {0, 0, 0};   // origin the pending near-list entries are relative to

    if (n_active > 0) {
        const long long uvol = (long long)(uhi[0] - ulo[0] + 1) * (uhi[1] - ulo[1] + 1) * (uhi[2] - ulo[2] + 1);
        const bool single = (uvol <= sum_vol) && (uhi[0] - ulo[0] < 1024) && (uhi[1] - ulo[1] < 1024) && (uhi[2] - ulo[2] < 1024);
        const int n_groups = single ? 1 : TS_WAVES;
        for (int g = 0; g < n_groups; g++) {
            int glo[3], ghi[3];
            if (single) {
#pragma unroll
                for (int a = 0; a < 3; a++) { glo[a] = ulo[a]; ghi[a] = uhi[a]; }
            } else {
                if (!s_box[g][6]) continue;            // uniform across the workgroup
#pragma unroll
                for (int a = 0; a < 3; a++) { glo[a] = s_box[g][a]; ghi[a] = s_box[g][3 + a]; }
            }
            const bool member = active && (single || g == wave);
            const unsigned ux = ghi[0] - glo[0] + 1, uy = ghi[1] - glo[1] + 1, uz = ghi[2] - glo[2] + 1;
            const unsigned total = ux * uy * uz;
            const bool small = total < (1u << 24);
            const float inv_uz = 1.0f / (float)uz, inv_uy = 1.0f / (float)uy;
            // entries pending in the near list refer to the previous group's origin: drain them first
            if (near_cnt > 0 && (near_ulo[0] != glo[0] || near_ulo[1] != glo[1] || near_ulo[2] != glo[2])) {
                wave_lds_sync();
                near_process<KIND>(P, s_near[wave], near_cnt, lane, near_ulo, pos, R, quat, acc, n_grad);
                near_cnt = 0;
                wave_lds_sync();
            }
            near_ulo[0] = glo[0]; near_ulo[1] = glo[1]; near_ulo[2] = glo[2];

            for (unsigned cbase = 0; cbase < total; cbase += TS_CHUNK) {
                // ---------------- phase 1: stage TS_CHUNK union voxels -> compact LDS list ----------------
                {
                    const unsigned wbase = cbase + wave * TS_SEG;
                    bool q[TS_ROUNDS];
                    unsigned packed[TS_ROUNDS];
#pragma unroll
                    for (int r = 0; r < TS_ROUNDS; r++) {
                        const unsigned v = wbase + r * 64 + lane;
                        q[r] = false;
                        packed[r] = 0;
                        if (v < total) {
                            const unsigned t = fast_div(v, uz, inv_uz, small);
                            const unsigned lz = v - t * uz;
                            const unsigned lx = fast_div(t, uy, inv_uy, small);
                            const unsigned ly = t - lx * uy;
                            const size_t addr = ((size_t)(glo[0] + lx) * G.Y + (glo[1] + ly)) * G.Z + (glo[2] + lz);
                            if (P.variant == ISDF_V3_ESDF_TILE) q[r] = G.esdf[addr] <= P.occ_thresh;
                            else q[r] = G.occ[addr] != 0;
                            packed[r] = lx | (ly << 10) | (lz << 20);
                        }
                    }
                    int cnt = 0;
#pragma unroll
                    for (int r = 0; r < TS_ROUNDS; r++) {
                        const unsigned long long mask = __ballot(q[r]);
                        if (q[r]) {
                            const int pre = __popcll(mask & ((1ull << lane) - 1ull));
                            s_list[wave * TS_SEG + cnt + pre] = packed[r];
                        }
                        cnt += __popcll(mask);
                    }
                    if (lane == 0) s_cnt[wave] = cnt;
                }
                __syncthreads();
                // ---------------- phase 2: this wave's pose against the staged list ----------------
                if (member) {
#pragma unroll 1
                    for (int w2 = 0; w2 < TS_WAVES; w2++) {
                        const int cnt = s_cnt[w2];
                        for (int base = 0; base < cnt; base += 64) {
                            const int e = base + lane;
                            bool is_near = false;
                            unsigned ent = 0;
                            if (e < cnt) {
                                ent = s_list[w2 * TS_SEG + e];
                                const int gx = glo[0] + (int)(ent & 1023u), gy = glo[1] + (int)((ent >> 10) & 1023u), gz = glo[2] + (int)(ent >> 20);
                                const bool inside = gx >= lo[0] && gx <= hi[0] && gy >= lo[1] && gy <= hi[1] && gz >= lo[2] && gz <= hi[2];
                                if (inside) {
                                    const d3 c = mk3((gx + 0.5) * G.res + G.bmin[0], (gy + 0.5) * G.res + G.bmin[1], (gz + 0.5) * G.res + G.bmin[2]);
                                    const d3 prel = mat_tmul(R, c - pos);
                                    if (!(fabs(prel.x) > P.bd_half || fabs(prel.y) > P.bd_half || fabs(prel.z) > P.bd_half)) {
                                        const double sdf = shape_sdf<KIND>(P.shape, prel);
                                        n_pairs++;
                                        is_near = (P.safety_hor - sdf) > 0.0;
                                    }
                                }
                            }
                            const unsigned long long nm = __ballot(is_near);
                            if (nm) {
                                const int nn = __popcll(nm);
                                if (near_cnt + nn > TS_NEAR_CAP) {
                                    wave_lds_sync();
                                    near_process<KIND>(P, s_near[wave], near_cnt, lane, near_ulo, pos, R, quat, acc, n_grad);
                                    near_cnt = 0;
                                    wave_lds_sync();
                                }
                                if (is_near) s_near[wave][near_cnt + __popcll(nm & ((1ull << lane) - 1ull))] = ent;
                                near_cnt += nn;
                            }
                        }
                    }
                }
                __syncthreads();   // the list is overwritten by the next chunk
            }
        }
    }
    if (near_cnt > 0) {
        wave_lds_sync();
        near_process<KIND>(P, s_near[wave], near_cnt, lane, near_ulo, pos, R, quat, acc, n_grad);
    }

    // ---------------- phase 3: wave reduction, dynamics penalties, flatness reverse mode, partial record ----------------
    if (!valid) return;
    acc.costp = wave_sum(acc.costp);
    acc.gpx = wave_sum(acc.gpx); acc.gpy = wave_sum(acc.gpy); acc.gpz = wave_sum(acc.gpz);
    acc.gqw = wave_sum(acc.gqw); acc.gqx = wave_sum(acc.gqx); acc.gqy = wave_sum(acc.gqy); acc.gqz = wave_sum(acc.gqz);
    {
        unsigned long long np = n_pairs, ng = n_grad;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { np += __shfl_xor(np, off, 64); ng += __shfl_xor(ng, off, 64); }
        if (lane == 0) {
            atomicAdd(&P.stats[0], 1ull);
            if (culled) atomicAdd(&P.stats[1], 1ull);
            if (np) atomicAdd(&P.stats[2], np);
            if (ng) atomicAdd(&P.stats[3], ng);
        }
    }
    double pena = 0.0;
    d3 gradVel = mk3(0, 0, 0), gradPos = gradVel, gradOmg = gradVel;
    d4 gradQuat; gradQuat.w = gradQuat.x = gradQuat.y = gradQuat.z = 0.0;
    if (P.enable_dyn) {
        const double cos_theta = 1.0 - 2.0 * (quat.x * quat.x + quat.y * quat.y);
        const double violaVel = dot3(vel, vel) - P.vel_sqr_max;
        const double violaOmg = dot3(omg, omg) - P.omg_sqr_max;
        const double violaTheta = acos(cos_theta) - P.theta_max;
        double pe, pd;
        if (smoothed_l1(violaVel, P.mu, pe, pd)) {
            const double f = P.weight_v * pd * 2.0;
            gradVel = f * vel;
            pena += P.weight_v * pe;
        }
        if (smoothed_l1(violaOmg, P.mu, pe, pd)) {
            const double f = P.weight_omg * pd * 2.0;
            gradOmg = f * omg;
            pena += P.weight_omg * pe;
        }
        if (smoothed_l1(violaTheta, P.mu, pe, pd)) {
            const double f = P.weight_theta * pd / sqrt(1.0 - cos_theta * cos_theta) * 4.0;
            gradQuat.x += f * quat.x;
            gradQuat.y += f * quat.y;
            pena += P.weight_theta * pe;
        }
    }
    if (P.enable_pos && acc.costp > 0.0) {
        gradPos.x += P.weight_p * acc.gpx; gradPos.y += P.weight_p * acc.gpy; gradPos.z += P.weight_p * acc.gpz;
        gradQuat.w += P.weight_p * acc.gqw; gradQuat.x += P.weight_p * acc.gqx;
        gradQuat.y += P.weight_p * acc.gqy; gradQuat.z += P.weight_p * acc.gqz;
        pena += P.weight_p * acc.costp;
    }
    d3 gP, gV, gA, gJ;
    flat_backward(P.flat, vel, acc3, jer, gradPos, gradVel, gradQuat, gradOmg, gP, gV, gA, gJ);
    const double node = (j == 0 || j == P.K) ? 0.5 : 1.0;
    const double integral_frac = 1.0 / P.K;
    const double alpha = j * integral_frac;
    if (lane == 0) {
        double *o = s_out[wave];
        const double gPv[3] = {gP.x, gP.y, gP.z}, gVv[3] = {gV.x, gV.y, gV.z}, gAv[3] = {gA.x, gA.y, gA.z}, gJv[3] = {gJ.x, gJ.y, gJ.z};
#pragma unroll
        for (int d = 0; d < 3; d++)
#pragma unroll
            for (int r = 0; r < 6; r++)
                o[d * 6 + r] = (B.b0[r] * gPv[d] + B.b1[r] * gVv[d] + B.b2[r] * gAv[d] + B.b3[r] * gJv[d]) * node * step;
        o[18] = (dot3(gP, vel) + dot3(gV, acc3) + dot3(gA, jer) + dot3(gJ, sna)) * alpha * node * step + node * integral_frac * pena;
        o[19] = node * step * pena;
    }
    wave_lds_sync();
    if (lane < PARTIAL_STRIDE) P.partial[(size_t)sidx * PARTIAL_STRIDE + lane] = s_out[wave][lane];
}

// Deterministic per-piece reduction of the partial records (fixed order) into the packed output
//   out[b] = [cost | gradT[N] | gradC(6N x 3 col-major)].
// Blocks [0, n_pieces): one 64-lane block per piece of this rank's range; blocks [n_pieces, n_pieces + n_traj):
// per-trajectory cost.  `out` must have been zeroed (pieces outside the shard stay zero).
__global__ __launch_bounds__(64) void tile_reduce_kernel(const ReduceParams P) {
    const int n_pieces = P.piece_end - P.piece_begin;
    const int K1 = P.K + 1;
    const int lane = threadIdx.x;
    const size_t ostride = (size_t)1 + 19 * (size_t)P.N;
    if ((int)blockIdx.x < n_pieces) {
        const int gp = P.piece_begin + blockIdx.x;
        const int b = gp / P.N, i = gp - b * P.N;
        if (lane < 19) {
            const double *src = P.partial + (size_t)gp * K1 * PARTIAL_STRIDE + lane;
            double s = 0.0;
            for (int jj = 0; jj < K1; jj++) s += src[(size_t)jj * PARTIAL_STRIDE];
            double *o = P.out + (size_t)b * ostride;
            if (lane == 18) o[1 + i] = s;
            else {
                const int d = lane / 6, r = lane - 6 * d;
                o[1 + P.N + (size_t)d * 6 * P.N + 6 * i + r] = s;
            }
        }
    } else {
        const int b = blockIdx.x - n_pieces;
        // pieces of trajectory b inside this rank's range
        const int p0 = max(P.piece_begin, b * P.N), p1 = min(P.piece_end, (b + 1) * P.N);
        double s = 0.0;
        if (p1 > p0) {
            const long long n = (long long)(p1 - p0) * K1;
            const double *src = P.partial + (size_t)p0 * K1 * PARTIAL_STRIDE + 19;
            for (long long e = lane; e < n; e += 64) s += src[(size_t)e * PARTIAL_STRIDE];
        }
        s = wave_sum(s);
        if (lane == 0) P.out[(size_t)b * ostride] = s;
    }
}

template <int KIND>
static void launch_kind(const SweepParams &P, int blocks, hipStream_t stream) {
    hipLaunchKernelGGL(tile_sweep_kernel<KIND>, dim3(blocks), dim3(TS_THREADS), 0, stream, P);
}

void launch_tile_sweep(const SweepParams &P, hipStream_t stream) {
    const long long n_samples = (long long)(P.piece_end - P.piece_begin) * (P.K + 1);
    if (n_samples <= 0) return;
    const int blocks = (int)((n_samples + TS_WAVES - 1) / TS_WAVES);
    switch (P.shape.kind) {
    case ISDF_SHAPE_BOX: launch_kind<ISDF_SHAPE_BOX>(P, blocks, stream); break;
    case ISDF_SHAPE_ROUNDEDCONE: launch_kind<ISDF_SHAPE_ROUNDEDCONE>(P, blocks, stream); break;
    default: launch_kind<-1>(P, blocks, stream); break;
    }
}

void launch_tile_reduce(const ReduceParams &P, hipStream_t stream) {
    const int n_pieces = P.piece_end - P.piece_begin;
    if (n_pieces <= 0) return;
    hipLaunchKernelGGL(tile_reduce_kernel, dim3(n_pieces + P.n_traj), dim3(64), 0, stream, P);
}

} // namespace isdf
