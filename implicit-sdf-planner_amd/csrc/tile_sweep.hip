// V2 / V3 integral sweep on gfx950: the P x (K+1) constraint-sample loop of
//   addTimeIntPenaltyParallel   (back_end_optimizer.hpp:432-554)
// with the per-pose tile term grad_cost_p (:766-824) plugged in before the flatness back-prop (:536), the voxel
// qualifier being occupancy (V2) or esdf <= occ_thresh (V3).
//
// One optimizer step = four launches (DESIGN.md "integral sweep"):
//   pose_kernel     one THREAD per constraint sample: quintic evaluation, flatness forward, rotation, the tile index
//                   box of getPointsInAABB, optional whole-tile cull (trilinear ESDF at the pose) -> PoseRec.
//                   (The serial fp64 chain runs once per sample, not once per lane of a wavefront.)
//   tile_kernel     one WAVEFRONT per sample, 4 consecutive samples per 256-thread workgroup:
//                     phase 1 (staging): the workgroup walks the UNION of its samples' tiles once with coalesced
//                       z-row reads of the grid and stages the tile into LDS as a compacted list of qualifying voxels
//                       (ballot/popcount compaction, per-wave segments => deterministic order, no atomics);
//                       consecutive samples' tiles overlap >90 %, so the tile is fetched once per workgroup.
//                     phase 2: each wave streams the LDS list against its own pose: body-frame transform,
//                       |p_rel| cull, robot SDF; voxels with an active penalty are compacted again (per-wave LDS
//                       list) and get the 6 extra SDF evaluations of the central-difference gradient in dense rounds.
//                     phase 3: wave butterfly reduction of (cost, dL/dpos, dL/dquat) -> acc[sample][8].
//   tail_kernel     one block per piece, one thread per sample: dynamics penalties, flatness reverse mode,
//                   beta (x) grad scatter, then a fixed-order in-block sum over the piece's K+1 samples written
//                   straight into gradC rows 6i..6i+5 and gradT(i).
//   finalize_kernel one block per trajectory: cost = sum of piece costs (fixed order); zero-fills the entries of
//                   pieces outside this rank's shard so that the ranks' outputs sum to the full result.
// Everything on the data path is fp64 except the stored ESDF (fp32).  No MFMA: gather/interpolate work.
#include "isdf_internal.hpp"
#include "dev_mesh.hpp"

namespace isdf {

constexpr int TS_WAVES = 1;                               // waves are independent (one constraint sample each)
constexpr int TS_THREADS = 64 * TS_WAVES;
constexpr int TS_LIST_CAP = 64 * 32;                      // ring capacity >= voxels of one round (64 z-rows x 32)
constexpr int TS_BITS_CAP = 512;                          // z-rows whose qualifier bits are staged in LDS at a time
constexpr int TS_NEAR_CAP = 128;                          // per-wave candidate ring (power of two; a slot of 64 leaves when full)
constexpr float TS_FILTER_EPS = 2e-3f;                    // slack (m) of the fp32 pre-filter, >> its rounding error

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

// ---------------------------------------------------------------------------------------------------------------
// pose_kernel: thread per sample
// ---------------------------------------------------------------------------------------------------------------
struct SampleId { int b, i, j; long long s; };
__device__ __forceinline__ SampleId decode_sample(const SweepParams &P, long long s) {
    const int K1 = P.K + 1;
    SampleId id;
    id.s = s;
    const int gp = (int)(s / K1);
    id.j = (int)(s - (long long)gp * K1);
    id.b = gp / P.N;
    id.i = gp - id.b * P.N;
    return id;
}

__device__ __forceinline__ void compute_pose(const SweepParams &P, long long s, PoseRec &rec) {
    const SampleId id = decode_sample(P, s);
    const DevGrid &G = P.grid;
    const double *c = P.coeffs + (size_t)id.b * 18 * P.N + 6 * id.i;
    const int ld = 6 * P.N;
    const double step = P.T[(size_t)id.b * P.N + id.i] * (1.0 / P.K);
    Basis B;
    poly_basis(id.j * step, B);
    const d3 pos = ct_beta(c, ld, B.b0), vel = ct_beta(c, ld, B.b1), acc = ct_beta(c, ld, B.b2);
    FlatS fs;
    flat_core(P.flat, vel, acc, fs);
    const d4 q = flat_quat(fs);
    const m3 R = quat_to_rot(q);
    rec.pos[0] = pos.x; rec.pos[1] = pos.y; rec.pos[2] = pos.z;
#pragma unroll
    for (int k = 0; k < 9; k++) rec.R[k] = R.a[k];
    rec.quat[0] = q.w; rec.quat[1] = q.x; rec.quat[2] = q.y; rec.quat[3] = q.z;
    rec.lo[0] = rec.lo[1] = rec.lo[2] = 0;
    rec.hi[0] = rec.hi[1] = rec.hi[2] = -1;
    rec.flags = 0;
    rec.pad = 0;
    if (P.enable_pos) {
        bool active = true;
        if (P.cull_threshold > 0.0 && esdf_trilinear(G, pos) > P.cull_threshold) { active = false; rec.flags |= 2; }
        if (active) {
            // getPointsInAABB(pos, bd/2): corners projected into the map, floor-indexed (PCSmap_manager.h:150-155)
            const double p[3] = {pos.x, pos.y, pos.z};
            const int dim[3] = {G.X, G.Y, G.Z};
#pragma unroll
            for (int a = 0; a < 3; a++) {
                rec.lo[a] = axis_index(clampd(p[a] - P.bd_half, G.bmin[a], G.bmax[a]), G.bmin[a], G.res, dim[a]);
                rec.hi[a] = axis_index(clampd(p[a] + P.bd_half, G.bmin[a], G.bmax[a]), G.bmin[a], G.res, dim[a]);
            }
            // NaN poses make every comparison false: keep the box sane
            if (rec.hi[0] >= rec.lo[0] && rec.hi[1] >= rec.lo[1] && rec.hi[2] >= rec.lo[2]) rec.flags |= 1;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// scan_kernel: wave per sample, 4 samples per workgroup share one staged union tile; fp32 + integer work only
// ---------------------------------------------------------------------------------------------------------------
// The grid is read through its QUALIFIER BIT-GRID (1 bit per voxel: esdf <= occ_thresh for V3, occupied for V2;
// built once per set_grid by build_bits_kernel), 32 voxels of a z-row per dword.
// Candidate slots: a wave hands every 64 surviving voxels of its sample to the exact pass as one slot of the global
// pool.  Slot ids come from one atomicAdd on one of POOL_SHARDS counters (a single counter saturates at ~90
// allocations/us); slots of a sample are chained through `prev`, so the order in which the exact results are summed
// is fixed even though slot ids are not.
// wave-wide exclusive prefix sum with DPP row shifts / row broadcasts (no LDS traffic)
__device__ __forceinline__ int wave_excl_scan(int v, int lane, int &total) {
    int x = v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);   // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);   // row_shr:8  -> inclusive scan inside each row of 16
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    total = __builtin_amdgcn_readlane(x, 63);
    return x - v;
}

template <int KIND>
__global__ __launch_bounds__(TS_THREADS) void scan_kernel(const SweepParams P) {
    __shared__ unsigned s_bits[TS_WAVES][TS_BITS_CAP];   // qualifier bits of up to TS_BITS_CAP z-rows of the tile
    __shared__ unsigned s_list[TS_WAVES][TS_LIST_CAP];   // ring of compacted voxels awaiting the fp32 test
    __shared__ unsigned s_near[TS_WAVES][TS_NEAR_CAP];   // ring of candidates awaiting a pool slot

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int K1 = P.K + 1;
    const long long s_begin = (long long)P.piece_begin * K1, s_end = (long long)P.piece_end * K1;
    const long long sidx = s_begin + (long long)blockIdx.x * TS_WAVES + wave;
    if (sidx >= s_end) return;                                 // waves are independent: no workgroup barriers below
    const DevGrid &G = P.grid;
    // pose of this wave's sample (wave-uniform fp64 chain), published for exact_kernel / stats
    PoseRec rec;
    compute_pose(P, sidx, rec);
    if (lane == 0) P.pose[sidx] = rec;
    int prev_slot = -1;
    if (rec.flags & 1) {
        const int lox = rec.lo[0], loy = rec.lo[1], loz = rec.lo[2];
        const int nx = rec.hi[0] - lox + 1, ny = rec.hi[1] - loy + 1, nzt = rec.hi[2] - loz + 1;
        const int n_rows = nx * ny;
        const float inv_ny = 1.0f / (float)ny;
        const bool small = n_rows < (1 << 24);

        // fp32 pre-filter constants (wave-uniform)
        const bool use_filter = P.shape.filter_f32 != 0;
        float Rf[9];
#pragma unroll
        for (int k = 0; k < 9; k++) Rf[k] = (float)rec.R[k];
        const float resf = (float)G.res;
        const float bdh_f = (float)P.bd_half + TS_FILTER_EPS;
        const float thr_f = (float)P.safety_hor + TS_FILTER_EPS;
        // pose in the tile's voxel frame, pre-multiplied: m = l*res - pvr,  pvr = ((pos - bmin)/res - 0.5 - lo)*res
        const float pvx = (float)(((rec.pos[0] - G.bmin[0]) / G.res - 0.5 - lox) * G.res);
        const float pvy = (float)(((rec.pos[1] - G.bmin[1]) / G.res - 0.5 - loy) * G.res);
        const float pvz = (float)(((rec.pos[2] - G.bmin[2]) / G.res - 0.5 - loz) * G.res);

        int near_head = 0, near_cnt = 0;
        const int shard = (int)(blockIdx.x % POOL_SHARDS);
        const int shard_cap = P.pool_cap / POOL_SHARDS;
        unsigned *const pool_count = P.pool_count + (size_t)P.parity * POOL_SHARDS * POOL_COUNTER_STRIDE;
        // hand the oldest `n` (<= 64) candidates to the exact pass
        auto flush = [&](int n) {
            wave_lds_sync();
            int local = 0;
            if (lane == 0) local = (int)atomicAdd(pool_count + shard * POOL_COUNTER_STRIDE, 1u);
            local = __builtin_amdgcn_readfirstlane(local);
            if (local < shard_cap) {
                const int slot = shard * shard_cap + local;
                P.pool_entries[(size_t)slot * 64 + lane] = (lane < n) ? s_near[wave][(near_head + lane) & (TS_NEAR_CAP - 1)] : 0xFFFFFFFFu;
                if (lane == 0) {
                    P.slot_sample[slot] = (int)(sidx - s_begin);
                    P.slot_prev[slot] = prev_slot;
                }
                prev_slot = slot;
            } else if (lane == 0) {
                atomicOr((unsigned int *)&P.stats[4], 1u);   // pool overflow: result invalid, host reports ISDF_ERR_OVERFLOW
            }
            near_head = (near_head + n) & (TS_NEAR_CAP - 1);
            near_cnt -= n;
            wave_lds_sync();
        };

        // (lx, ly) of tile row r, packed lx | ly << 10
        auto row_xy = [&](int r) -> unsigned {
            unsigned lx, ly;
            if (small) {
                lx = (unsigned)((float)r * inv_ny);
                int rem = r - (int)lx * ny;
                if (rem < 0) { lx--; rem += ny; } else if (rem >= ny) { lx++; rem -= ny; }
                ly = (unsigned)rem;
            } else { lx = (unsigned)(r / ny); ly = (unsigned)(r - (int)lx * ny); }
            return lx | (ly << 10);
        };
        // the <= 32 qualifier bits of z-block zb of tile row r: 1-2 dwords of the bit-grid
        auto row_bits = [&](int r, int zb) -> unsigned {
            const bool rok = r < n_rows;
            const unsigned pk = rok ? row_xy(r) : 0u;
            const int z0 = loz + (zb << 5);
            const int nz = min(32, nzt - (zb << 5));
            const size_t rowbase = ((size_t)(lox + (int)(pk & 1023u)) * G.Y + (loy + (int)(pk >> 10))) * G.ZW;
            const int w0 = z0 >> 5, sh = z0 & 31;
            const unsigned b0 = G.bits[rowbase + w0];
            const unsigned b1 = G.bits[rowbase + min(w0 + 1, G.ZW - 1)];
            unsigned bits = sh ? ((b0 >> sh) | (b1 << (32 - sh))) : b0;     // when w0+1 == ZW the b1 bits are masked off below
            if (nz < 32) bits &= (1u << nz) - 1u;
            return rok ? bits : 0u;
        };

        // Row pruning: in the body frame the voxels of one z-row lie on the line o + lz*dz.  Only the part of the line
        // inside (shape bounding box inflated by safety_hor) ∩ (the |p_rel| <= bd/2 cube) can carry a penalty
        // (isdf_shape.bbox_* guarantee), so the rest of the row never enters the list.  Conservative by construction.
        const bool prune = P.shape.prune_rows != 0;
        float blo[3], bhi[3], dzv[3], dzi[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            blo[a] = fmaxf(P.shape.bbox_lo[a] - thr_f, -bdh_f);
            bhi[a] = fminf(P.shape.bbox_hi[a] + thr_f, bdh_f);
            dzv[a] = Rf[6 + a] * resf;
            dzi[a] = 1.0f / dzv[a];
        }
        auto row_mask = [&](unsigned pk, int zb) -> unsigned {
            const float mx = fmaf((float)(pk & 1023u), resf, -pvx), my = fmaf((float)(pk >> 10), resf, -pvy);
            float tmin = -1.0e30f, tmax = 1.0e30f;
            bool empty = false;
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const float o = Rf[a] * mx + Rf[3 + a] * my - Rf[6 + a] * pvz;
                if (fabsf(dzv[a]) > 1.0e-12f) {
                    const float t1 = (blo[a] - o) * dzi[a], t2 = (bhi[a] - o) * dzi[a];
                    tmin = fmaxf(tmin, fminf(t1, t2));      // fmaxf/fminf drop NaNs: never tightens on garbage
                    tmax = fminf(tmax, fmaxf(t1, t2));
                } else {
                    empty = empty | (o < blo[a] - 1.0e-3f) | (o > bhi[a] + 1.0e-3f);
                }
            }
            const float off = (float)(zb << 5);
            const float flo = fminf(fmaxf(ceilf(tmin - 0.02f) - off, 0.0f), 32.0f);
            const float fhi = fminf(fmaxf(floorf(tmax + 0.02f) - off, -1.0f), 31.0f);
            const int zlo = (int)flo, zhi = (int)fhi;
            if (empty || zlo > zhi) return 0u;
            const int n = zhi - zlo + 1;
            return (n >= 32) ? 0xFFFFFFFFu : (((1u << n) - 1u) << zlo);
        };

        int pend_head = 0, pend = 0;                 // ring of compacted voxels awaiting the fp32 test
        // conservative fp32 test of the oldest min(pend, 64) listed voxels (branch-free body)
        auto scan64 = [&]() {
            const int n = min(pend, 64);
            const unsigned ent = s_list[wave][(pend_head + min(lane, n - 1)) & (TS_LIST_CAP - 1)];
            bool cand = lane < n;
            if (use_filter) {
                const float flx = (float)(ent & 1023u), fly = (float)((ent >> 10) & 1023u), flz = (float)(ent >> 20);
                const float mx = fmaf(flx, resf, -pvx), my = fmaf(fly, resf, -pvy), mz = fmaf(flz, resf, -pvz);
                const f3 pr = mkv<float>(Rf[0] * mx + Rf[3] * my + Rf[6] * mz, Rf[1] * mx + Rf[4] * my + Rf[7] * mz,
                                         Rf[2] * mx + Rf[5] * my + Rf[8] * mz);
                const bool incube = fmaxf(fabsf(pr.x), fmaxf(fabsf(pr.y), fabsf(pr.z))) <= bdh_f;
                const float sdf = analytic_sdf<KIND, float>(P.shape.f, P.shape.kind, pr);
                cand = cand & incube & !(sdf >= thr_f);   // reject only what provably has zero penalty; NaN => candidate
            }
            pend_head = (pend_head + n) & (TS_LIST_CAP - 1);
            pend -= n;
            const unsigned long long nm = __ballot(cand);
            if (nm) {
                if (cand) s_near[wave][(near_head + near_cnt + __popcll(nm & ((1ull << lane) - 1ull))) & (TS_NEAR_CAP - 1)] = ent;
                near_cnt += __popcll(nm);
                if (near_cnt >= 64) flush(64);
            }
        };

        const int n_zb = (nzt + 31) >> 5;            // 32-voxel blocks along z (1 for kernel_size <= 31)
        for (int zb = 0; zb < n_zb; zb++)
        for (int rblk = 0; rblk < n_rows; rblk += TS_BITS_CAP) {
            // ---- stage the qualifier bits of up to TS_BITS_CAP rows into LDS: all loads in flight together
            const int nrb = min(TS_BITS_CAP, n_rows - rblk);
#pragma unroll 8
            for (int r = lane; r < nrb; r += 64) s_bits[wave][r] = row_bits(rblk + r, zb);
            wave_lds_sync();
            // ---- one lane per row, 64 rows per round: prune, compact into the ring, test full wavefronts
            for (int r0 = 0; r0 < nrb; r0 += 64) {
                const int r = r0 + lane;
                unsigned bits_cur = (r < nrb) ? s_bits[wave][r] : 0u;
                unsigned pk = 0;
                if (bits_cur) {
                    pk = row_xy(rblk + r);
                    if (prune) bits_cur &= row_mask(pk, zb);
                }
                if (!__ballot(bits_cur != 0u)) continue;
                int total;
                int pos = wave_excl_scan(__popc(bits_cur), lane, total);
                if (pend + total > TS_LIST_CAP) {           // make room: drain the (partial) backlog first
                    while (pend > 0) scan64();
                }
                unsigned hb = bits_cur;
                const unsigned zoff = (unsigned)(zb << 5);
                pos += pend_head + pend;
                while (hb) {
                    const unsigned z = (unsigned)__ffs(hb) - 1u;
                    hb &= hb - 1u;
                    s_list[wave][(pos++) & (TS_LIST_CAP - 1)] = pk | ((zoff + z) << 20);
                }
                pend += total;
                wave_lds_sync();
                while (pend >= 64) scan64();
            }
            wave_lds_sync();    // s_bits is rewritten by the next block
        }
        while (pend > 0) scan64();
        if (near_cnt > 0) flush(near_cnt);
    }
    if (lane == 0) P.sample_last[sidx - s_begin] = prev_slot;
}

// ---------------------------------------------------------------------------------------------------------------
// exact_kernel: one wave per candidate slot (64 voxels of one pose), fp64
// ---------------------------------------------------------------------------------------------------------------
// Voxel centre, body-frame point, the |p_rel| > bd/2 cull of grad_cost_p (:800-803), robot SDF, smoothedL1 and — when
// the penalty is active — the central-difference gradient and the pose-gradient terms (:805-820).  Slots are spread
// over the whole chip, so a pose buried in obstacles does not serialise behind one wavefront.
constexpr int EX_WAVES = 4;
constexpr int EX_BLOCKS_PER_SHARD = 8;

template <int KIND>
__global__ __launch_bounds__(64 * EX_WAVES) void exact_kernel(const SweepParams P) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int K1 = P.K + 1;
    const long long s_begin = (long long)P.piece_begin * K1;
    const int shard = blockIdx.x;
    const int shard_cap = P.pool_cap / POOL_SHARDS;
    const int n_local = (int)min(P.pool_count[((size_t)P.parity * POOL_SHARDS + shard) * POOL_COUNTER_STRIDE], (unsigned)shard_cap);
    for (int local = blockIdx.y * EX_WAVES + wave; local < n_local; local += gridDim.y * EX_WAVES) {
        const int slot = shard * shard_cap + local;
        const PoseRec *rec = P.pose + (s_begin + P.slot_sample[slot]);   // wave-uniform
        const unsigned ent = P.pool_entries[(size_t)slot * 64 + lane];
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
        bool is_pair = false, is_grad = false;
        if (ent != 0xFFFFFFFFu) {
            const d3 pos = mk3(rec->pos[0], rec->pos[1], rec->pos[2]);
            m3 R;
#pragma unroll
            for (int k = 0; k < 9; k++) R.a[k] = rec->R[k];
            const int gx = rec->lo[0] + (int)(ent & 1023u), gy = rec->lo[1] + (int)((ent >> 10) & 1023u), gz = rec->lo[2] + (int)(ent >> 20);
            const d3 c = mk3((gx + 0.5) * P.grid.res + P.grid.bmin[0], (gy + 0.5) * P.grid.res + P.grid.bmin[1],
                             (gz + 0.5) * P.grid.res + P.grid.bmin[2]);
            const d3 pm = c - pos;
            const d3 prel = mat_tmul(R, pm);
            if (!(fabs(prel.x) > P.bd_half || fabs(prel.y) > P.bd_half || fabs(prel.z) > P.bd_half)) {
                const double sdf = shape_sdf<KIND>(P.shape, prel);
                is_pair = true;
                double f = 0.0, df = 0.0;
                smoothed_l1(P.safety_hor - sdf, P.mu, f, df);
                if (f > 0.0) {
                    const d3 g = shape_grad<KIND>(P.shape, prel);
                    is_grad = true;
                    a0 = f;
                    const d3 wg = mat_mul(R, g);
                    a1 = df * wg.x; a2 = df * wg.y; a3 = df * wg.z;
                    d4 quat; quat.w = rec->quat[0]; quat.x = rec->quat[1]; quat.y = rec->quat[2]; quat.z = rec->quat[3];
                    const d4 sq = quat_step_grad(quat, g, pm);
                    a4 = -df * sq.w; a5 = -df * sq.x; a6 = -df * sq.y; a7 = -df * sq.z;
                }
            }
        }
        a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2); a3 = wave_sum(a3);
        a4 = wave_sum(a4); a5 = wave_sum(a5); a6 = wave_sum(a6); a7 = wave_sum(a7);
        const unsigned long long pm_ = __ballot(is_pair), gm_ = __ballot(is_grad);
        if (lane == 0) {
            double *o = P.slot_res + (size_t)slot * ACC_STRIDE;
            o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3; o[4] = a4; o[5] = a5; o[6] = a6; o[7] = a7;
            P.slot_cnt[slot] = (unsigned)__popcll(pm_) | ((unsigned)__popcll(gm_) << 16);   // summed on demand (stats_kernel)
        }
    }
}

// on-demand statistics (isdf_get_stats / host API): culled poses, exact pairs, active pairs.  Not on the timed path.
__global__ __launch_bounds__(256) void stats_kernel(const SweepParams P) {
    const int K1 = P.K + 1;
    const long long n_samples = (long long)(P.piece_end - P.piece_begin) * K1;
    const long long s_begin = (long long)P.piece_begin * K1;
    unsigned long long culled = 0, pairs = 0, grads = 0;
    if (P.enable_pos)
        for (long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x; s < n_samples; s += (long long)gridDim.x * blockDim.x)
            if (P.pose[s_begin + s].flags & 2) culled++;
    if (P.enable_pos) {
        const int shard_cap = P.pool_cap / POOL_SHARDS;
        for (int shard = blockIdx.x; shard < POOL_SHARDS; shard += gridDim.x) {
            const int n_local = (int)min(P.pool_count[((size_t)P.parity * POOL_SHARDS + shard) * POOL_COUNTER_STRIDE], (unsigned)shard_cap);
            for (int l = threadIdx.x; l < n_local; l += blockDim.x) {
                const unsigned c = P.slot_cnt[shard * shard_cap + l];
                pairs += c & 0xFFFFu; grads += c >> 16;
            }
        }
    }
    if (culled) atomicAdd(&P.stats[1], culled);
    if (pairs) atomicAdd(&P.stats[2], pairs);
    if (grads) atomicAdd(&P.stats[3], grads);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&P.stats[0], (unsigned long long)n_samples);
}

// ---------------------------------------------------------------------------------------------------------------
// tail_kernel: block per piece, thread per sample
// ---------------------------------------------------------------------------------------------------------------
constexpr int TL_THREADS = 128;

__global__ __launch_bounds__(TL_THREADS) void tail_kernel(const SweepParams P) {
    __shared__ double s_val[TL_THREADS][PARTIAL_STRIDE + 1];   // +1: column reads by 20 threads hit distinct banks
    const int K1 = P.K + 1;
    const int gp = P.piece_begin + blockIdx.x;
    const int b = gp / P.N, i = gp - b * P.N;
    const double *c = P.coeffs + (size_t)b * 18 * P.N + 6 * i;
    const int ld = 6 * P.N;
    const double integral_frac = 1.0 / P.K;
    const double step = P.T[(size_t)b * P.N + i] * integral_frac;
    double colsum = 0.0;   // threads 0..19: running sum of their output column, samples in ascending order
    for (int j0 = 0; j0 < K1; j0 += TL_THREADS) {
        const int j = j0 + threadIdx.x;
        if (j < K1) {
            const long long s = (long long)gp * K1 + j;
            Basis B;
            poly_basis(j * step, B);
            const d3 vel = ct_beta(c, ld, B.b1), acc3 = ct_beta(c, ld, B.b2), jer = ct_beta(c, ld, B.b3), sna = ct_beta(c, ld, B.b4);
            double pena = 0.0;
            d3 gradVel = mk3(0, 0, 0), gradPos = gradVel, gradOmg = gradVel;
            d4 gradQuat; gradQuat.w = gradQuat.x = gradQuat.y = gradQuat.z = 0.0;
            if (P.enable_dyn) {
                FlatS fs; FlatS2 ft;
                flat_core(P.flat, vel, acc3, fs);
                flat_core2(P.flat, vel, acc3, jer, fs, ft);
                const d4 quat = flat_quat(fs);
                const d3 omg = flat_omg(fs, ft);
                const double cos_theta = 1.0 - 2.0 * (quat.x * quat.x + quat.y * quat.y);
                const double violaVel = dot3(vel, vel) - P.vel_sqr_max;
                const double violaOmg = dot3(omg, omg) - P.omg_sqr_max;
                const double violaTheta = acos(cos_theta) - P.theta_max;
                double pe, pd;
                if (smoothed_l1(violaVel, P.mu, pe, pd)) {
                    gradVel = (P.weight_v * pd * 2.0) * vel;
                    pena += P.weight_v * pe;
                }
                if (smoothed_l1(violaOmg, P.mu, pe, pd)) {
                    gradOmg = (P.weight_omg * pd * 2.0) * omg;
                    pena += P.weight_omg * pe;
                }
                if (smoothed_l1(violaTheta, P.mu, pe, pd)) {
                    const double f = P.weight_theta * pd / sqrt(1.0 - cos_theta * cos_theta) * 4.0;
                    gradQuat.x += f * quat.x;
                    gradQuat.y += f * quat.y;
                    pena += P.weight_theta * pe;
                }
            }
            if (P.enable_pos) {
                // this pose's exact results: walk its slot chain (fixed order: last slot first)
                double a[ACC_STRIDE] = {0, 0, 0, 0, 0, 0, 0, 0};
                for (int sl = P.sample_last[s - (long long)P.piece_begin * K1]; sl >= 0; sl = P.slot_prev[sl]) {
                    const double *r = P.slot_res + (size_t)sl * ACC_STRIDE;
#pragma unroll
                    for (int k = 0; k < ACC_STRIDE; k++) a[k] += r[k];
                }
                if (a[0] > 0.0) {
                    gradPos.x += P.weight_p * a[1]; gradPos.y += P.weight_p * a[2]; gradPos.z += P.weight_p * a[3];
                    gradQuat.w += P.weight_p * a[4]; gradQuat.x += P.weight_p * a[5];
                    gradQuat.y += P.weight_p * a[6]; gradQuat.z += P.weight_p * a[7];
                    pena += P.weight_p * a[0];
                }
            }
            d3 gP, gV, gA, gJ;
            flat_backward(P.flat, vel, acc3, jer, gradPos, gradVel, gradQuat, gradOmg, gP, gV, gA, gJ);
            const double node = (j == 0 || j == P.K) ? 0.5 : 1.0;
            const double alpha = j * integral_frac;
            double *o = s_val[threadIdx.x];
            const double gPv[3] = {gP.x, gP.y, gP.z}, gVv[3] = {gV.x, gV.y, gV.z}, gAv[3] = {gA.x, gA.y, gA.z}, gJv[3] = {gJ.x, gJ.y, gJ.z};
#pragma unroll
            for (int d = 0; d < 3; d++)
#pragma unroll
                for (int r = 0; r < 6; r++)
                    o[d * 6 + r] = (B.b0[r] * gPv[d] + B.b1[r] * gVv[d] + B.b2[r] * gAv[d] + B.b3[r] * gJv[d]) * node * step;
            o[18] = (dot3(gP, vel) + dot3(gV, acc3) + dot3(gA, jer) + dot3(gJ, sna)) * alpha * node * step + node * integral_frac * pena;
            o[19] = node * step * pena;
        }
        __syncthreads();
        if (threadIdx.x < PARTIAL_STRIDE) {
            const int n = min(TL_THREADS, K1 - j0);
            for (int t = 0; t < n; t++) colsum += s_val[t][threadIdx.x];
        }
        __syncthreads();
    }
    if (threadIdx.x < PARTIAL_STRIDE) {
        const size_t ostride = (size_t)1 + 19 * (size_t)P.N;
        double *o = P.out + (size_t)b * ostride;
        const int q = threadIdx.x;
        if (q == 19) P.piece_cost[gp] = colsum;
        else if (q == 18) o[1 + i] = colsum;
        else {
            const int d = q / 6, r = q - 6 * d;
            o[1 + P.N + (size_t)d * 6 * P.N + 6 * i + r] = colsum;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// finalize_kernel: block per trajectory
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void finalize_kernel(const SweepParams P) {
    const int b = blockIdx.x;
    const size_t ostride = (size_t)1 + 19 * (size_t)P.N;
    double *o = P.out + (size_t)b * ostride;
    const int p0 = max(P.piece_begin, b * P.N) - b * P.N, p1 = min(P.piece_end, (b + 1) * P.N) - b * P.N;   // local piece range in shard
    // zero the entries of pieces that belong to other ranks
    for (int e = threadIdx.x; e < 19 * P.N; e += blockDim.x) {
        int i;
        if (e < P.N) i = e;
        else { const int r = (e - P.N) % (6 * P.N); i = r / 6; }
        if (i < p0 || i >= p1) o[1 + e] = 0.0;
    }
    if (b == 0) {   // the next step allocates from the other set of pool counters: reset it now
        unsigned *nxt = P.pool_count + (size_t)(P.parity ^ 1) * POOL_SHARDS * POOL_COUNTER_STRIDE;
        for (int k = threadIdx.x; k < POOL_SHARDS; k += blockDim.x) nxt[k * POOL_COUNTER_STRIDE] = 0u;
    }
    if (threadIdx.x < 64) {
        double s = 0.0;
        if (p1 > p0)
            for (int i = p0 + (int)threadIdx.x; i < p1; i += 64) s += P.piece_cost[(size_t)b * P.N + i];
        s = wave_sum(s);
        if (threadIdx.x == 0) o[0] = s;
    }
}

void launch_scan(const SweepParams &P, hipStream_t stream) {
    const long long n_samples = (long long)(P.piece_end - P.piece_begin) * (P.K + 1);
    if (n_samples <= 0 || !P.enable_pos) return;
    const int blocks = (int)((n_samples + TS_WAVES - 1) / TS_WAVES);
    switch (P.shape.kind) {
    case ISDF_SHAPE_BOX: hipLaunchKernelGGL(scan_kernel<ISDF_SHAPE_BOX>, dim3(blocks), dim3(TS_THREADS), 0, stream, P); break;
    case ISDF_SHAPE_ROUNDEDCONE: hipLaunchKernelGGL(scan_kernel<ISDF_SHAPE_ROUNDEDCONE>, dim3(blocks), dim3(TS_THREADS), 0, stream, P); break;
    default: hipLaunchKernelGGL(scan_kernel<-1>, dim3(blocks), dim3(TS_THREADS), 0, stream, P); break;
    }
}

void launch_exact(const SweepParams &P, hipStream_t stream) {
    const long long n_samples = (long long)(P.piece_end - P.piece_begin) * (P.K + 1);
    if (n_samples <= 0 || !P.enable_pos) return;
    const dim3 grid(POOL_SHARDS, EX_BLOCKS_PER_SHARD);
    switch (P.shape.kind) {
    case ISDF_SHAPE_BOX: hipLaunchKernelGGL(exact_kernel<ISDF_SHAPE_BOX>, grid, dim3(64 * EX_WAVES), 0, stream, P); break;
    case ISDF_SHAPE_ROUNDEDCONE: hipLaunchKernelGGL(exact_kernel<ISDF_SHAPE_ROUNDEDCONE>, grid, dim3(64 * EX_WAVES), 0, stream, P); break;
    default: hipLaunchKernelGGL(exact_kernel<-1>, grid, dim3(64 * EX_WAVES), 0, stream, P); break;
    }
}

void launch_stats(const SweepParams &P, hipStream_t stream) {
    hipLaunchKernelGGL(stats_kernel, dim3(64), dim3(256), 0, stream, P);
}

// qualifier bit-grid: bit z&31 of dword (ix*Y + iy)*ZW + (z>>5) is set iff the voxel qualifies.
// One lane per voxel, 64 consecutive z per wavefront: coalesced reads, the ballot is the two output dwords.
__global__ __launch_bounds__(256) void build_bits_kernel(DevGrid G, int use_esdf, float thresh, unsigned *out) {
    const int lane = threadIdx.x & 63;
    const size_t n_rows = (size_t)G.X * G.Y;
    const int chunks = (G.Z + 63) >> 6;                       // 64-voxel chunks per z-row
    const size_t n_work = n_rows * chunks;
    const size_t wave0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t w = wave0; w < n_work; w += n_waves) {
        const size_t row = w / chunks;
        const int ch = (int)(w - row * chunks);
        const int z = ch * 64 + lane;
        bool q = false;
        if (z < G.Z) {
            const size_t a = row * G.Z + z;
            q = use_esdf ? (G.esdf[a] <= thresh) : (G.occ[a] != 0);
        }
        const unsigned long long m = __ballot(q);
        if (lane == 0) {
            out[row * G.ZW + 2 * ch] = (unsigned)m;
            if (2 * ch + 1 < G.ZW) out[row * G.ZW + 2 * ch + 1] = (unsigned)(m >> 32);
        }
    }
}
void launch_build_bits(const DevGrid &G, int use_esdf, float thresh, unsigned *out, hipStream_t stream) {
    hipLaunchKernelGGL(build_bits_kernel, dim3(2048), dim3(256), 0, stream, G, use_esdf, thresh, out);
}

void launch_tail(const SweepParams &P, hipStream_t stream) {
    const int n_pieces = P.piece_end - P.piece_begin;
    if (n_pieces <= 0) return;
    hipLaunchKernelGGL(tail_kernel, dim3(n_pieces), dim3(TL_THREADS), 0, stream, P);
}

void launch_finalize(const SweepParams &P, hipStream_t stream) {
    hipLaunchKernelGGL(finalize_kernel, dim3(P.n_traj), dim3(256), 0, stream, P);
}

} // namespace isdf
