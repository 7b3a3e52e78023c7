// V2 / V3 integral sweep on gfx950: the P x (K+1) constraint-sample loop of
//   addTimeIntPenaltyParallel   (back_end_optimizer.hpp:432-554)
// with the per-pose tile term grad_cost_p (:766-824) plugged in before the flatness back-prop (:536), the voxel
// qualifier being occupancy (V2) or esdf <= occ_thresh (V3).
//
// One optimizer step = ONE launch when all its workgroups are resident (one trajectory, a rank's shard, small batches: the
// tail's workgroups ride behind the sweep's, FUSED), otherwise two (DESIGN.md "integral sweep"):
//   sweep_kernel<KIND, IDENT, FUSED>  one WORKGROUP per 4 constraint samples, one wavefront per sample:
//                     (1) poses: one LANE per sample runs the serial fp64 chain (quintic, flatness forward, rotation, the
//                         tile index box of getPointsInAABB, optional whole-tile cull, the window of tile rows the
//                         shape's inflated bounding box can reach) and publishes it through LDS;
//                     (2) one lane per tile row: row pruning against the shape's inflated body-frame box, the row's
//                         qualifier bits (1-2 dwords of the bit-grid, requested one round ahead), DPP prefix-sum
//                         compaction of the surviving voxels into an LDS ring;
//                     (3) conservative fp32 filter on full wavefronts (body-frame transform, cube cull, robot SDF);
//                     (4) survivors, in blocks of up to 64 of one sample (partial blocks of like size share a wavefront
//                         iteration), get the exact fp64 treatment of grad_cost_p, dealt over the workgroup's waves: voxel centre, R^T(c - x), cube cull (:800-803), robot SDF, smoothedL1 and - when the
//                         penalty is active - the central-difference gradient and the pose-gradient terms (:805-820);
//                     (5) wave butterfly -> acc[sample][8].
//                   No read-modify-write atomics, no work lists: results are bitwise reproducible.
//   tail_piece      (tail_kernel, or the trailing workgroups of a fused launch) one workgroup per piece, one thread per
//                   sample: dynamics penalties, flatness reverse mode, beta (x) grad scatter, then a fixed-order in-block
//                   sum over the piece's K+1 samples written straight into gradC rows 6i..6i+5 and gradT(i).  Results
//                   change hands through self-resetting 8-byte slots (value == ready flag): the sweep's sums per sample,
//                   the piece costs that the last-dispatched workgroup of a trajectory adds up in fixed order, and - in a
//                   multi-GPU step with the exchange inside the launch (XFuse) - the rows and costs the owner of a piece
//                   stores into every peer's board.  Workgroups of pieces outside this rank's shard write zeros (or, with
//                   XFuse, fetch the owner's rows) so that every rank ends up with, or sums to, the full result.
// Everything that reaches the result is fp64 except the stored ESDF (fp32).  No MFMA: gather/interpolate work.
#define ISDF_LEAN_MATH 1      // dev_math.hpp / dev_shapes.hpp: lean sqrt and normalisation in this translation unit
#ifndef ISDF_SWEEP_STATS
#define ISDF_SWEEP_STATS 0   // 1: per-wave statistics for tools/timing.py (developer builds)
#endif
#include "isdf_internal.hpp"
#include "dev_mesh.hpp"

namespace isdf {

constexpr int SW_LIST_CAP = 1024;                         // ring of compacted voxels (one pass adds <= 64 * SW_PASS)
constexpr int MQ_BLOCK = 16;                             // mesh robots, queued exact pass: voxels per work item = one quad round (dev_mesh.hpp)
constexpr int SW_PASS = 8;                                // voxels one lane contributes to the ring per pass
constexpr int SW_NEAR_CAP = 512;                          // per-sample candidate ring (drained by the whole workgroup)
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

// ---------------------------------------------------------------------------------------------------------------
// sweep_kernel: workgroup = SW_SAMPLES consecutive constraint samples, one wavefront each
// ---------------------------------------------------------------------------------------------------------------
// The grid is read through its QUALIFIER BIT-GRID (1 bit per voxel: esdf <= occ_thresh for V3, occupied for V2;
// built once per set_grid by build_bits_kernel), 32 voxels of a z-row per dword.
constexpr int SW_SAMPLES = 4;

// Hand-over between the sweep and the per-piece tail: every 8-byte result slot (the 8 collision sums of a sample, the cost
// of a piece) is its own "ready" flag.  A slot is created EMPTY (all-ones), the producer fills it with one agent-scope
// store, the consumer polls it, takes the value and empties it again - so the consumer may already be running while the
// producer still works (the fused launch below), and nothing has to be cleared between steps.
constexpr int TL_THREADS = 128;                        // tail: threads per piece (one per sample, K + 1 <= 128 per pass)
constexpr int TL_GROUPS = TL_THREADS / PARTIAL_STRIDE;     // 6 row groups x 20 columns
constexpr unsigned long long SLOT_EMPTY = ~0ull;
constexpr int SLOT_SPIN_MAX = 1 << 20;                 // bounded wait: a lost producer must not hang the stream
__device__ __forceinline__ void slot_publish(double *slot, double v) {
    unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    if (bits == SLOT_EMPTY) bits = 0x7FF8000000000000ull;          // an all-ones NaN would read as "not published"
    __hip_atomic_store((unsigned long long *)slot, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long slot_peek(const double *slot) {
    return __hip_atomic_load((const unsigned long long *)slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void slot_clear(double *slot) {
    __hip_atomic_store((unsigned long long *)slot, SLOT_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the same three for slots other GPUs write or read (boards of the in-kernel exchange: IPC-mapped uncached memory)
// peers may start their step later than this rank: the wait for a row is bounded by the device wall clock (XFuse::timeout_ticks =
// isdf_xchg_timeout_ms; 100 MHz, the same bound on every box), armed at a lane's first miss and shared by its later takes
__device__ __forceinline__ void xslot_publish(double *slot, double v) {
    unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    if (bits == SLOT_EMPTY) bits = 0x7FF8000000000000ull;
    __hip_atomic_store((unsigned long long *)slot, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ bool xslot_take(double *slot, double &v, int timeout_ticks, unsigned long long &deadline) {        // bounded poll, then hands the slot back empty
    unsigned long long b = __hip_atomic_load((unsigned long long *)slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (b == SLOT_EMPTY) {
        if (deadline == 0ull) deadline = wall_clock64() + (unsigned long long)timeout_ticks;
        while (b == SLOT_EMPTY && wall_clock64() <= deadline) {
            __builtin_amdgcn_s_sleep(4);
            b = __hip_atomic_load((unsigned long long *)slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    __hip_atomic_store((unsigned long long *)slot, SLOT_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    v = __longlong_as_double((long long)b);
    return b != SLOT_EMPTY;
}

// ---- inputs of one piece: from the caller's device arrays, or - host-direct step - from the staging buffer the first
// workgroups of the launch fill from host-mapped memory
struct PieceIn { const double *c; int ld; double T; };
__device__ __forceinline__ PieceIn piece_in(const SweepParams &P, int b, int i) {
    PieceIn r;
    if (P.host_T) { const double *s = P.stage + ((size_t)b * P.N + i) * 19; r.c = s + 1; r.ld = 6; r.T = s[0]; }
    else { r.c = P.coeffs + (size_t)b * 18 * P.N + 6 * i; r.ld = 6 * P.N; r.T = P.T[(size_t)b * P.N + i]; }
    return r;
}
// one lane: wait (bounded) until the group of global piece `gp` has been staged, then make the staged doubles readable
__device__ __forceinline__ bool stage_wait(const SweepParams &P, int gp) {
    const unsigned long long *f = P.stage_flags + gp / STAGE_G;
    bool ok = false;
    for (int spin = 0; spin < SLOT_SPIN_MAX; spin++) {
        if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == P.seq) { ok = true; break; }
        __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return ok;
}
// workgroup `blk` of a host-direct launch copies the inputs of pieces [blk * STAGE_G, +STAGE_G) host -> stage, raises their flag
__device__ __forceinline__ void stage_in(const SweepParams &P, int blk) {
    const int n_pieces = P.n_traj * P.N;
    for (int e = threadIdx.x; e < STAGE_G * 19; e += blockDim.x) {
        const int gp = blk * STAGE_G + e / 19, k = e % 19;
        if (gp < n_pieces) {
            const int b = gp / P.N, i = gp - b * P.N;
            double v;
            if (k == 0) v = P.host_T[gp];
            else { const int d = (k - 1) / 6, r = (k - 1) - 6 * d; v = P.host_coeffs[(size_t)b * 18 * P.N + (size_t)d * 6 * P.N + 6 * i + r]; }
            P.stage[(size_t)gp * 19 + k] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_store(P.stage_flags + blk, P.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Row pruning needs "sdf(p) >= distance(p, bbox)".  The mesh kind's sdf is (1 - 2 w) * distance with libigl's APPROXIMATE
// winding number (|w error| up to a few 1e-3 outside, Shape.cpp:110), so its penalty band is widened by 5 % for pruning.
__device__ __forceinline__ double mesh_band_scale(int kind) { return kind == ISDF_SHAPE_MESH ? 1.05 : 1.0; }

struct SweepPose {                 // LDS record of one sample, written by the pose lane
    double pos[3];
    double R[9];                   // row-major body->world rotation
    double quat[4];                // w x y z
    int lo[3];                     // first voxel of the tile (getPointsInAABB box)
    int nzt;                       // tile extent along the scan axis
    int rx0, ry0, rnx, rny;        // window of tile rows (u, v) that can reach the shape's inflated bounding box
    int flags;                     // bit0 active, bit1 culled
    int sa;                        // scan axis: rows run along world axis sa, (u, v) = the other two axes, ascending
    int nn[3];                     // tile extent per axis (0 or less: no tile)
    int pad;
    double L0[3], L1[3];           // extent of the shape's inflated bounding box on each world axis, in voxels from the map's first
    // fp32 constants of the scan stage (scan_consts): filled by pose_kernel, one THREAD per sample, and fetched by the sweep's waves
    // with scalar loads - worked out per wave they are ~150 wave-uniform VALU / LDS instructions of a kernel that is bound by its
    // instruction count.  (A fused step works them out per wave, from the record in LDS: these fields are unused there.)
    float Rf[9], pv[3];            // rotation; position in the tile's voxel frame, pre-multiplied (fp32 filter)
    float Ru[3], Rv[3], Rw[3];     // body-frame components of the tile axes as the rows see them: along u, along v, along the scan axis
    float pvu, pvv, pvw;
    float dzv[3], dzi[3];          // body-frame step of one voxel along a row, and its reciprocals
    float inv_rny;
    float cone_alpha, cone_ia;     // rounded cone: the row's quadratic (see row_mask)
    int cone_on;
};
static_assert(sizeof(SweepPose) % 8 == 0 && sizeof(SweepPose) <= 64 * 8, "the record is copied by one wavefront, 8 bytes per lane");

// launch-wide fp32 constants of the scan stage
struct ScanLaunch { float resf, bdh_f, thr_f, cone_A, cone_B; bool cone_shape; };
__device__ __forceinline__ ScanLaunch scan_launch_consts(const SweepParams &P, int kind) {
#pragma clang fp contract(off)
    ScanLaunch L;
    L.resf = (float)P.grid.res;
    L.bdh_f = (float)P.bd_half + TS_FILTER_EPS;
    L.thr_f = (float)P.safety_hor * (float)mesh_band_scale(kind) + TS_FILTER_EPS;
    L.cone_A = 0.0f; L.cone_B = 0.0f; L.cone_shape = false;
    if (kind == ISDF_SHAPE_ROUNDEDCONE) {
        // RoundedCone (Shape.hpp:1018-1029; axis z, sphere r1 at the origin, r2 at z = h): inflated by the band it is still a rounded
        // cone, and it lies inside the cone that is tangent to both spheres, rho <= A - B z with B = b / a, A = (r1 + band) / a,
        // b = (r1 - r2) / h, a = sqrt(1 - b^2).
        const float r1 = P.shape.f.p[0], r2 = P.shape.f.p[1], hh = P.shape.f.p[2];
        const float b = (r1 - r2) / hh, a2 = 1.0f - b * b;
        if (P.shape.prune_rows != 0 && hh > 0.0f && a2 > 1.0e-3f) {
            const float ia = 1.0f / sqrtf(a2);
            L.cone_A = (r1 + L.thr_f) * ia; L.cone_B = b * ia; L.cone_shape = true;
        }
    }
    return L;
}
// The per-sample fp32 constants of the scan stage from the pose record's position, rotation, tile origin, scan axis and window.
// ONE definition for pose_kernel and the fused step, contraction off: both must produce the same bits (the candidate lists, and
// with them the summation order of the exact pass, depend on them - a batch stays bitwise equal to its trajectories one by one).
__device__ __forceinline__ void scan_consts(const SweepParams &P, const ScanLaunch &L, const double *pos, const double *R, const int *lo, int sa, int rny,
                                            bool cone_ident, SweepPose &o) {
#pragma clang fp contract(off)
    const DevGrid &G = P.grid;
#pragma unroll
    for (int k = 0; k < 9; k++) o.Rf[k] = (float)R[k];
    // m = l*res - pvr,  pvr = ((pos - bmin)/res - 0.5 - lo)*res  (fp32 filter stage only: conservative by TS_FILTER_EPS, so the
    // algebraically equal form without divisions is fine)
#pragma unroll
    for (int a = 0; a < 3; a++) o.pv[a] = (float)(pos[a] - G.bmin[a] - (0.5 + lo[a]) * G.res);
#pragma unroll
    for (int a = 0; a < 3; a++) {
        o.Ru[a] = sa == 0 ? o.Rf[3 + a] : o.Rf[a];
        o.Rv[a] = sa == 2 ? o.Rf[3 + a] : o.Rf[6 + a];
        o.Rw[a] = sa == 2 ? o.Rf[6 + a] : (sa == 0 ? o.Rf[a] : o.Rf[3 + a]);
        o.dzv[a] = o.Rw[a] * L.resf;
        o.dzi[a] = 1.0f / o.dzv[a];
    }
    o.pvu = sa == 0 ? o.pv[1] : o.pv[0]; o.pvv = sa == 2 ? o.pv[1] : o.pv[2]; o.pvw = sa == 2 ? o.pv[2] : (sa == 0 ? o.pv[0] : o.pv[1]);
    o.inv_rny = 1.0f / (float)max(rny, 1);
    o.cone_alpha = 0.0f; o.cone_ia = 0.0f; o.cone_on = 0;
    if (cone_ident && L.cone_shape) {
        // a row's part inside the tangent cone is one more interval (a quadratic in the row coordinate), much tighter than the
        // bounding box around the thin end: 141 -> ~70 listed voxels per sample.  Only where the row is flatter than the cone
        // (alpha > 0: the inside is BETWEEN the roots).
        o.cone_alpha = o.dzv[0] * o.dzv[0] + o.dzv[1] * o.dzv[1] - L.cone_B * L.cone_B * o.dzv[2] * o.dzv[2];
        o.cone_on = o.cone_alpha > 1.0e-3f * (o.dzv[0] * o.dzv[0] + o.dzv[1] * o.dzv[1] + o.dzv[2] * o.dzv[2]) ? 1 : 0;
        o.cone_ia = 1.0f / o.cone_alpha;
    }
}

__device__ __forceinline__ double uni(double v) {     // wave-uniform double -> scalar registers
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float unif(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }


// Dedicated instantiations (Box, RoundedCone and the robots of the shipped configs) are held to 128 VGPRs (4 waves per SIMD),
// their identity-offset variants to 102 (5 waves) - worth 10-30 % when the device is full (C3); the generic and mesh
// instantiations need more and are held to 3 waves.
// waves per SIMD the register allocator is held to, by formula size (measured: no scratch spills at these settings except a
// few dwords for the smooth-min shapes): 5 needs <= 102 VGPRs, 4 <= 128, 3 <= 168
constexpr int sweep_waves_per_simd(int kind, bool ident) {
    switch (kind) {
    case ISDF_SHAPE_TORUS: case ISDF_SHAPE_CAPPEDTORUS: case ISDF_SHAPE_ROUNDEDCONE: case ISDF_SHAPE_BENDLINEAR:
    case ISDF_SHAPE_SMOOTHDIFFERENCE: case ISDF_SHAPE_SMOOTHINTERSECTION: return ident ? 5 : 4;
    case ISDF_SHAPE_BOX: return ident ? 6 : 4;      // 86 registers as it was: one more than a sixth wave allows - held to 85, the batch gains 6 %
    case ISDF_SHAPE_CAPPEDCONE: case ISDF_SHAPE_WIREFRAMEBOX: return 4;
    case ISDF_SHAPE_TWISTBOX: case ISDF_SHAPE_BENDBOX: case ISDF_SHAPE_TABLE: case ISDF_SHAPE_TREFOIL: case ISDF_SHAPE_CSG: return ident ? 4 : 3;
    default: return 3;      // mesh, generic
    }
}

// Longest-first dispatch order of a fused single-launch step.  Such a launch is resident as a whole, 2 - 3 workgroups per CU,
// and a CU's instruction issue is what its workgroups share: the step lasts as long as the CU with the most work.  The
// dispatcher hands workgroup b to XCD b % 8 and, there, to the next CU in turn, so workgroups [0, #CUs) are the first on their
// CU: they get the heaviest samples, everyone after them the lighter ones (measured with orders built on the host,
// tools/lpt_experiment.py: rounded cone 33.1 -> 26.3 us per step, box 22.2 -> 21.5; orders that only even out the workgroups,
// per-piece quotas, or a launch padded to the same number of workgroups on every CU with evened-out workgroups gain less or nothing).  Order = stable counting sort of the samples by work class (exact pairs / 32 of an EARLIER
// step - the trajectory moves little between optimizer steps), classes descending, trajectory order inside a class
// (neighbouring workgroups keep neighbouring tiles).  Built by the tail workgroups beside their own work, in two generations
// so that no workgroup needs another one's output of the same launch (and none scans more than its own piece):
//   step k:     sweep waves leave the class of every sample (cls);
//   step k + 1: the tail workgroup of piece i turns its samples' classes into (class, rank inside piece and class) records
//               and the piece's class histogram (lr, hist);
//   step k + 2: it adds the other pieces' histograms - total per class, pieces before i - and writes the rank of each of its
//               samples in the launch (map); step k + 3 is dispatched in that order.
// Scheduling only: which wave evaluates a sample has no influence on any result.
__device__ __forceinline__ int plan_rank(int pos, int Z, int nb) {
    const int q0 = 4 * Z;
    if (pos < q0) { const int w = pos / Z; return w * nb + (pos - w * Z); }
    const int n3 = nb - Z, p1 = pos - q0, w = p1 / n3;
    return w * nb + Z + (p1 - w * n3);
}
// One wavefront per piece does it (the tail's fourth wave, which otherwise only keeps the barriers), with wave-level operations
// only: lane l looks after samples l and l + 64 of the piece, and after pieces l, l + 64, ... when the histograms are added.
__device__ __forceinline__ uint4 plan_add(uint4 a, uint4 b) {      // 8 counts of 16 bits; no sum exceeds PLAN_MAX_SAMPLES
    return make_uint4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
__device__ __forceinline__ uint4 plan_wave_sum(uint4 v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) v = plan_add(v, make_uint4(__shfl_xor(v.x, d), __shfl_xor(v.y, d), __shfl_xor(v.z, d), __shfl_xor(v.w, d)));
    return v;
}
__device__ __forceinline__ unsigned plan_count(uint4 v, int q) {
    const unsigned w = (q >> 1) == 0 ? v.x : (q >> 1) == 1 ? v.y : (q >> 1) == 2 ? v.z : v.w;
    return (w >> (16 * (q & 1))) & 0xFFFFu;
}
// Launches that are NOT resident as a whole (a 400-piece trajectory, a batch, the shard of a multi-GPU step, every two-launch
// step) use the same records differently (plan_zone = 0): the sorted order is cut into workgroups of four NEIGHBOURS - samples
// of like weight share a workgroup, so its four scans end together instead of three waves idling at the barrier for the
// fourth, and the heaviest workgroups are dispatched first, so the launch does not end on them (measured with host-built
// orders, tools/lpt_big_experiment.py: 400 pieces 106 -> 94 us, 100 pieces 37.8 -> 33.8).  Sorting is per GROUP of pieces (the
// whole launch, or one trajectory of a batch: at most 65 535 samples, so that the packed 16-bit counts cannot overflow).
__device__ void plan_wave(const SweepParams &P, const int t, const int lane) {      // t: local piece (0 = this rank's first)
    const int K1 = P.K + 1, n_pieces = P.piece_end - P.piece_begin;
    const int ns = n_pieces * K1, nb = P.n_sweep_blocks, Z = P.plan_zone;
    const int G = P.plan_group, gp0 = (t / G) * G, gn = min(G, n_pieces - gp0);      // this piece's group: pieces [gp0, gp0 + gn)
    const int s0 = t * K1 + lane, s1 = s0 + 64;
    const bool h0 = lane < K1, h1 = lane + 64 < K1;
    if (P.plan_cls_in) {
        // ---- generation g: classes -> (class, rank inside piece and class) records, histogram of the piece
        const int c0 = h0 ? (int)(P.plan_cls_in[s0] & (PLAN_CLASSES - 1)) : -1, c1 = h1 ? (int)(P.plan_cls_in[s1] & (PLAN_CLASSES - 1)) : -1;
        int r0 = 0, r1 = 0, hist = 0;
        const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
        for (int q = 0; q < PLAN_CLASSES; q++) {
            const unsigned long long m0 = __ballot(c0 == q), m1 = __ballot(c1 == q);
            if (c0 == q) r0 = __popcll(m0 & lt);
            if (c1 == q) r1 = __popcll(m0) + __popcll(m1 & lt);
            if (lane == q) hist = __popcll(m0) + __popcll(m1);
        }
        if (h0) P.plan_lr_out[s0] = (unsigned short)(c0 | (r0 << 3));
        if (h1) P.plan_lr_out[s1] = (unsigned short)(c1 | (r1 << 3));
        if (lane < PLAN_CLASSES) P.plan_hist_out[t * PLAN_CLASSES + lane] = (unsigned short)hist;
    }
    if (P.plan_lr_in) {
        // ---- generation g - 1: the group's histograms -> first position of this piece's samples of each class -> ranks
        uint4 tot = make_uint4(0, 0, 0, 0), bef = tot;
        for (int i = gp0 + lane; i < gp0 + gn; i += 64) {
            const uint4 h = reinterpret_cast<const uint4 *>(P.plan_hist_in)[i];
            tot = plan_add(tot, h);
            if (i < t) bef = plan_add(bef, h);
        }
        tot = plan_wave_sum(tot); bef = plan_wave_sum(bef);
        int base = 0;                                // lane q < 8: classes descending
        if (lane < PLAN_CLASSES) {
            base = (int)plan_count(bef, lane);
#pragma unroll
            for (int q = 1; q < PLAN_CLASSES; q++) if (q > lane) base += (int)plan_count(tot, q);
        }
        // (the shuffles run with all lanes: a lane may need the base a lane without a sample holds)
        const unsigned v0 = h0 ? P.plan_lr_in[s0] : 0u, v1 = h1 ? P.plan_lr_in[s1] : 0u;
        const int b0 = __shfl(base, (int)(v0 & 7u)), b1 = __shfl(base, (int)(v1 & 7u));
        const int p0 = gp0 * K1 + b0 + (int)(v0 >> 3), p1 = gp0 * K1 + b1 + (int)(v1 >> 3);      // positions in the launch's order
        if (Z > 0) {
            if (h0) P.plan_map_out[plan_rank(p0, Z, nb)] = s0;
            if (h1) P.plan_map_out[plan_rank(p1, Z, nb)] = s1;
            if (t == 0) for (int pos = ns + lane; pos < 4 * nb; pos += 64) P.plan_map_out[plan_rank(pos, Z, nb)] = -1;     // ranks behind the last sample stay empty
        } else {
            if (h0) P.plan_map_out[(p0 & 3) * nb + (p0 >> 2)] = s0;
            if (h1) P.plan_map_out[(p1 & 3) * nb + (p1 >> 2)] = s1;
            if (t == 0) for (int pos = ns + lane; pos < 4 * nb; pos += 64) P.plan_map_out[(pos & 3) * nb + (pos >> 2)] = -1;
        }
    }
}

// Poses of a launch that is not one fused step, one THREAD per sample: exactly the record the fused kernel's first stage builds
// in LDS (same functions, same operations: a batch stays bitwise equal to its trajectories evaluated one by one) - the serial
// chain position -> flatness -> rotation -> where the shape's inflated bounding box lies on each axis, the tile's index box as
// getPointsInAABB (PCSmap_manager.h:150-155), the window of rows and the scan axis.
constexpr int POSE_THREADS = 64;
constexpr int POSE_REC = (int)(sizeof(SweepPose) / 8);       // doubles per record
__global__ __launch_bounds__(POSE_THREADS) void pose_kernel(const SweepParams P) {
    // the records leave through LDS: a thread's own record is 368 bytes away from its neighbour's - stored from the registers it
    // is 46 scattered 8-byte writes per thread; the workgroup's records are one contiguous range, written out lane after lane
    __shared__ double s_rec[POSE_THREADS][POSE_REC + 1];
    const int K1 = P.K + 1;
    const long long s_begin = (long long)P.piece_begin * K1, s_end = (long long)P.piece_end * K1;
    const long long s_first = s_begin + (long long)blockIdx.x * POSE_THREADS;
    const long long sidx = s_first + threadIdx.x;
    if (P.mq_items && blockIdx.x == 0 && threadIdx.x == 0) { P.mq_count[0] = 0u; P.mq_count[1] = 0u; }      // mesh robots: this step's queue of exact-pass blocks starts empty (items queued / items claimed)
    if (sidx < s_end) {
    const DevGrid &G = P.grid;
    SweepPose &sp = *reinterpret_cast<SweepPose *>(&s_rec[threadIdx.x][0]);       // built in place (a local copy ends up in scratch memory)
#pragma unroll
    for (int k = 0; k < POSE_REC; k++) s_rec[threadIdx.x][k] = 0.0;
    const SampleId id = decode_sample(P, sidx);
    const PieceIn pin = piece_in(P, id.b, id.i);
    const double *c = pin.c;
    const int ld = pin.ld;
    const double step = pin.T * P.inv_K;
    Basis B;
    poly_basis(id.j * step, B);
    const d3 pos = ct_beta(c, ld, B.b0), vel = ct_beta<1>(c, ld, B.b1), acc = ct_beta<2>(c, ld, B.b2);
    FlatS fs;
    flat_core(P.flat, vel, acc, fs);
    const d4 quat = flat_quat(fs);
    const m3 R = quat_to_rot(quat);
    sp.pos[0] = pos.x; sp.pos[1] = pos.y; sp.pos[2] = pos.z;
#pragma unroll
    for (int k = 0; k < 9; k++) sp.R[k] = R.a[k];
    sp.quat[0] = quat.w; sp.quat[1] = quat.x; sp.quat[2] = quat.y; sp.quat[3] = quat.z;
    int flags = 0;
    if (P.enable_pos) {
        flags = 1;
        if (P.cull_threshold > 0.0 && esdf_trilinear(G, pos) > P.cull_threshold) flags = 2;
    }
    if ((flags & 1) && P.shape.prune_rows) {
        const double infl = P.safety_hor * mesh_band_scale(P.shape.kind) + 4.0e-3 + 0.05 * G.res;
        const double inv_res = m_rcp(G.res);
        const double hb[3] = {0.5 * ((double)P.shape.bbox_hi[0] - (double)P.shape.bbox_lo[0]) + infl,
                              0.5 * ((double)P.shape.bbox_hi[1] - (double)P.shape.bbox_lo[1]) + infl,
                              0.5 * ((double)P.shape.bbox_hi[2] - (double)P.shape.bbox_lo[2]) + infl};
        const double cb[3] = {0.5 * ((double)P.shape.bbox_hi[0] + (double)P.shape.bbox_lo[0]),
                              0.5 * ((double)P.shape.bbox_hi[1] + (double)P.shape.bbox_lo[1]),
                              0.5 * ((double)P.shape.bbox_hi[2] + (double)P.shape.bbox_lo[2])};
        const double p[3] = {pos.x, pos.y, pos.z};
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const double ctr = p[a] + R.a[3 * a] * cb[0] + R.a[3 * a + 1] * cb[1] + R.a[3 * a + 2] * cb[2];
            const double ext = fabs(R.a[3 * a]) * hb[0] + fabs(R.a[3 * a + 1]) * hb[1] + fabs(R.a[3 * a + 2]) * hb[2];
            sp.L0[a] = (ctr - ext - G.bmin[a]) * inv_res - 0.5;
            sp.L1[a] = (ctr + ext - G.bmin[a]) * inv_res - 0.5;
        }
    }
    int nn[3] = {0, 0, 0};
    if (P.enable_pos) {
        const double pa[3] = {pos.x, pos.y, pos.z};
        const int dims[3] = {G.X, G.Y, G.Z};
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const int lo = axis_index(clampd(pa[a] - P.bd_half, G.bmin[a], G.bmax[a]), G.bmin[a], G.res, dims[a]);
            const int hi = axis_index(clampd(pa[a] + P.bd_half, G.bmin[a], G.bmax[a]), G.bmin[a], G.res, dims[a]);
            sp.lo[a] = lo;
            sp.nn[a] = nn[a] = hi - lo + 1;
        }
    }
    // window of rows and scan axis (the fused kernel's per-wave finish)
    int flags_w = flags;
    if ((flags_w & 1) && !(nn[0] > 0 && nn[1] > 0 && nn[2] > 0)) flags_w &= ~1;
    int w0[3] = {0, 0, 0}, w1[3] = {nn[0] - 1, nn[1] - 1, nn[2] - 1};
    const bool windowed = (flags_w & 1) && P.shape.prune_rows;
    if (windowed) {
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const double l0 = sp.L0[a] - sp.lo[a], l1 = sp.L1[a] - sp.lo[a];
            if (l0 > 0.0) w0[a] = (int)fmin(floor(l0), (double)nn[a]);
            if (l1 < (double)(nn[a] - 1)) w1[a] = (int)fmax(ceil(l1), -1.0);
        }
    }
    int sa_w = 2;
    if (windowed) {
        const int e0 = w1[0] - w0[0], e1 = w1[1] - w0[1], e2 = w1[2] - w0[2];
        if (P.shape.kind != ISDF_SHAPE_MESH) sa_w = (e2 >= e0 && e2 >= e1) ? 2 : (e0 >= e1 ? 0 : 1);
    }
    const int u0 = sa_w == 0 ? w0[1] : w0[0], u1 = sa_w == 0 ? w1[1] : w1[0];
    const int v0 = sa_w == 2 ? w0[1] : w0[2], v1 = sa_w == 2 ? w1[1] : w1[2];
    sp.rx0 = (flags_w & 1) ? u0 : 0; sp.ry0 = (flags_w & 1) ? v0 : 0;
    sp.rnx = (flags_w & 1) ? max(0, u1 - u0 + 1) : 0; sp.rny = (flags_w & 1) ? max(0, v1 - v0 + 1) : 0;
    sp.nzt = sa_w == 2 ? nn[2] : (sa_w == 0 ? nn[0] : nn[1]);
    sp.flags = flags_w;
    sp.sa = sa_w;
    scan_consts(P, scan_launch_consts(P, P.shape.kind), sp.pos, sp.R, sp.lo, sa_w, sp.rny, P.shape.kind == ISDF_SHAPE_ROUNDEDCONE && P.shape.f.ident != 0, sp);
    }
    __syncthreads();
    const int n_rec = (int)min((long long)POSE_THREADS, s_end - s_first);
    double *dst = (double *)((SweepPose *)P.poses + (s_first - s_begin));
    for (int i = threadIdx.x; i < n_rec * POSE_REC; i += POSE_THREADS) dst[i] = s_rec[i / POSE_REC][i % POSE_REC];
}
size_t sweep_pose_bytes() { return sizeof(SweepPose); }

// IDENT: the shape's body offset is the identity (poly_params all zero - every shipped config): those instantiations drop
// the offset and the 24 scalar registers it pins, which takes Box to 91 VGPRs and a fifth wave per SIMD.
template <bool EARLY, bool XF> __device__ __forceinline__ void tail_piece(const SweepParams &P, const int t, double (*s_val)[PARTIAL_STRIDE + 1]);

// FUSED: the launch carries, after its P.n_sweep_blocks sweep workgroups, one workgroup per piece that runs the tail
// (tail_piece) - dispatched last, it does the collision-independent part of its samples while the sweep works and picks the
// collision sums up from their slots as they arrive, so a small step (one trajectory) is one launch without the inter-kernel
// gap.  Only launches whose workgroups are all resident at 3 per CU use it (launch_sweep); register budget: 3 waves per SIMD.
template <int KIND, bool IDENT, bool FUSED>
__global__ __launch_bounds__(64 * SW_SAMPLES, FUSED ? 3 : sweep_waves_per_simd(KIND, IDENT)) void sweep_kernel(const SweepParams P) {
    const int n_blocks = FUSED ? P.n_sweep_blocks : (int)gridDim.x;       // sweep workgroups of this launch
    if constexpr (FUSED) {
        __shared__ double s_tail[TL_THREADS][PARTIAL_STRIDE + 1];
        if ((int)blockIdx.x >= n_blocks) {
            if ((P.plan_cls_in || P.plan_lr_in) && !(P.dbg_flags & 2) && (threadIdx.x >> 6) == SW_SAMPLES - 1) plan_wave(P, (int)blockIdx.x - n_blocks, threadIdx.x & 63);       // while the sweep works: the NEXT step's dispatch order
            if (P.xf.world > 1) tail_piece<false, true>(P, (int)blockIdx.x - n_blocks, s_tail);      // multi-GPU step: exchange inside
            else tail_piece<false, false>(P, (int)blockIdx.x - n_blocks, s_tail);
            return;
        }
    }
    if constexpr (FUSED) {
        // host-direct step: the first workgroups bring the inputs in (one PCIe read round trip, instead of a copy command and
        // its completion in front of the launch)
        if (P.host_T && (int)blockIdx.x < (P.n_traj * P.N + STAGE_G - 1) / STAGE_G) stage_in(P, (int)blockIdx.x);
    }
    __shared__ SweepPose s_pose[SW_SAMPLES];
    __shared__ unsigned s_list[SW_SAMPLES][SW_LIST_CAP];          // ring of compacted voxels awaiting the fp32 test
    __shared__ unsigned s_near[SW_SAMPLES][SW_NEAR_CAP];          // ring of candidates awaiting the exact pass
    __shared__ __attribute__((aligned(16))) int s_plan[SW_SAMPLES][4];                         // per sample: candidates, ring head, finished
    __shared__ double s_res[SW_SAMPLES][SW_NEAR_CAP / 64][ACC_STRIDE];   // per chunk: the 8 collision sums
    __shared__ int s_resn[SW_SAMPLES][SW_NEAR_CAP / 64][2];       // per chunk: exact pairs, active pairs
    __shared__ __attribute__((aligned(8))) unsigned s_mframes[KIND == ISDF_SHAPE_MESH ? SW_SAMPLES : 1][KIND == ISDF_SHAPE_MESH ? mesh_frames_words(MESH_Q_LEVELS) : 2];   // mesh robots: the quad walks' frames, one set per wavefront

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int K1 = P.K + 1;
    const long long s_begin = (long long)P.piece_begin * K1, s_end = (long long)P.piece_end * K1;
    const DevGrid &G = P.grid;
    // rank of this wave in the launch -> its sample: b + w * n_blocks (the samples of one workgroup are far apart on the
    // trajectory, so their candidate counts are uncorrelated), or through the longest-first order of plan_piece
    const long long rank_w = (long long)blockIdx.x + (long long)wave * n_blocks;
    long long sidx = s_begin + rank_w;
    if (P.sample_map) { const int m = __builtin_amdgcn_readfirstlane(P.sample_map[rank_w]); sidx = m >= 0 ? s_begin + m : s_end; }
    // (time stamps and developer switches: ISDF_SWEEP_STATS builds only - the pointer alone is two scalar registers for the whole kernel)
    unsigned long long *const dbg = (ISDF_SWEEP_STATS && P.dbg && sidx < s_end) ? P.dbg + (size_t)(sidx - s_begin) * 8 : nullptr;
    if (dbg && lane == 0) { dbg[0] = wall_clock64(); dbg[1] = clock64(); }

    // ---- poses of the workgroup's samples: one LANE per sample (the serial fp64 chain is issued once, not per wave)
    // ---- poses of the workgroup's samples.  Three parties share the serial fp64 work (it is the launch's first 2.5 us, with
    // nothing else to run beside it):
    //   wave 0, one LANE per sample: quintic, flatness forward, rotation, and - in floating point - where the shape's
    //                                inflated bounding box lies on each world axis;
    //   wave 1, one lane per (sample, axis): position component -> the tile's index box on that axis, exactly as
    //                                getPointsInAABB (PCSmap_manager.h:150-155);
    //   after the barrier every wave finishes ITS sample: integer window of rows, scan axis.
    auto sample_of_rank = [&](long long rank) -> long long {
        if (P.sample_map) { const int m = P.sample_map[rank]; return m >= 0 ? s_begin + m : s_end; }
        return s_begin + rank;
    };
    if constexpr (FUSED) {
    long long pose_sidx = s_end;
    if (wave == 0 && lane < SW_SAMPLES) pose_sidx = sample_of_rank((long long)blockIdx.x + (long long)lane * n_blocks);
    if (pose_sidx < s_end) {
        const SampleId id = decode_sample(P, pose_sidx);
        bool staged = true;
        if constexpr (FUSED) if (P.host_T) staged = stage_wait(P, id.b * P.N + id.i);
        if (!staged && P.stats) atomicMax(&P.stats[4], 1ull);
        const PieceIn pin = piece_in(P, id.b, id.i);
        const double *c = pin.c;
        const int ld = pin.ld;
        const double step = pin.T * P.inv_K;
        Basis B;
        poly_basis(id.j * step, B);
        const d3 pos = ct_beta(c, ld, B.b0), vel = ct_beta<1>(c, ld, B.b1), acc = ct_beta<2>(c, ld, B.b2);
        FlatS fs;
        flat_core(P.flat, vel, acc, fs);
        const d4 quat = flat_quat(fs);
        const m3 R = quat_to_rot(quat);
        SweepPose &sp = s_pose[lane];
        sp.pos[0] = pos.x; sp.pos[1] = pos.y; sp.pos[2] = pos.z;
#pragma unroll
        for (int k = 0; k < 9; k++) sp.R[k] = R.a[k];
        sp.quat[0] = quat.w; sp.quat[1] = quat.x; sp.quat[2] = quat.y; sp.quat[3] = quat.z;
        int flags = 0;
        if (P.enable_pos) {
            flags = 1;
            if (P.cull_threshold > 0.0 && esdf_trilinear(G, pos) > P.cull_threshold) flags = 2;
        }
        sp.flags = flags;      // bit 0 is confirmed by the wave that finishes the sample (the tile must exist)
        if ((flags & 1) && P.shape.prune_rows) {
            // world-frame extent of the shape's body-frame bounding box inflated by the penalty band: rows whose
            // (u, v) lies outside it cannot hold a voxel with a non-zero penalty (conservative: +-1 voxel, NaN keeps all)
            const double infl = P.safety_hor * mesh_band_scale(P.shape.kind) + 4.0e-3 + 0.05 * G.res;
            const double inv_res = m_rcp(G.res);     // the window has 0.05 voxel of slack: one reciprocal serves its quotients
            const double hb[3] = {0.5 * ((double)P.shape.bbox_hi[0] - (double)P.shape.bbox_lo[0]) + infl,
                                  0.5 * ((double)P.shape.bbox_hi[1] - (double)P.shape.bbox_lo[1]) + infl,
                                  0.5 * ((double)P.shape.bbox_hi[2] - (double)P.shape.bbox_lo[2]) + infl};
            const double cb[3] = {0.5 * ((double)P.shape.bbox_hi[0] + (double)P.shape.bbox_lo[0]),
                                  0.5 * ((double)P.shape.bbox_hi[1] + (double)P.shape.bbox_lo[1]),
                                  0.5 * ((double)P.shape.bbox_hi[2] + (double)P.shape.bbox_lo[2])};
            const double p[3] = {pos.x, pos.y, pos.z};
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const double ctr = p[a] + R.a[3 * a] * cb[0] + R.a[3 * a + 1] * cb[1] + R.a[3 * a + 2] * cb[2];
                const double ext = fabs(R.a[3 * a]) * hb[0] + fabs(R.a[3 * a + 1]) * hb[1] + fabs(R.a[3 * a + 2]) * hb[2];
                sp.L0[a] = (ctr - ext - G.bmin[a]) * inv_res - 0.5;
                sp.L1[a] = (ctr + ext - G.bmin[a]) * inv_res - 0.5;
            }
        }
    }
    if (wave == 1 && lane < 3 * SW_SAMPLES && P.enable_pos) {
        const int ps = lane / 3, a = lane - 3 * ps;
        const long long sx = sample_of_rank((long long)blockIdx.x + (long long)ps * n_blocks);
        if (sx < s_end) {
            const SampleId id = decode_sample(P, sx);
            if constexpr (FUSED) if (P.host_T) (void)stage_wait(P, id.b * P.N + id.i);      // (a miss is reported by wave 0)
            const PieceIn pin = piece_in(P, id.b, id.i);
            const double step = pin.T * P.inv_K;
            Basis B;
            poly_basis(id.j * step, B);
            const double *ca = pin.c + (size_t)a * pin.ld;
            double pa = 0.0;
#pragma unroll
            for (int k = 0; k < 6; k++) pa += ca[k] * B.b0[k];                   // component a of ct_beta(c, ld, b0): same operations
            const double bmn = a == 0 ? G.bmin[0] : (a == 1 ? G.bmin[1] : G.bmin[2]), bmx = a == 0 ? G.bmax[0] : (a == 1 ? G.bmax[1] : G.bmax[2]);
            const int dim = a == 0 ? G.X : (a == 1 ? G.Y : G.Z);
            // getPointsInAABB(pos, bd/2): corners projected into the map, floor-indexed (PCSmap_manager.h:150-155)
            const int lo = axis_index(clampd(pa - P.bd_half, bmn, bmx), bmn, G.res, dim);
            const int hi = axis_index(clampd(pa + P.bd_half, bmn, bmx), bmn, G.res, dim);
            s_pose[ps].lo[a] = lo;
            s_pose[ps].nn[a] = hi - lo + 1;             // NaN poses make every comparison false: the extent then says "no tile"
        }
    }
    } else {
        // a launch that is not one resident fused step (a batch, a long trajectory, a shard): the poses of ALL its samples were formed
        // by pose_kernel, one THREAD per sample - here the serial chain above would be issued once per workgroup for four active
        // lanes, a quarter of this kernel's instruction issue (profiles/r3a_c3_*) - and every wave only fetches its sample's record
        const SweepPose *gp = (const SweepPose *)P.poses;
        double *dst = (double *)&s_pose[wave];
        if (sidx < s_end) { const double *src = (const double *)(gp + (sidx - s_begin)); if (lane < (int)(sizeof(SweepPose) / 8)) dst[lane] = src[lane]; }
        else if (lane == 0) s_pose[wave].flags = 0;
    }
    __syncthreads();
    if (dbg && lane == 0) dbg[2] = clock64();

    const bool have_sample = sidx < s_end;
    SweepPose &sp = s_pose[wave];
    if constexpr (FUSED) {
        // ---- this wave's sample: window of rows and scan axis (wave-uniform; every lane computes the same)
        // Rows run along the world axis on which the shape's inflated bounding box is LONGEST (the bit-grid exists packed along
        // each axis): the window of rows spans the two short extents - a 3.2 m robot lying along x needs ~50 rows of ~20 voxels
        // instead of ~140 rows of ~6.
        // lane a < 3 works on axis a; the three results come back as scalars
        const int ax = min(lane, 2);
        int flags_w = have_sample ? __builtin_amdgcn_readfirstlane(sp.flags) : 0;
        const int nn_a = have_sample ? sp.nn[ax] : 0;
        const int nn[3] = {__builtin_amdgcn_readlane(nn_a, 0), __builtin_amdgcn_readlane(nn_a, 1), __builtin_amdgcn_readlane(nn_a, 2)};
        if ((flags_w & 1) && !(nn[0] > 0 && nn[1] > 0 && nn[2] > 0)) flags_w &= ~1;
        int w0a = 0, w1a = nn_a - 1;
        const bool windowed = (flags_w & 1) && P.shape.prune_rows;
        if (windowed) {
            const double l0 = sp.L0[ax] - sp.lo[ax], l1 = sp.L1[ax] - sp.lo[ax];
            if (l0 > 0.0) w0a = (int)fmin(floor(l0), (double)nn_a);          // > nn-1 -> empty window
            if (l1 < (double)(nn_a - 1)) w1a = (int)fmax(ceil(l1), -1.0);
        }
        const int w0[3] = {__builtin_amdgcn_readlane(w0a, 0), __builtin_amdgcn_readlane(w0a, 1), __builtin_amdgcn_readlane(w0a, 2)};
        const int w1[3] = {__builtin_amdgcn_readlane(w1a, 0), __builtin_amdgcn_readlane(w1a, 1), __builtin_amdgcn_readlane(w1a, 2)};
        int sa_w = 2;
        if (windowed) {
            const int e0 = w1[0] - w0[0], e1 = w1[1] - w0[1], e2 = w1[2] - w0[2];
            // (the mesh kind keeps z: measured 231 -> 217 us per step with 20 faces, 16.4 -> 14.4 ms with 4 992 - its hierarchy walks
            // are more coherent over voxels that follow each other along z)
            if (KIND != ISDF_SHAPE_MESH) sa_w = (e2 >= e0 && e2 >= e1) ? 2 : (e0 >= e1 ? 0 : 1);
        }
        // (u, v): the axes other than sa, ascending
        const int u0 = sa_w == 0 ? w0[1] : w0[0], u1 = sa_w == 0 ? w1[1] : w1[0];
        const int v0 = sa_w == 2 ? w0[1] : w0[2], v1 = sa_w == 2 ? w1[1] : w1[2];
        if (have_sample && lane == 0) {
            sp.rx0 = (flags_w & 1) ? u0 : 0; sp.ry0 = (flags_w & 1) ? v0 : 0;
            sp.rnx = (flags_w & 1) ? max(0, u1 - u0 + 1) : 0; sp.rny = (flags_w & 1) ? max(0, v1 - v0 + 1) : 0;
            sp.nzt = sa_w == 2 ? nn[2] : (sa_w == 0 ? nn[0] : nn[1]);
            sp.flags = flags_w;
            sp.sa = sa_w;
        }
        wave_lds_sync();
    }
    // What is wave-uniform about the sample comes (non-fused) from its record in global memory by SCALAR loads - no LDS round
    // trips, no v_readfirstlane; a fused step reads the record its first stage left in LDS.
    const mesh_cptr<SweepPose> rec = mesh_const_uni((const SweepPose *)P.poses + (have_sample && !FUSED ? sidx - s_begin : 0));
    int flags = 0, rnx = 0, rny = 0, nzt0 = 0;
    if (have_sample) {
        if constexpr (FUSED) {
            flags = __builtin_amdgcn_readfirstlane(sp.flags);
            rnx = __builtin_amdgcn_readfirstlane(sp.rnx); rny = __builtin_amdgcn_readfirstlane(sp.rny); nzt0 = __builtin_amdgcn_readfirstlane(sp.nzt);
        } else { flags = rec->flags; rnx = rec->rnx; rny = rec->rny; nzt0 = rec->nzt; }
    }
    const int n_rows = (flags & 1) ? rnx * rny : 0;
    const bool scanning = n_rows > 0;

    // scan state of this wave's sample
    const int n_zb = (nzt0 + 31) >> 5;   // 32-voxel blocks along z (1 for kernel_size <= 31)
    int near_head = 0, near_cnt = 0;             // ring of candidates awaiting the exact pass
    int pend_head = 0, pend = 0;                 // ring of compacted voxels awaiting the fp32 test
    int fzb = 0, ft = 0;                         // fetch cursor: z-block and first window row of the next round
    bool rows_left = scanning;
    unsigned hb = 0, pk = 0, zoff = 0;           // this lane's row: pending qualifier bits, packed (lx, ly), z offset
    unsigned nbits = 0, nmask = 0, npk = 0, nzoff = 0;   // the NEXT round's row, requested one round ahead
    int rounds_left = scanning ? n_zb * ((n_rows + 63) >> 6) : 0;
    bool primed = false;
    double tot = 0.0;                            // wave `s`, lane k < 8: component k of sample s's collision sums
    int tot_n = 0;                               // lanes 8 / 9: exact pairs / active pairs
    int mq_seq = 0;                              // mesh robots, queued exact pass: blocks of this sample queued so far
    // per-wave statistics of tools/timing.py (row rounds, passes, listed voxels, candidates, exact iterations): six scalar
    // registers held through the whole kernel and an increment per loop iteration - compiled in with -DISDF_SWEEP_STATS=1 only
#if ISDF_SWEEP_STATS
#define SW_STAT(x) x
#else
#define SW_STAT(x)
#endif
    SW_STAT(int n_rounds = 0;)
    SW_STAT(unsigned cnt_rounds = 0; unsigned cnt_passes = 0; unsigned cnt_filters = 0; unsigned cnt_listed = 0; unsigned cnt_cand = 0;)

    for (;;) {
        // ================= scan phase: produce -> fp32 filter -> candidate ring, until done or the ring is full
        bool finished = !rows_left && pend == 0;
        if (!finished) {
        // The scan-stage constants live only inside this block: they are rebuilt from the LDS pose record when the scan is
        // (re)entered, so that they do not hold scalar registers during the exact phase (SGPR spills cost VALU issue slots).
    int lox, loy, loz, nzt, rx0, ry0, sa;
    if constexpr (FUSED) {
        lox = __builtin_amdgcn_readfirstlane(sp.lo[0]); loy = __builtin_amdgcn_readfirstlane(sp.lo[1]); loz = __builtin_amdgcn_readfirstlane(sp.lo[2]);
        nzt = __builtin_amdgcn_readfirstlane(sp.nzt);
        rx0 = __builtin_amdgcn_readfirstlane(sp.rx0); ry0 = __builtin_amdgcn_readfirstlane(sp.ry0);
        sa = __builtin_amdgcn_readfirstlane(sp.sa);
    } else { lox = rec->lo[0]; loy = rec->lo[1]; loz = rec->lo[2]; nzt = rec->nzt; rx0 = rec->rx0; ry0 = rec->ry0; sa = rec->sa; }
    // scan axis sa (rows run along it), row coordinates (u, v) = the other two axes; everything below is wave-uniform
    const int lo_u = sa == 0 ? loy : lox, lo_v = sa == 2 ? loy : loz, lo_w = sa == 2 ? loz : (sa == 0 ? lox : loy);
    const int dim_v = sa == 2 ? G.Y : G.Z, n_ww = sa == 2 ? G.ZW : (sa == 0 ? G.XW : G.YW);
    const unsigned *const bits_a = sa == 2 ? G.bits : (sa == 0 ? G.bits_x : G.bits_y);
    const unsigned sh_u = sa == 0 ? 10u : 0u, sh_v = sa == 2 ? 10u : 20u, sh_w = 10u * (unsigned)sa;     // fields of a ring entry: x | y << 10 | z << 20
    // Box: the row masks already cut the rows to the box's own inflated extent, which IS the set the fp32 test would keep (but
    // for the rounded corners): measured 52 listed -> 48 kept, so the test is skipped for this kind
    constexpr bool DIRECT = (KIND == ISDF_SHAPE_BOX);
    const bool use_filter = !DIRECT && P.shape.filter_f32 != 0;
    // mesh robots: the filter is the distance lattice of isdf_set_shape (DevMesh::dl; wave-uniform: scalar loads)
    [[maybe_unused]] const float *ml_dl = nullptr;
    [[maybe_unused]] int ml_n0 = 0, ml_n1 = 0, ml_n2 = 0;
    [[maybe_unused]] float ml_m0 = 0.f, ml_m1 = 0.f, ml_m2 = 0.f, ml_inv = 0.f, ml_h = 0.f;
    if constexpr (KIND == ISDF_SHAPE_MESH) {
        if (use_filter) {
            const mesh_cptr<DevMesh> mm = mesh_const_uni(P.shape.mesh);
            ml_dl = mm->dl; ml_n0 = mm->dln[0]; ml_n1 = mm->dln[1]; ml_n2 = mm->dln[2];
            ml_m0 = mm->dl_min[0]; ml_m1 = mm->dl_min[1]; ml_m2 = mm->dl_min[2]; ml_inv = mm->dl_inv; ml_h = mm->dl_h;
        }
    }
    // the fp32 constants of the scan stage (scan_consts)
    const ScanLaunch SL = scan_launch_consts(P, KIND >= 0 ? KIND : P.shape.kind);
    const float resf = SL.resf, bdh_f = SL.bdh_f, thr_f = SL.thr_f;
    float Rf[9], pvx, pvy, pvz, Ru[3], Rv[3], Rw[3], pv_u, pv_v, pv_w, dzv[3], dzi[3], inv_rny, cone_alpha, cone_ia;
    bool cone_on;
    if constexpr (FUSED) {
        double pos_[3], R_[9];
        const int lo_[3] = {lox, loy, loz};
#pragma unroll
        for (int a = 0; a < 3; a++) pos_[a] = sp.pos[a];
#pragma unroll
        for (int k = 0; k < 9; k++) R_[k] = sp.R[k];
        SweepPose w;
        scan_consts(P, SL, pos_, R_, lo_, sa, rny, KIND == ISDF_SHAPE_ROUNDEDCONE && IDENT, w);
#pragma unroll
        for (int k = 0; k < 9; k++) Rf[k] = unif(w.Rf[k]);
        pvx = unif(w.pv[0]); pvy = unif(w.pv[1]); pvz = unif(w.pv[2]);
#pragma unroll
        for (int a = 0; a < 3; a++) { Ru[a] = w.Ru[a]; Rv[a] = w.Rv[a]; Rw[a] = w.Rw[a]; dzv[a] = w.dzv[a]; dzi[a] = w.dzi[a]; }
        pv_u = w.pvu; pv_v = w.pvv; pv_w = w.pvw; inv_rny = w.inv_rny;
        cone_alpha = w.cone_alpha; cone_ia = w.cone_ia; cone_on = w.cone_on != 0;
    } else {
#pragma unroll
        for (int k = 0; k < 9; k++) Rf[k] = rec->Rf[k];
        pvx = rec->pv[0]; pvy = rec->pv[1]; pvz = rec->pv[2];
#pragma unroll
        for (int a = 0; a < 3; a++) { Ru[a] = rec->Ru[a]; Rv[a] = rec->Rv[a]; Rw[a] = rec->Rw[a]; dzv[a] = rec->dzv[a]; dzi[a] = rec->dzi[a]; }
        pv_u = rec->pvu; pv_v = rec->pvv; pv_w = rec->pvw; inv_rny = rec->inv_rny;
        cone_alpha = rec->cone_alpha; cone_ia = rec->cone_ia; cone_on = rec->cone_on != 0;
        // (held in vector registers, like the values a fused step computes: the scalar file is the scarcer one)
#pragma unroll
        for (int a = 0; a < 3; a++) { asm volatile("" : "+v"(dzv[a])); asm volatile("" : "+v"(dzi[a])); }
    }
    const float cone_A = SL.cone_A, cone_B = SL.cone_B;

    // tile coordinates (u, v) of window row t, already in their fields of the ring entry
    auto row_xy = [&](int t) -> unsigned {
        // t / rny through fp32 with a one-step correction: exact for t < 2^24, and a window has at most 1024 x 1024 rows (the
        // 10-bit fields of a ring entry)
        unsigned wx = (unsigned)((float)t * inv_rny);
        int rem = t - (int)wx * rny;
        if (rem < 0) { wx--; rem += rny; } else if (rem >= rny) { wx++; rem -= rny; }
        const unsigned wy = (unsigned)rem;
        return ((wx + (unsigned)rx0) << sh_u) | ((wy + (unsigned)ry0) << sh_v);
    };
    // the <= 32 qualifier bits of block zb of tile row pk: 1-2 dwords of the bit-grid packed along the scan axis
    auto row_bits = [&](unsigned pk, int zb) -> unsigned {
        const int z0 = lo_w + (zb << 5);
        const int nz = min(32, nzt - (zb << 5));
        const size_t rowbase = ((size_t)(lo_u + (int)((pk >> sh_u) & 1023u)) * dim_v + (lo_v + (int)((pk >> sh_v) & 1023u))) * n_ww;
        const int w0 = z0 >> 5, sh = z0 & 31;
        const unsigned b0 = bits_a[rowbase + w0];
        const unsigned b1 = bits_a[rowbase + min(w0 + 1, n_ww - 1)];
        unsigned bits = sh ? ((b0 >> sh) | (b1 << (32 - sh))) : b0;     // when w0+1 == n_ww the b1 bits are masked off below
        if (nz < 32) bits &= (1u << nz) - 1u;
        return bits;
    };
    // Row pruning: in the body frame the voxels of one row lie on the line o + lw*dz.  Only the part of the line
    // inside (shape bounding box inflated by safety_hor) ∩ (the |p_rel| <= bd/2 cube) can carry a penalty
    // (isdf_shape.bbox_* guarantee), so the rest of the row never enters the list.  Conservative by construction.
    const bool prune = P.shape.prune_rows != 0;
    float blo[3], bhi[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        blo[a] = fmaxf(P.shape.bbox_lo[a] - thr_f, -bdh_f);
        bhi[a] = fminf(P.shape.bbox_hi[a] + thr_f, bdh_f);
    }
    // RoundedCone: inside the cone that is tangent to both inflated spheres a row's part is one more interval (a quadratic in the
    // row coordinate: scan_launch_consts / scan_consts).  Conservative like the box test (the band carries the filter's 2 mm; a
    // negative discriminant of a row that really touches would need rounding errors a thousand times larger).
    auto row_mask = [&](unsigned pk, int zb) -> unsigned {
        const float mx = fmaf((float)((pk >> sh_u) & 1023u), resf, -pv_u), my = fmaf((float)((pk >> sh_v) & 1023u), resf, -pv_v);
        float tmin = -1.0e30f, tmax = 1.0e30f;
        bool empty = false;
        float ob[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float o = Ru[a] * mx + Rv[a] * my - Rw[a] * pv_w;
            ob[a] = o;
            if (fabsf(dzv[a]) > 1.0e-12f) {
                const float t1 = (blo[a] - o) * dzi[a], t2 = (bhi[a] - o) * dzi[a];
                tmin = fmaxf(tmin, fminf(t1, t2));      // fmaxf/fminf drop NaNs: never tightens on garbage
                tmax = fminf(tmax, fmaxf(t1, t2));
            } else {
                empty = empty | (o < blo[a] - 1.0e-3f) | (o > bhi[a] + 1.0e-3f);
            }
        }
        if constexpr (KIND == ISDF_SHAPE_ROUNDEDCONE && IDENT) {
            if (cone_on) {
                const float e = cone_A - cone_B * ob[2];
                const float beta = ob[0] * dzv[0] + ob[1] * dzv[1] + cone_B * dzv[2] * e;
                const float gamma = ob[0] * ob[0] + ob[1] * ob[1] - e * e;
                const float disc = beta * beta - cone_alpha * gamma;
                if (disc < 0.0f) empty = true;           // (a NaN fails the comparison: the row stays)
                else {
                    const float sq = sqrtf(disc);
                    tmin = fmaxf(tmin, (-beta - sq) * cone_ia);
                    tmax = fminf(tmax, (-beta + sq) * cone_ia);
                }
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

        while (!finished) {
            if (pend >= 64 || (!rows_left && pend > 0)) {
                if (near_cnt > SW_NEAR_CAP - 64) break;            // ring full: the workgroup drains it first
                // ---- (3) conservative fp32 test of the oldest min(pend, 64) listed voxels (branch-free body)
                const int n = min(pend, 64);
                const unsigned ent = s_list[wave][(pend_head + min(lane, n - 1)) & (SW_LIST_CAP - 1)];
                bool cand = lane < n;
                if (use_filter) {
                    const float flx = (float)(ent & 1023u), fly = (float)((ent >> 10) & 1023u), flz = (float)(ent >> 20);
                    const float mx = fmaf(flx, resf, -pvx), my = fmaf(fly, resf, -pvy), mz = fmaf(flz, resf, -pvz);
                    const f3 pr = mkv<float>(Rf[0] * mx + Rf[3] * my + Rf[6] * mz, Rf[1] * mx + Rf[4] * my + Rf[7] * mz,
                                             Rf[2] * mx + Rf[5] * my + Rf[8] * mz);
                    const bool incube = fmaxf(fabsf(pr.x), fmaxf(fabsf(pr.y), fabsf(pr.z))) <= bdh_f;
                    float sdf;
                    if constexpr (KIND == ISDF_SHAPE_MESH) {
                        // nearest lattice node: outside the mesh and farther from it than the node spacing -> the voxel is at least
                        // (that distance - spacing) away, on the outside (thr_f carries the 5 % by which the approximate winding
                        // number may scale the distance, like the row pruning); anything else, or a NaN, stays a candidate
                        const float fx = (pr.x - ml_m0) * ml_inv, fy = (pr.y - ml_m1) * ml_inv, fz = (pr.z - ml_m2) * ml_inv;
                        const bool in = fx > -0.5f && fx < (float)ml_n0 - 0.5f && fy > -0.5f && fy < (float)ml_n1 - 0.5f && fz > -0.5f && fz < (float)ml_n2 - 0.5f;
                        const int ix = in ? (int)rintf(fx) : 0, iy = in ? (int)rintf(fy) : 0, iz = in ? (int)rintf(fz) : 0;
                        const float v = in ? ml_dl[((size_t)ix * ml_n1 + iy) * ml_n2 + iz] : -1.0f;
                        sdf = v > 0.0f ? v - ml_h : -1.0e30f;
                    } else sdf = analytic_sdf<KIND, float, IDENT>(P.shape.f, P.shape.kind, pr);
                    cand = cand & incube & !(sdf >= thr_f);   // reject only what provably has zero penalty; NaN => candidate
                }
                pend_head = (pend_head + n) & (SW_LIST_CAP - 1);
                pend -= n;
                SW_STAT(cnt_filters++;)
                const unsigned long long nm = __ballot(cand);
                if (cand) s_near[wave][(near_head + near_cnt + __popcll(nm & ((1ull << lane) - 1ull))) & (SW_NEAR_CAP - 1)] = ent;
                near_cnt += __popcll(nm);
                SW_STAT(cnt_cand += (unsigned)__popcll(nm);)
            } else {
            // (a round that has just been taken is compacted in the same turn of this loop: the same sequence of operations, one
            // turn of the state machine - a back edge, its conditions, the moves of the loop-carried scalars - per round less)
            if (!__ballot(hb != 0u)) {
                // ---- (2a) take the round requested earlier, request the one after it
                if (primed && rounds_left == 0) {
                    rows_left = false;
                    if (dbg && lane == 0) dbg[4] = clock64();
                } else {
                    if (primed) {
                        SW_STAT(cnt_rounds++;)
                        rounds_left--;
                        hb = nbits & nmask;          // waits for the loads requested one round ago
                        pk = npk; zoff = nzoff;
                    }
                    primed = true;
                    // request round (fzb, ft): its loads stay in flight while the current round is processed
                    nmask = 0u; nbits = 0u;
                    if (fzb < n_zb) {
                        const int t = ft + lane;
                        if (t < n_rows) {
                            npk = row_xy(t);
                            nmask = prune ? row_mask(npk, fzb) : 0xFFFFFFFFu;
                            if (nmask) nbits = row_bits(npk, fzb);
                        }
                        nzoff = (unsigned)(fzb << 5);
                        ft += 64;
                        if (ft >= n_rows) { ft = 0; fzb++; }
                    }
                }
            }
            if (__ballot(hb != 0u)) {
                // ---- (2b) one pass: every lane moves its lowest <= SW_PASS set bits into the ring
                if constexpr (DIRECT) {
                    // no fp32 test for this kind: the listed voxels ARE the candidates, written straight into the candidate ring
                    const int room = SW_NEAR_CAP - near_cnt;
                    if (room < 64) break;                              // ring full: the workgroup drains it first
                    // the whole round at once when it fits (nearly always: one pass per round, no per-lane limit to work out);
                    // otherwise the lowest bits of every lane that are guaranteed to fit, the rest in later passes
                    unsigned take = hb;
                    int total;
                    int wpos = wave_excl_scan(__popc(take), lane, total);
                    if (total > room) {
                        const int lim = min(SW_PASS, room >> 6);
                        unsigned t = hb;
                        for (int k = 0; k < lim; k++) t &= t - 1u;
                        take = hb ^ t;
                        wpos = wave_excl_scan(__popc(take), lane, total);
                    }
                    hb ^= take;
                    wpos += near_head + near_cnt;
                    while (take) {
                        const unsigned z = (unsigned)__ffs(take) - 1u;
                        take &= take - 1u;
                        s_near[wave][(wpos++) & (SW_NEAR_CAP - 1)] = pk | ((zoff + z) << sh_w);
                    }
                    near_cnt += total;
                    SW_STAT(cnt_passes++; cnt_listed += (unsigned)total; cnt_cand += (unsigned)total;)
                } else {
                // (like the direct form above: the whole round when it fits the list, else <= SW_PASS voxels per lane)
                unsigned take = hb;
                int total;
                int wpos = wave_excl_scan(__popc(take), lane, total);
                if (total > SW_LIST_CAP - pend) {
                    unsigned t = hb;
#pragma unroll
                    for (int k = 0; k < SW_PASS; k++) t &= t - 1u;
                    take = hb ^ t;
                    wpos = wave_excl_scan(__popc(take), lane, total);
                }
                hb ^= take;
                wpos += pend_head + pend;
                while (take) {
                    const unsigned z = (unsigned)__ffs(take) - 1u;
                    take &= take - 1u;
                    s_list[wave][(wpos++) & (SW_LIST_CAP - 1)] = pk | ((zoff + z) << sh_w);
                }
                pend += total;
                SW_STAT(cnt_passes++; cnt_listed += (unsigned)total;)
                }
                wave_lds_sync();
            }
            // (the last round has been taken and nothing of it is left: what a further turn would find out)
            if (primed && rounds_left == 0 && !__ballot(hb != 0u)) rows_left = false;
            }
            finished = !rows_left && pend == 0;
        }
        }
        if (lane == 0) { s_plan[wave][0] = near_cnt; s_plan[wave][1] = near_head; s_plan[wave][2] = finished ? 1 : 0; }
        __syncthreads();
        if constexpr (KIND == ISDF_SHAPE_MESH) {
            if (P.mq_items) {
                // Mesh robot, queued exact pass: this wavefront's candidates leave as BLOCKS of 64 consecutive voxels of its sample's
                // stream (the last one partial, when the sample is finished) - independent work items of mesh_exact_kernel, whose
                // results mesh_sum_kernel adds per sample in block order.  A sample's blocks depend on its own candidates only.
                const bool all_fin = (s_plan[0][2] & s_plan[1][2] & s_plan[2][2] & s_plan[3][2]) != 0;
                const int nb = near_cnt / MQ_BLOCK + ((finished && (near_cnt % MQ_BLOCK)) ? 1 : 0);
                if (nb > 0 && have_sample) {
                    const int sl = (int)(sidx - s_begin);
                    unsigned base = 0u;
                    if (lane == 0) base = atomicAdd(&P.mq_count[0], (unsigned)nb);
                    base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
                    const bool fits = base + (unsigned)nb <= P.mq_cap && mq_seq + nb <= P.mq_kmax;
                    if (!fits && lane == 0) atomicMax(&P.stats[4], 1ull);                  // (cannot happen with the host's sizing; the result is flagged invalid)
                    const int used = min(near_cnt, nb * MQ_BLOCK);
                    if (fits) {
                        // the entries of the nb blocks are contiguous in the queue: 64 lanes copy them, lane k < nb writes block k's header
                        for (int e0 = 0; e0 < used; e0 += 64) {
                            const int e = e0 + lane;
                            if (e < used) P.mq_entries[(size_t)base * MQ_BLOCK + e] = s_near[wave][(near_head + e) & (SW_NEAR_CAP - 1)];
                        }
                        for (int bk = lane; bk < nb; bk += 64) {
                            P.mq_items[base + bk] = make_int2(sl, min(MQ_BLOCK, used - bk * MQ_BLOCK));
                            P.mq_sample_items[(size_t)sl * P.mq_kmax + mq_seq + bk] = (int)(base + bk);
                        }
                        mq_seq += nb;
                    }
                    near_head = (near_head + used) & (SW_NEAR_CAP - 1);
                    near_cnt -= used;
                }
                if (all_fin) break;
                __syncthreads();        // s_plan is rewritten by the next round
                continue;
            }
        }

        // ================= exact phase: the workgroup's candidates in blocks of up to 64 of one sample, dealt round-robin to its
        // waves.  A sample's FULL blocks are one wavefront iteration each.  Its last, partial block is classed by size - whole
        // (> 32 candidates), half (<= 32) or quarter (<= 16) - and blocks of the same class share an iteration: two halves on lanes
        // 0-31 / 32-63, four quarters on the four rows of 16.  A block's sums are reduced by the SAME butterfly whatever it shares
        // the wavefront with (wave_sum8_halves / _rows only leave out the levels that add the zeros of a part-filled wavefront),
        // so a sample's result does not depend on its neighbours in the workgroup: 17-19 % fewer iterations at C2 / C3
        // (tools/chunk_hist.py) with every bitwise identity (batch == single, shards, cull on == off) kept.
        // The plan is identical in every wave.  It is worked out on the lanes (lane l looks at sample l & 3: one LDS read, a few
        // compares whose wave-wide ballots ARE the class masks, two quad exchanges) and kept packed in scalars:
        // f_pk = full blocks per sample (4 bits each), cls_m = samples whose partial block is whole (bits 0-3), half (4-7), quarter (8-11).
        int f_pk, n_full, cls_m, all_done;
        {
            const int4 pl = *reinterpret_cast<const int4 *>(&s_plan[lane & 3][0]);       // candidates, ring head, finished
            const int r = pl.x & 63;
            const int m_w = (int)__ballot(r > 32) & 15, m_h = (int)__ballot(r > 16 && r <= 32) & 15, m_q = (int)__ballot(r > 0 && r <= 16) & 15;
            cls_m = m_w | (m_h << 4) | (m_q << 8);
            all_done = ((int)__ballot(pl.z != 0) & 15) == 15;
            int f = pl.x >> 6, fp = f << (4 * (lane & 3));
            f += __builtin_amdgcn_mov_dpp(f, 0xB1, 0xf, 0xf, false);   fp |= __builtin_amdgcn_mov_dpp(fp, 0xB1, 0xf, 0xf, false);
            f += __builtin_amdgcn_mov_dpp(f, 0x4E, 0xf, 0xf, false);   fp |= __builtin_amdgcn_mov_dpp(fp, 0x4E, 0xf, 0xf, false);
            n_full = __builtin_amdgcn_readfirstlane(f); f_pk = __builtin_amdgcn_readfirstlane(fp);
        }
        const int n_w = __builtin_popcount(cls_m & 15), n_hb = (__builtin_popcount((cls_m >> 4) & 15) + 1) >> 1;
        const int n_items = n_full + n_w + n_hb + ((cls_m >> 8) != 0);
        const int lane_q3 = 3 * (lane >> 4);
        for (int e = wave; e < n_items; e += SW_SAMPLES) {
            // this item: owner of each row of 16 lanes (3 bits each, 7 = nobody), lanes per block - 1, block index (-1: the partial one).
            // Items in order: full blocks by sample; whole partial blocks; pairs of halves (in sample order); the quarters
            // (all in one item, sample s on row s).
            int ob, lmask = 63, cc_u = -1;
            {
                int t = e;
                if (t < n_full) {
                    int cs = 0;
#pragma unroll
                    for (int w = 0; w < SW_SAMPLES; w++) {
                        const int f_w = (f_pk >> (4 * w)) & 15;
                        const bool hit = t >= 0 && t < f_w;
                        cs = hit ? w : cs; cc_u = hit ? t : cc_u;
                        t -= f_w;
                    }
                    ob = cs * 0x249;
                } else if ((t -= n_full) < n_w) {
                    int m = cls_m & 15;
                    if (t >= 1) m &= m - 1;
                    if (t >= 2) m &= m - 1;
                    if (t >= 3) m &= m - 1;
                    ob = __builtin_ctz(m) * 0x249;
                } else if ((t -= n_w) < n_hb) {
                    lmask = 31;
                    int m = (cls_m >> 4) & 15;
                    if (t >= 1) { m &= m - 1; m &= m - 1; }
                    const int a = __builtin_ctz(m);
                    m &= m - 1;
                    const int b = m ? __builtin_ctz(m) : 7;
                    ob = a * 0x9 | b * (0x9 << 6);
                } else {
                    lmask = 15;
                    const int inv = ~(cls_m >> 8) & 15;
                    ob = 0x688 | 7 * ((inv & 1) | ((inv & 2) << 2) | ((inv & 4) << 4) | ((inv & 8) << 6));
                }
            }
            // ---- (4) exact fp64 pass over one wavefront of candidates: grad_cost_p (:766-824)
            int ls = (ob >> lane_q3) & 7;                    // this lane's sample (of the workgroup's four)
            const bool seg_valid = ls != 7;
            ls &= 3;
            const int l = lane & lmask;                      // lane inside its block
            const int2 pl = *reinterpret_cast<const int2 *>(&s_plan[ls][0]);      // candidates, ring head: one LDS round trip
            const int ccnt = pl.x, chead = pl.y;
            const int cc = cc_u >= 0 ? cc_u : (ccnt >> 6);
            const int n = min(64, ccnt - (cc << 6));
            const unsigned ent = s_near[ls][(chead + (cc << 6) + min(l, n - 1)) & (SW_NEAR_CAP - 1)];
            double v[ACC_STRIDE] = {0, 0, 0, 0, 0, 0, 0, 0};
            bool is_pair = false, is_grad = false;
            if constexpr (KIND == ISDF_SHAPE_MESH) {
                // Mesh robot: the hierarchy is walked by QUADS (dev_mesh.hpp: 4 lanes per voxel, one per child of the node, the walk
                // wave-uniform, frames in LDS) - 16 voxels of the chunk at a time; consecutive ring entries are neighbouring voxels of
                // a row, so the 16 open nearly the same nodes.  The per-lane walk of round 3 (every lane chasing its own 368-byte
                // records with an 800-byte private stack, node and triangle branches serialised) ran at 16 % of the VALU issue rate.
                bool want = false;
                d3 prel = mk3(0, 0, 0), pm = prel;
                int csa = ls;
                asm volatile("" : "+v"(csa));
                if (seg_valid && l < n && !(ISDF_SWEEP_STATS && (P.dbg_flags & 1))) {
                    const SweepPose &qa = s_pose[csa];
                    const int gx = qa.lo[0] + (int)(ent & 1023u), gy = qa.lo[1] + (int)((ent >> 10) & 1023u), gz = qa.lo[2] + (int)(ent >> 20);
                    const d3 ctr = mk3((gx + 0.5) * G.res + G.bmin[0], (gy + 0.5) * G.res + G.bmin[1], (gz + 0.5) * G.res + G.bmin[2]);
                    const d3 pos = mk3(qa.pos[0], qa.pos[1], qa.pos[2]);
                    m3 R;
#pragma unroll
                    for (int k = 0; k < 9; k++) R.a[k] = qa.R[k];
                    pm = ctr - pos;
                    prel = mat_tmul(R, pm);
                    if (!(fmax(fmax(fabs(prel.x), fabs(prel.y)), fabs(prel.z)) > P.bd_half)) { is_pair = true; want = true; }
                }
                double sdf = 0.0;
                d3 g = mk3(0, 0, 0);
                bool below = false;
                const unsigned long long wm = __ballot(want);
                for (int r = 0; r < 4; r++) {
                    if (!((wm >> (16 * r)) & 0xFFFFull)) continue;                       // (wave-uniform)
                    const int src = 16 * r + (lane >> 2);                                // the voxel this quad works on
                    const d3 qp = mk3(__shfl(prel.x, src, 64), __shfl(prel.y, src, 64), __shfl(prel.z, src, 64));
                    const bool on = (wm >> src) & 1ull;
                    double s2 = 0.0;
                    d3 g2 = mk3(0, 0, 0);
                    const bool b2 = mesh_sdf_grad_below_quad(P.shape.mesh, MeshFrames{&s_mframes[wave][0], MESH_Q_LEVELS}, qp, on, P.safety_hor, s2, g2);
                    const int from = 4 * (lane & 15);                                    // back to the voxel's own lane
                    const double s3 = __shfl(s2, from, 64), gx3 = __shfl(g2.x, from, 64), gy3 = __shfl(g2.y, from, 64), gz3 = __shfl(g2.z, from, 64);
                    const int b3 = __shfl(b2 ? 1 : 0, from, 64);
                    if ((lane >> 4) == r) { sdf = s3; g = mk3(gx3, gy3, gz3); below = b3 != 0; }
                }
                if (want && below) {
                    double f = 0.0, df = 0.0;
                    smoothed_l1_inv(P.safety_hor - sdf, P.mu, P.inv_mu, f, df);
                    if (f > 0.0) {
                        is_grad = true;
                        v[0] = f;
                        int csb = ls;
                        asm volatile("" : "+v"(csb));
                        const SweepPose &qb = s_pose[csb];
                        m3 R;
#pragma unroll
                        for (int k = 0; k < 9; k++) R.a[k] = qb.R[k];
                        d4 quat; quat.w = qb.quat[0]; quat.x = qb.quat[1]; quat.y = qb.quat[2]; quat.z = qb.quat[3];
                        const d3 wg = mat_mul(R, g);
                        v[1] = df * wg.x; v[2] = df * wg.y; v[3] = df * wg.z;
                        const d4 sq = quat_step_grad(quat, g, pm);
                        v[4] = -df * sq.w; v[5] = -df * sq.x; v[6] = -df * sq.y; v[7] = -df * sq.z;
                    }
                }
            } else
            if (seg_valid && l < n && !(ISDF_SWEEP_STATS && (P.dbg_flags & 1))) {
                // The pose (16 doubles) is READ FROM LDS where it is used - once here, once more for the force terms below -
                // instead of living in 32 scalar registers for the whole loop: the loop's scalar state no longer fits the SGPR
                // file with it, and every spilled scalar costs a v_readlane, i.e. a VALU issue slot of this issue-bound kernel
                // (broadcast ds_reads go down the LDS pipe).  The index is laundered so that the two reads stay two reads.
                int csa = ls;
                asm volatile("" : "+v"(csa));
                const SweepPose &qa = s_pose[csa];
                const int gx = qa.lo[0] + (int)(ent & 1023u), gy = qa.lo[1] + (int)((ent >> 10) & 1023u), gz = qa.lo[2] + (int)(ent >> 20);
                const d3 ctr = mk3((gx + 0.5) * G.res + G.bmin[0], (gy + 0.5) * G.res + G.bmin[1], (gz + 0.5) * G.res + G.bmin[2]);
                const d3 pos = mk3(qa.pos[0], qa.pos[1], qa.pos[2]);
                m3 R;
#pragma unroll
                for (int k = 0; k < 9; k++) R.a[k] = qa.R[k];
                const d3 pm = ctr - pos;
                const d3 prel = mat_tmul(R, pm);
                // the cube cull (:800-803) as ONE test: three short-circuit comparisons compile to three nested masked regions, each
                // re-creating the eight result registers (fmax drops a NaN like the comparisons it replaces treat it: inside)
                if (!(fmax(fmax(fabs(prel.x), fabs(prel.y)), fabs(prel.z)) > P.bd_half)) {
                    is_pair = true;
                    double f = 0.0, df = 0.0;
                    d3 g = mk3(0, 0, 0);
                    smoothed_l1_inv(P.safety_hor - shape_sdf<KIND, IDENT>(P.shape, prel), P.mu, P.inv_mu, f, df);
                    if (f > 0.0) {
                        if constexpr (KIND != ISDF_SHAPE_MESH) g = shape_grad<KIND, IDENT>(P.shape, prel);
                        is_grad = true;
                        v[0] = f;
                        int csb = ls;
                        asm volatile("" : "+v"(csb));
                        const SweepPose &qb = s_pose[csb];
                        m3 R;
#pragma unroll
                        for (int k = 0; k < 9; k++) R.a[k] = qb.R[k];
                        d4 quat; quat.w = qb.quat[0]; quat.x = qb.quat[1]; quat.y = qb.quat[2]; quat.z = qb.quat[3];
                        const d3 wg = mat_mul(R, g);
                        v[1] = df * wg.x; v[2] = df * wg.y; v[3] = df * wg.z;
                        const d4 sq = quat_step_grad(quat, g, pm);
                        v[4] = -df * sq.w; v[5] = -df * sq.x; v[6] = -df * sq.y; v[7] = -df * sq.z;
                    }
                }
            }
            // sums per block: component k of a block ends up in v[0] of its lanes with `comp == k`
            // (the lane index is laundered: otherwise every lane-bit predicate of the three butterflies is hoisted out of all loops
            // and sits in a scalar register pair for the whole kernel - spilled scalars cost VALU slots in the scan loop)
            int ln = lane;
            asm volatile("" : "+v"(ln));
            int comp; bool writer;
            if (lmask == 63)      { wave_sum8(v, ln);        writer = (ln & 7) == 0; comp = ((ln >> 3) & 1) | (((ln >> 4) & 1) << 1) | ((ln >> 5) << 2); }
            else if (lmask == 31) { wave_sum8_halves(v, ln); writer = (ln & 3) == 0; comp = (ln >> 2) & 7; }
            else                  { wave_sum8_rows(v, ln);   writer = (ln & 1) == 0; comp = (ln >> 1) & 7; }
            const unsigned long long blk = lmask == 63 ? ~0ull : ((1ull << (lmask + 1)) - 1ull);
            const int first = ln & ~lmask;                   // first lane of this lane's block
            const int np = __popcll((__ballot(is_pair) >> first) & blk), ng = __popcll((__ballot(is_grad) >> first) & blk);
            if (seg_valid) {
                if (writer) s_res[ls][cc][comp] = v[0];
                if (l == 0) { s_resn[ls][cc][0] = np; s_resn[ls][cc][1] = ng; }
            }
            SW_STAT(n_rounds++;)
        }
        __syncthreads();
        // chunk results of this wave's sample, in chunk order
        {
            const int my_nch = (near_cnt + 63) >> 6;
            for (int cidx = 0; cidx < my_nch; cidx++) {
                if (lane < ACC_STRIDE) tot += s_res[wave][cidx][lane];
                else if (lane < ACC_STRIDE + 2) tot_n += s_resn[wave][cidx][lane - ACC_STRIDE];
            }
        }
        near_head = (near_head + near_cnt) & (SW_NEAR_CAP - 1);
        near_cnt = 0;
        if (all_done) break;
        __syncthreads();        // s_plan / s_res are rewritten by the next round
    }

    // ---- (5) results of this wave's sample
    if constexpr (KIND == ISDF_SHAPE_MESH) {
        if (P.mq_items) {            // queued exact pass: mesh_sum_kernel writes the sample's sums; here only how many blocks it has
            if (have_sample && lane == 0) P.mq_sample_n[sidx - s_begin] = mq_seq | ((flags & 2) ? (1 << 30) : 0);
            return;
        }
    }
    if (have_sample) {
        const long long sl = sidx - s_begin;
        if (lane < ACC_STRIDE) {
            if constexpr (FUSED) slot_publish(&P.acc[(size_t)sl * ACC_STRIDE + lane], tot);    // the tail may already be polling
            else P.acc[(size_t)sl * ACC_STRIDE + lane] = tot;                                  // read by the next launch
        }
        else if (lane == ACC_STRIDE) {
            P.sample_info[2 * sl] = tot_n;
            if (P.plan_cls_out && !(P.dbg_flags & 4)) P.plan_cls_out[sl] = (unsigned char)min((tot_n + 31) >> 5, PLAN_CLASSES - 1);     // work class: exact pairs / 32
        }
        else if (lane == ACC_STRIDE + 1) P.sample_info[2 * sl + 1] = tot_n | ((flags & 2) ? (int)0x80000000 : 0);
    }
    if (dbg && lane == 0) {
        dbg[5] = clock64(); dbg[6] = wall_clock64();
        SW_STAT(dbg[7] = (unsigned long long)n_rounds;)
        SW_STAT(dbg[3] = (unsigned long long)cnt_rounds | ((unsigned long long)cnt_passes << 8) | ((unsigned long long)cnt_filters << 16) | ((unsigned long long)cnt_listed << 24) | ((unsigned long long)cnt_cand << 44);)
    }
}

// Mesh robots, queued exact pass (round 4).  With the exact pass inside the sweep kernel a C2 step was ONE resident launch of 650
// workgroups, each walking the hierarchy for its own four samples' voxels: 40 % of the VALU issue rate, 1.4 wavefronts per SIMD on
// average.  Here a work item is ONE quad round - MQ_BLOCK = 16 consecutive voxels of a sample's candidate stream, 4 lanes per voxel -
// and the wavefronts of the launch CLAIM items one at a time (an atomic counter): the device stays full until the queue is empty.
// grad_cost_p (:766-824) per voxel exactly as the in-kernel pass forms it; a block's 8 sums by the same butterfly (the empty lanes add zeros).
__global__ __launch_bounds__(64 * SW_SAMPLES, 5) void mesh_exact_kernel(const SweepParams P) {
    __shared__ __attribute__((aligned(8))) unsigned s_mframes[SW_SAMPLES][mesh_frames_words(MESH_Q_LEVELS)];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const DevGrid &G = P.grid;
    const unsigned n_items = min((unsigned)__builtin_amdgcn_readfirstlane((int)P.mq_count[0]), P.mq_cap);
    const int qv = lane >> 2;                                                    // this quad's voxel of the item
    // (a wavefront's first item is its own number: no counter traffic for the launch's first round - read-modify-writes of ONE
    // address take their turns in the L2)
    const unsigned n_waves = (unsigned)__builtin_amdgcn_readfirstlane((int)(gridDim.x * SW_SAMPLES));
    for (unsigned round = 0;; round++) {
        unsigned it = (unsigned)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * SW_SAMPLES + wave));
        if (round > 0) {           // (two static rounds: 20 faces 146 -> 129 us, 720 faces 1.55 -> 1.58 ms - the queue's balance is worth more there)
            if (lane == 0) it = atomicAdd(&P.mq_count[1], 1u);
            it = n_waves + (unsigned)__builtin_amdgcn_readfirstlane((int)it);
        }
        if (it >= n_items) break;
        const int2 item = P.mq_items[it];
        const int sl = __builtin_amdgcn_readfirstlane(item.x), n = __builtin_amdgcn_readfirstlane(item.y);
        if (sl < 0 || sl >= (P.piece_end - P.piece_begin) * (P.K + 1) || n < 1 || n > MQ_BLOCK) {      // (a header nobody wrote: flagged, not followed)
            if (lane == 0) atomicMax(&P.stats[4], 1ull);
            continue;
        }
        const mesh_cptr<SweepPose> q = mesh_const_uni((const SweepPose *)P.poses + sl);        // the sample's pose record: scalar loads
        double v[ACC_STRIDE] = {0, 0, 0, 0, 0, 0, 0, 0};
        bool want = false;
        d3 prel = mk3(0, 0, 0), pm = prel;
        m3 R;
#pragma unroll
        for (int k = 0; k < 9; k++) R.a[k] = q->R[k];
        if (qv < n) {                                                            // (the four lanes of a quad form the same numbers)
            const unsigned ent = P.mq_entries[(size_t)it * MQ_BLOCK + qv];
            const int gx = q->lo[0] + (int)(ent & 1023u), gy = q->lo[1] + (int)((ent >> 10) & 1023u), gz = q->lo[2] + (int)(ent >> 20);
            const d3 ctr = mk3((gx + 0.5) * G.res + G.bmin[0], (gy + 0.5) * G.res + G.bmin[1], (gz + 0.5) * G.res + G.bmin[2]);
            const d3 pos = mk3(q->pos[0], q->pos[1], q->pos[2]);
            pm = ctr - pos;
            prel = mat_tmul(R, pm);
            want = !(fmax(fmax(fabs(prel.x), fabs(prel.y)), fabs(prel.z)) > P.bd_half);       // the cube cull (:800-803)
        }
        double sdf = 0.0;
        d3 g = mk3(0, 0, 0);
        const bool below = mesh_sdf_grad_below_quad(P.shape.mesh, MeshFrames{&s_mframes[wave][0], MESH_Q_LEVELS}, prel, want, P.safety_hor, sdf, g);
        const bool owner = (lane & 3) == 0;                                      // one lane per voxel carries its terms into the sums
        bool is_grad = false;
        if (owner && want && below) {
            double f = 0.0, df = 0.0;
            smoothed_l1_inv(P.safety_hor - sdf, P.mu, P.inv_mu, f, df);
            if (f > 0.0) {
                is_grad = true;
                v[0] = f;
                d4 quat; quat.w = q->quat[0]; quat.x = q->quat[1]; quat.y = q->quat[2]; quat.z = q->quat[3];
                const d3 wg = mat_mul(R, g);
                v[1] = df * wg.x; v[2] = df * wg.y; v[3] = df * wg.z;
                const d4 sq = quat_step_grad(quat, g, pm);
                v[4] = -df * sq.w; v[5] = -df * sq.x; v[6] = -df * sq.y; v[7] = -df * sq.z;
            }
        }
        wave_sum8(v, lane);                                                      // component comp in v[0] of the lanes with (lane & 7) == 0
        const int comp = ((lane >> 3) & 1) | (((lane >> 4) & 1) << 1) | ((lane >> 5) << 2);
        const int np = __popcll(__ballot(owner && want)), ng = __popcll(__ballot(is_grad));
        double *o = P.mq_res + (size_t)it * 10;
        if ((lane & 7) == 0) o[comp] = v[0];
        if (lane == 1) { o[8] = (double)np; o[9] = (double)ng; }
    }
}
// ... and the sample's sums: its blocks' results added in block order (one thread per sample)
__global__ __launch_bounds__(256) void mesh_sum_kernel(const SweepParams P) {
    const long long n_samples = (long long)(P.piece_end - P.piece_begin) * (P.K + 1);
    const long long sl = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (sl >= n_samples) return;
    const int w = P.mq_sample_n[sl], nb = w & 0x3fffffff;
    double tot[ACC_STRIDE] = {0, 0, 0, 0, 0, 0, 0, 0};
    int np = 0, ng = 0;
    for (int bk = 0; bk < nb; bk++) {
        const double *o = P.mq_res + (size_t)P.mq_sample_items[(size_t)sl * P.mq_kmax + bk] * 10;
#pragma unroll
        for (int k = 0; k < ACC_STRIDE; k++) tot[k] += o[k];
        np += (int)o[8]; ng += (int)o[9];
    }
#pragma unroll
    for (int k = 0; k < ACC_STRIDE; k++) P.acc[(size_t)sl * ACC_STRIDE + k] = tot[k];
    P.sample_info[2 * sl] = np;
    P.sample_info[2 * sl + 1] = ng | ((w & (1 << 30)) ? (int)0x80000000 : 0);
}

// on-demand statistics (isdf_get_stats / host API): culled poses, exact pairs, active pairs.  Not on the timed path.
__global__ __launch_bounds__(256) void stats_kernel(const SweepParams P) {
    const int K1 = P.K + 1;
    const long long n_samples = (long long)(P.piece_end - P.piece_begin) * K1;
    unsigned long long culled = 0, pairs = 0, grads = 0;
    if (P.enable_pos)
        for (long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x; s < n_samples; s += (long long)gridDim.x * blockDim.x) {
            const int a = P.sample_info[2 * s], b = P.sample_info[2 * s + 1];
            pairs += (unsigned)a; grads += (unsigned)(b & 0x7fffffff);
            if (b < 0) culled++;
        }
    if (culled) atomicAdd(&P.stats[1], culled);
    if (pairs) atomicAdd(&P.stats[2], pairs);
    if (grads) atomicAdd(&P.stats[3], grads);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&P.stats[0], (unsigned long long)n_samples);
}

// ---------------------------------------------------------------------------------------------------------------
// tail: workgroup per piece, thread per sample (tail_kernel, or the trailing workgroups of a fused sweep launch)
// ---------------------------------------------------------------------------------------------------------------
// EARLY: the collision sums are already there (separate launch after the sweep): request them before the forward pass.
// XF: multi-GPU step with the exchange inside the launch (XFuse): tail workgroup t takes this rank's pieces first (they have
// work to overlap with the sweep), then the other ranks' pieces, whose rows it fetches from this rank's board.
template <bool EARLY, bool XF> __device__ __forceinline__ void tail_piece(const SweepParams &P, const int t, double (*s_val)[PARTIAL_STRIDE + 1]) {
    const int K1 = P.K + 1;
    const int tid = threadIdx.x;                     // threads >= TL_THREADS (fused launch: 256 per workgroup) only keep the barriers
    const int n_own = P.piece_end - P.piece_begin;
    int gp = t;                                      // global piece index (trajectory-major)
    if constexpr (XF) gp = t < n_own ? P.piece_begin + t : (t - n_own < P.piece_begin ? t - n_own : t);
    const int b = gp / P.N, i = gp - b * P.N;
    const size_t ostride = (size_t)1 + 19 * (size_t)P.N;
    const int p0 = max(P.piece_begin, b * P.N) - b * P.N, p1 = min(P.piece_end, (b + 1) * P.N) - b * P.N;   // this rank's pieces of trajectory b
    double *const o = P.out + (size_t)b * ostride;
    double *const my_board = XF ? P.xf.board[P.xf.rank] + (size_t)P.xf.parity * P.xf.parity_stride : nullptr;
    bool lost = false;     // a result slot that never arrived
    const bool mine = i >= p0 && i < p1;
    unsigned long long *const dbg = (ISDF_SWEEP_STATS && P.dbg && mine) ? P.dbg + (size_t)(P.piece_end - P.piece_begin) * K1 * 8 + (size_t)(gp - P.piece_begin) * 4 : nullptr;
    if (!mine) {
        const int q = tid;
        if constexpr (XF) {
            // a piece of another rank: its owner stores the 19 rows into this rank's board (the cost slot is the collector's)
            if (q < 19) {
                double v;
                unsigned long long deadline = 0ull;
                lost = !xslot_take(my_board + (size_t)gp * XF_ROW + q, v, P.xf.timeout_ticks, deadline);
                if (q == 18) o[1 + i] = v;
                else { const int d = q / 6, r = q - 6 * d; o[1 + P.N + (size_t)d * 6 * P.N + 6 * i + r] = v; }
            }
        } else {
            // a piece of another rank: its entries are zero here, so that the ranks' outputs sum to the full result
            if (q == 18) o[1 + i] = 0.0;
            else if (q < 18) { const int d = q / 6, r = q - 6 * d; o[1 + P.N + (size_t)d * 6 * P.N + 6 * i + r] = 0.0; }
            else if (q == 19 && p1 <= p0 && i == 0) o[0] = 0.0;    // no piece of this trajectory is ours
            return;
        }
    } else {
    if constexpr (!EARLY) {
        if (P.host_T) {          // host-direct step: this piece's inputs come through the staging buffer
            __shared__ int s_staged;
            if (tid == 0) s_staged = stage_wait(P, gp) ? 1 : 0;
            __syncthreads();
            lost = lost || !s_staged;
        }
    }
    const PieceIn pin = piece_in(P, b, i);
    const double *c = pin.c;
    const int ld = pin.ld;
    const double integral_frac = P.inv_K;
    const double step = pin.T * integral_frac;
    double colsum = 0.0;   // threads 0..19: running sum of their output column, samples in ascending order
    unsigned long long tl_c0 = 0, tl_marks = 0;
    if (dbg && tid == 0) { dbg[0] = wall_clock64(); tl_c0 = clock64(); }
#define TL_MARK(k) do { if (dbg && tid == 0) tl_marks |= (((clock64() - tl_c0) >> 4) & 0xFFFFull) << (16 * (k)); } while (0)
    // (a fused step has K + 1 <= TL_THREADS - sweep_can_fuse - and says so: with a loop the compiler hoists the constants of the
    // penalties' polynomials out of it, runs out of the fused kernels' 168 vector registers and spills them to scratch memory)
    for (int j0 = 0; j0 < (EARLY ? K1 : min(K1, 1)); j0 += TL_THREADS) {
        const int j = j0 + tid;
        if (tid < TL_THREADS && j < K1) {
            const long long s = (long long)gp * K1 + j;
            double *const ap = P.acc + (size_t)(s - (long long)P.piece_begin * K1) * ACC_STRIDE;
            double ab[EARLY ? ACC_STRIDE : 1];
            if (EARLY && P.enable_pos) {           // requested first, so the loads fly under the flatness forward pass
#pragma unroll
                for (int q = 0; q < ACC_STRIDE; q++) ab[q] = ap[q];
            }
            d3 vel, acc3, jer0;
            {
                Basis B;
                poly_basis(j * step, B);
                vel = ct_beta<1>(c, ld, B.b1); acc3 = ct_beta<2>(c, ld, B.b2); jer0 = ct_beta<3>(c, ld, B.b3);
            }
            TL_MARK(0);
            double pena = 0.0;
            d3 gradVel = mk3(0, 0, 0), gradPos = gradVel, gradOmg = gradVel;
            d4 gradQuat; gradQuat.w = gradQuat.x = gradQuat.y = gradQuat.z = 0.0;
            FlatS fs; FlatS2 ft;               // forward intermediates: shared by the penalties and the reverse mode
            flat_core(P.flat, vel, acc3, fs);
            flat_core2(P.flat, vel, acc3, jer0, fs, ft);
            if (P.enable_dyn) {
                const d4 quat = flat_quat(fs);
                const d3 omg = flat_omg(fs, ft);
                const double cos_theta = 1.0 - 2.0 * (quat.x * quat.x + quat.y * quat.y);
                const double violaVel = dot3(vel, vel) - P.vel_sqr_max;
                const double violaOmg = dot3(omg, omg) - P.omg_sqr_max;
                const double violaTheta = acos(cos_theta) - P.theta_max;
                double pe, pd;
                if (smoothed_l1_inv(violaVel, P.mu, P.inv_mu, pe, pd)) {
                    gradVel = (P.weight_v * pd * 2.0) * vel;
                    pena += P.weight_v * pe;
                }
                if (smoothed_l1_inv(violaOmg, P.mu, P.inv_mu, pe, pd)) {
                    gradOmg = (P.weight_omg * pd * 2.0) * omg;
                    pena += P.weight_omg * pe;
                }
                if (smoothed_l1_inv(violaTheta, P.mu, P.inv_mu, pe, pd)) {
                    const double f = P.weight_theta * pd / sqrt(1.0 - cos_theta * cos_theta) * 4.0;
                    gradQuat.x += f * quat.x;
                    gradQuat.y += f * quat.y;
                    pena += P.weight_theta * pe;
                }
            }
            TL_MARK(1);
            if (P.enable_pos) {
                // separate launch: the sums are there; fused launch: the sweep's workgroups run beside this one and the sums
                // may still be on their way - poll the 8 slots (requested together) until all have arrived.  Either way the
                // slots are handed back empty.
                double a[ACC_STRIDE];
                if constexpr (EARLY) {
#pragma unroll
                    for (int q = 0; q < ACC_STRIDE; q++) { a[q] = ab[q]; ap[q] = __longlong_as_double((long long)SLOT_EMPTY); }
                } else {
                    unsigned long long v[ACC_STRIDE];
                    bool ok = false;
                    for (int spin = 0; spin < SLOT_SPIN_MAX; spin++) {
                        ok = true;
#pragma unroll
                        for (int q = 0; q < ACC_STRIDE; q++) { v[q] = slot_peek(ap + q); ok = ok && v[q] != SLOT_EMPTY; }
                        if (ok) break;
                        __builtin_amdgcn_s_sleep(2);
                    }
                    lost = lost || !ok;
#pragma unroll
                    for (int q = 0; q < ACC_STRIDE; q++) { a[q] = __longlong_as_double((long long)v[q]); slot_clear(ap + q); }
                }
                if (a[0] > 0.0) {
                    gradPos.x += P.weight_p * a[1]; gradPos.y += P.weight_p * a[2]; gradPos.z += P.weight_p * a[3];
                    gradQuat.w += P.weight_p * a[4]; gradQuat.x += P.weight_p * a[5];
                    gradQuat.y += P.weight_p * a[6]; gradQuat.z += P.weight_p * a[7];
                    pena += P.weight_p * a[0];
                }
            }
            d3 gP, gV, gA, gJ;
            TL_MARK(2);
            flat_backward_from(P.flat, vel, acc3, fs, ft, gradPos, gradVel, gradQuat, gradOmg, gP, gV, gA, gJ);
            TL_MARK(3);
            const double node = (j == 0 || j == P.K) ? 0.5 : 1.0;
            const double alpha = j * integral_frac;
            double *o = s_val[tid];
            // the basis is formed again here (15 multiplications) instead of living in 60 registers across the passes above
            double s1 = j * step;
            asm volatile("" : "+v"(s1));
            Basis B;
            poly_basis(s1, B);
            const d3 jer = ct_beta<3>(c, ld, B.b3), sna = ct_beta<4>(c, ld, B.b4);      // likewise jerk and snap (same bits as above)
            const double gPv[3] = {gP.x, gP.y, gP.z}, gVv[3] = {gV.x, gV.y, gV.z}, gAv[3] = {gA.x, gA.y, gA.z}, gJv[3] = {gJ.x, gJ.y, gJ.z};
            // rows below a basis' derivative order meet its structural zeros: those products are skipped (same bits)
#pragma unroll
            for (int d = 0; d < 3; d++)
#pragma unroll
                for (int r = 0; r < 6; r++) {
                    double v = B.b0[r] * gPv[d];
                    if (r >= 1) v += B.b1[r] * gVv[d];
                    if (r >= 2) v += B.b2[r] * gAv[d];
                    if (r >= 3) v += B.b3[r] * gJv[d];
                    o[d * 6 + r] = v * node * step;
                }
            o[18] = (dot3(gP, vel) + dot3(gV, acc3) + dot3(gA, jer) + dot3(gJ, sna)) * alpha * node * step + node * integral_frac * pena;
            o[19] = node * step * pena;
        }
        __syncthreads();
        // column sums in two fixed-shape stages: TL_GROUPS partial sums per column (rows g, g + TL_GROUPS, ...), then their sum
        {
            const int n = min(TL_THREADS, K1 - j0);
            const int col = tid % PARTIAL_STRIDE, grp = tid / PARTIAL_STRIDE;
            double part = 0.0;
            if (grp < TL_GROUPS) for (int row = grp; row < n; row += TL_GROUPS) part += s_val[row][col];
            __syncthreads();                                   // every row has been read: rows 0..TL_GROUPS-1 are reused
            if (grp < TL_GROUPS) s_val[grp][col] = part;
            __syncthreads();
            if (tid < PARTIAL_STRIDE) {
#pragma unroll
                for (int g = 0; g < TL_GROUPS; g++) colsum += s_val[g][tid];
            }
        }
        __syncthreads();
    }
    if (dbg && tid == 0) { dbg[2] = tl_marks; dbg[3] = (clock64() - tl_c0) >> 4; }
#undef TL_MARK
    if (P.host_flag) lost = __syncthreads_or(lost ? 1 : 0) != 0;      // host-direct step: the overflow word must be up before the piece's cost is
    if (tid < PARTIAL_STRIDE) {
        const int q = tid;
        if (q < 19) {
            if (q == 18) o[1 + i] = colsum;
            else { const int d = q / 6, r = q - 6 * d; o[1 + P.N + (size_t)d * 6 * P.N + 6 * i + r] = colsum; }
            if constexpr (XF) {               // the same row into every peer's board
                for (int r = 0; r < P.xf.world; r++)
                    if (r != P.xf.rank) xslot_publish(P.xf.board[r] + (size_t)P.xf.parity * P.xf.parity_stride + (size_t)gp * XF_ROW + q, colsum);
            }
        } else {
            // Trajectory cost = fixed-order sum of the piece costs: each workgroup publishes its piece's cost in the piece's slot
            // (multi-GPU step: in the piece's cost slot of EVERY rank's board, this rank's included)
            // host-direct step: `out` is host-mapped memory - the rows above (same wavefront, lanes 0..18) went over PCIe and
            // have to be there before the collector, which waits for this cost, tells the host that the trajectory is
            // complete.  (Measured against the alternative - rows into device memory, the collector copying the whole
            // trajectory to the host in one coalesced sweep: 3 us slower, its copy sits on the critical path.)
            if (P.host_flag) { if (lost && P.stats) atomicMax(&P.stats[4], 1ull); __builtin_amdgcn_fence(__ATOMIC_RELEASE, ""); }
            if constexpr (XF) {
                for (int r = 0; r < P.xf.world; r++) xslot_publish(P.xf.board[r] + (size_t)P.xf.parity * P.xf.parity_stride + (size_t)gp * XF_ROW + 19, colsum);
            } else slot_publish(&P.piece_cost[gp], colsum);
        }
    }
    }   // mine
    if (lost && P.stats) atomicMax(&P.stats[4], 1ull);
    // One workgroup per trajectory collects the piece costs: the one dispatched LAST among the trajectory's (workgroups are
    // dispatched in index order, so every one it waits for is already resident or finished - here: the last piece of this
    // rank's range; multi-GPU step: the highest t, and the costs of all N pieces come from this rank's board, where their
    // owners put them).  One lane per piece polls its slot, takes the value and empties the slot for the next step; the poll
    // is bounded (NaN cost + the overflow flag instead of a hung stream).
    bool collector;
    int k0, k1;
    if constexpr (XF) {
        const int first = b * P.N, last = first + P.N - 1;
        const int t_last = (last >= P.piece_begin && last < P.piece_end) ? last - P.piece_begin : (last < P.piece_begin ? last + n_own : last);
        const int below = P.piece_begin - 1;                                   // highest piece before this rank's range
        const int t_below = (below >= first && below <= last) ? below + n_own : -1;
        collector = t == max(t_last, t_below);
        k0 = 0; k1 = P.N;
    } else {
        collector = i == p1 - 1;
        k0 = p0; k1 = p1;
    }
    if (collector && tid < 64) {
        double sum = 0.0;
        bool gone = false;
        unsigned long long deadline = 0ull;
        for (int k = k0 + tid; k < k1; k += 64) {
            double v;
            if constexpr (XF) gone = !xslot_take(my_board + ((size_t)b * P.N + k) * XF_ROW + 19, v, P.xf.timeout_ticks, deadline) || gone;
            else {
                double *slot = &P.piece_cost[(size_t)b * P.N + k];
                unsigned long long bits = slot_peek(slot);
                for (int spin = 0; bits == SLOT_EMPTY && spin < SLOT_SPIN_MAX; spin++) {
                    __builtin_amdgcn_s_sleep(1);
                    bits = slot_peek(slot);
                }
                gone = gone || bits == SLOT_EMPTY;
                slot_clear(slot);
                v = __longlong_as_double((long long)bits);
            }
            sum += v;
        }
        sum = wave_sum(sum);
        if (__ballot(gone) != 0ull) { sum = __longlong_as_double(0x7FF8000000000000ll); if (tid == 0 && P.stats) atomicMax(&P.stats[4], 1ull); }
        if (tid == 0) o[0] = sum;
    }
    if constexpr (!XF) {
        if (P.host_flag && collector && tid == 0) {
            // every piece's rows were released (system scope) before its cost became visible above: after this acquire /
            // release pair the host, once it sees the flag, sees the whole [cost | gradT | gradC] of the trajectory - the flag
            // is the ONLY thing it waits for (no copy command, no stream synchronisation)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const unsigned long long ovf = P.stats ? __hip_atomic_load(&P.stats[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            __hip_atomic_store(&P.host_flag[b], P.seq | (ovf ? HOST_FLAG_OVERFLOW : 0ull), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (dbg && tid == 0) dbg[1] = wall_clock64();
}

// The separate tail launch carries one more wavefront than its samples need: it builds the next step's dispatch order
// (plan_wave) beside the tail's own work.
constexpr int TL_LAUNCH_THREADS = TL_THREADS + 64;
template <bool XF> __device__ __forceinline__ void tail_plan(const SweepParams &P) {
    if (!(P.plan_cls_in || P.plan_lr_in) || (P.dbg_flags & 2) || (int)(threadIdx.x >> 6) != TL_THREADS / 64) return;
    int gp = (int)blockIdx.x;
    if constexpr (XF) { const int n_own = P.piece_end - P.piece_begin, t = gp; gp = t < n_own ? P.piece_begin + t : (t - n_own < P.piece_begin ? t - n_own : t); }
    if (gp >= P.piece_begin && gp < P.piece_end) plan_wave(P, gp - P.piece_begin, (int)(threadIdx.x & 63));
}
__global__ __launch_bounds__(TL_LAUNCH_THREADS) void tail_kernel(const SweepParams P) {
    __shared__ double s_val[TL_THREADS][PARTIAL_STRIDE + 1];   // +1: column reads by 20 threads hit distinct banks
    tail_plan<false>(P);
    tail_piece<true, false>(P, (int)blockIdx.x, s_val);        // one workgroup per piece of the launch
}
// Multi-GPU step that is NOT one fused launch (a 50-piece shard of a 400-piece trajectory: 813 + 50 workgroups, beyond what is
// resident at once; mesh robots; body offsets): the exchange rides in the tail launch instead - the workgroups of this
// rank's pieces (dispatched first) store their rows into every peer's board, the others poll theirs out of this rank's.
__global__ __launch_bounds__(TL_LAUNCH_THREADS) void tail_kernel_xf(const SweepParams P) {
    __shared__ double s_val[TL_THREADS][PARTIAL_STRIDE + 1];
    tail_plan<true>(P);
    tail_piece<true, true>(P, (int)blockIdx.x, s_val);
}

// Workgroups a fused launch may have: all of them resident at once, 3 per CU (the fused kernels' register budget).
// Developer override: ISDF_FUSE_MAX_BLOCKS.
static int fused_max_blocks() {
    static const int v = [] { const char *e = getenv("ISDF_FUSE_MAX_BLOCKS"); const int n = e ? atoi(e) : 0; return n > 0 ? n : 3 * 256; }();
    return v;
}

// Workgroups of a NON-fused sweep launch that are resident at once (its instantiation's waves per SIMD = workgroups per CU)
int sweep_resident_blocks(const SweepParams &P, int n_cus) {
    const bool ident = P.shape.d.ident != 0;
    const int kind = P.shape.kind == ISDF_SHAPE_MESH ? ISDF_SHAPE_MESH : P.shape.kind;
    return sweep_waves_per_simd(kind, kind == ISDF_SHAPE_MESH ? false : ident) * n_cus;
}

bool sweep_can_fuse(const SweepParams &P) {
    if (!P.enable_pos || P.shape.kind == ISDF_SHAPE_MESH || P.shape.d.ident == 0) return false;
    const long long n_samples = (long long)(P.piece_end - P.piece_begin) * (P.K + 1);
    if (n_samples <= 0 || P.K + 1 > TL_THREADS) return false;      // (the fused tail makes ONE pass over a piece's samples)
    // tail workgroups of OTHER ranks' pieces only write zeros and leave: they do not count
    return (n_samples + SW_SAMPLES - 1) / SW_SAMPLES + (long long)(P.piece_end - P.piece_begin) <= fused_max_blocks() && (long long)P.n_traj * P.N <= 8 * fused_max_blocks();
}

static int sweep_target_blocks_mesh() {      // persistent workgroups of mesh_exact_kernel: 5 per CU (96 registers; at 6 the spills cost more than the sixth wavefront brings: 2.43 -> 3.01 ms)
    static int cus = 0;                       // (asked once: the property query costs milliseconds)
    if (cus == 0) {
        int dev = 0, n = 0;
        cus = 256;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
    }
    return 5 * cus;
}
// fused = true: sweep workgroups + one tail workgroup per piece in ONE launch (the caller then skips launch_tail)
void launch_sweep(const SweepParams &P0, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop, bool fused) {
    SweepParams P = P0;
    const long long n_samples = (long long)(P.piece_end - P.piece_begin) * (P.K + 1);
    if (n_samples <= 0 || !P.enable_pos) return;
    P.n_sweep_blocks = (int)((n_samples + SW_SAMPLES - 1) / SW_SAMPLES);
    const dim3 grid((unsigned)(P.n_sweep_blocks + (fused ? P.n_traj * P.N : 0))), block(64 * SW_SAMPLES);
    const bool ident = P.shape.d.ident != 0;
    if (!fused) hipLaunchKernelGGL(pose_kernel, dim3((unsigned)((n_samples + POSE_THREADS - 1) / POSE_THREADS)), dim3(POSE_THREADS), 0, stream, P);      // the poses of every sample, one thread each
#define ISDF_LAUNCH_SWEEP(K, I, F) hipExtLaunchKernelGGL((sweep_kernel<K, I, F>), grid, block, 0, stream, ev_start, ev_stop, 0, P)
#define ISDF_SWEEP_CASE(K) case K: if (fused) ISDF_LAUNCH_SWEEP(K, true, true); else if (ident) ISDF_LAUNCH_SWEEP(K, true, false); else ISDF_LAUNCH_SWEEP(K, false, false); break;
    switch (P.shape.kind) {
    // one instantiation per analytic class (sw_manager.hpp:74-123 + Box): each inlines exactly one SDF formula
    ISDF_SWEEP_CASE(ISDF_SHAPE_TORUS) ISDF_SWEEP_CASE(ISDF_SHAPE_CAPPEDTORUS) ISDF_SWEEP_CASE(ISDF_SHAPE_CAPPEDCONE)
    ISDF_SWEEP_CASE(ISDF_SHAPE_ROUNDEDCONE) ISDF_SWEEP_CASE(ISDF_SHAPE_WIREFRAMEBOX) ISDF_SWEEP_CASE(ISDF_SHAPE_BENDLINEAR)
    ISDF_SWEEP_CASE(ISDF_SHAPE_TWISTBOX) ISDF_SWEEP_CASE(ISDF_SHAPE_BENDBOX) ISDF_SWEEP_CASE(ISDF_SHAPE_TABLE)
    ISDF_SWEEP_CASE(ISDF_SHAPE_TREFOIL) ISDF_SWEEP_CASE(ISDF_SHAPE_SMOOTHDIFFERENCE) ISDF_SWEEP_CASE(ISDF_SHAPE_SMOOTHINTERSECTION)
    ISDF_SWEEP_CASE(ISDF_SHAPE_CSG) ISDF_SWEEP_CASE(ISDF_SHAPE_BOX)
    case ISDF_SHAPE_MESH:
        if (P.mq_items) {      // scan launch -> queue -> exact launch -> per-sample sums (the events span the three)
            hipExtLaunchKernelGGL((sweep_kernel<ISDF_SHAPE_MESH, false, false>), grid, block, 0, stream, ev_start, nullptr, 0, P);
            hipLaunchKernelGGL(mesh_exact_kernel, dim3((unsigned)std::min<long long>((long long)sweep_target_blocks_mesh(), ((long long)P.mq_cap + SW_SAMPLES - 1) / SW_SAMPLES)), block, 0, stream, P);
            hipExtLaunchKernelGGL(mesh_sum_kernel, dim3((unsigned)((n_samples + 255) / 256)), dim3(256), 0, stream, nullptr, ev_stop, 0, P);
        } else ISDF_LAUNCH_SWEEP(ISDF_SHAPE_MESH, false, false);
        break;
    default: if (fused) ISDF_LAUNCH_SWEEP(-1, true, true); else if (ident) ISDF_LAUNCH_SWEEP(-1, true, false); else ISDF_LAUNCH_SWEEP(-1, false, false); break;      // Ball (and anything new)
    }
#undef ISDF_SWEEP_CASE
#undef ISDF_LAUNCH_SWEEP
}

void launch_stats(const SweepParams &P, hipStream_t stream) {
    hipLaunchKernelGGL(stats_kernel, dim3(64), dim3(256), 0, stream, P);
}

// qualifier bit-grid: bit z&31 of dword (ix*Y + iy)*ZW + (z>>5) is set iff the voxel qualifies.
// One lane per voxel, 64 consecutive z per wavefront: coalesced reads, the ballot is the two output dwords.
__global__ __launch_bounds__(256) void build_bits_kernel(DevGrid G, int use_esdf, double thresh, unsigned *out) {
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
            q = use_esdf ? ((double)G.esdf[a] <= thresh) : (G.occ[a] != 0);      // esdf <= occ_thresh, compared in double (the stored value is exact there)
        }
        const unsigned long long m = __ballot(q);
        if (lane == 0) {
            out[row * G.ZW + 2 * ch] = (unsigned)m;
            if (2 * ch + 1 < G.ZW) out[row * G.ZW + 2 * ch + 1] = (unsigned)(m >> 32);
        }
    }
}
// The same qualifier bits packed along x (AX = 0: row (y, z)) or along y (AX = 1: row (x, z)).  One lane per (row, 32-voxel
// block): lanes run along z, so the 32 reads of a lane are each coalesced across the wavefront.
template <int AX>
__global__ __launch_bounds__(256) void build_bits_axis_kernel(DevGrid G, int use_esdf, double thresh, unsigned *out) {
    const int NW = AX == 0 ? G.XW : G.YW, NA = AX == 0 ? G.X : G.Y, NU = AX == 0 ? G.Y : G.X;
    const size_t n_work = (size_t)NU * NW * G.Z;
    for (size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_work; w += (size_t)gridDim.x * blockDim.x) {
        const int z = (int)(w % G.Z);
        const size_t r = w / G.Z;
        const int blk = (int)(r % NW), u = (int)(r / NW);
        unsigned m = 0u;
        for (int k = 0; k < 32; k++) {
            const int a = blk * 32 + k;
            if (a < NA) {
                const size_t idx = AX == 0 ? ((size_t)a * G.Y + u) * G.Z + z : ((size_t)u * G.Y + a) * G.Z + z;
                const bool q = use_esdf ? ((double)G.esdf[idx] <= thresh) : (G.occ[idx] != 0);
                m |= (q ? 1u : 0u) << k;
            }
        }
        out[((size_t)u * G.Z + z) * NW + blk] = m;
    }
}
void launch_build_bits_xy(const DevGrid &G, int use_esdf, double thresh, unsigned *out_x, unsigned *out_y, hipStream_t stream) {
    hipLaunchKernelGGL(build_bits_axis_kernel<0>, dim3(4096), dim3(256), 0, stream, G, use_esdf, thresh, out_x);
    hipLaunchKernelGGL(build_bits_axis_kernel<1>, dim3(4096), dim3(256), 0, stream, G, use_esdf, thresh, out_y);
}

void launch_build_bits(const DevGrid &G, int use_esdf, double thresh, unsigned *out, hipStream_t stream) {
    hipLaunchKernelGGL(build_bits_kernel, dim3(2048), dim3(256), 0, stream, G, use_esdf, thresh, out);
}

void launch_tail(const SweepParams &P, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop) {
    if (P.xf.world > 1) hipExtLaunchKernelGGL(tail_kernel_xf, dim3(P.n_traj * P.N), dim3(TL_LAUNCH_THREADS), 0, stream, ev_start, ev_stop, 0, P);
    else hipExtLaunchKernelGGL(tail_kernel, dim3(P.n_traj * P.N), dim3(TL_LAUNCH_THREADS), 0, stream, ev_start, ev_stop, 0, P);
}

} // namespace isdf
