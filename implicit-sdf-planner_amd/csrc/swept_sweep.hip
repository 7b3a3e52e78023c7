// V1 swept-volume sweep on gfx950: addSaftyPenaOnSweptVolumeParallel
//   (src/planner_algorithm/include/planner_algorithm/back_end_optimizer.hpp:557-649)
// with the swept-volume SDF query of src/swept_volume/include/swept_volume/sw_manager.hpp
//   (getSDFofSweptVolume :710-747, choiceTInit :367-445, gradientDescent :1000-1062,
//    getSDF_DOTAtTimeStampOptimized :593-662, getStateOnTrajStamp :332-341).
//
// One optimizer step = five launches (prepare, scan, descent, back-prop, reduce; ~1.5 us between two of them on this system):
//   swept_prepare_kernel  one block: the "traj_duration" state (refreshed only when the total is < 300 s, :287-296),
//                         the coarse time table t += 0.2 by repeated fp64 addition (one lane, like the reference loop),
//                         then the pose (position + rotation) at every coarse time, shared by all obstacle points.
//   swept_scan_kernel     one WAVEFRONT per obstacle point, swept_descent_kernel one per (point, time interval): a point
//                         whose trajectory passes it several times used to run its descents one after the other (31
//                         dependent passes against 13 for the longest single interval), and the launch lasts as long as its
//                         longest wavefront.  The trajectory (durations + 6N x 3 coefficients, 152 B per piece) is staged
//                         ONCE per workgroup into LDS - every pose evaluation of every pass reads it from there (the first
//                         version spent 64 % of its wave cycles in s_waitcnt on the global table, profiles/r2a_v1_pmc_sq.txt):
//                           coarse scan   lanes = coarse times (table poses), ballot -> maximal in-range runs
//                                         (intervals; a run still open at the end is dropped, quirk q2);
//                           fine scan     lanes = 0.02 s steps of one interval (each lane accumulates its own t by
//                                         repeated addition), first-minimum reduction, running minimum and seed
//                                         carried ACROSS intervals (quirk q1);
//                           descent       the 9 step-halving candidates of one sign-descent iteration are evaluated
//                                         by 9 lanes at once, the 6 central-difference SDF samples by 6 lanes; the
//                                         remaining lanes evaluate AHEAD along the current direction (pose, point
//                                         velocity and gradient samples at the next five full steps), so a descent that
//                                         is marching with full steps advances up to six iterations per pass - same
//                                         evaluations, same accept tests, same t* as the one-by-one loop;
//                           -> (t*, sdf*, body-frame gradient) per point.
//   swept_backprop_kernel one THREAD per obstacle point: penalty (mu = 0.01), flatness reverse mode, 20-double record.
//   swept_reduce_kernel   deterministic sums: gradC rows per piece, cost, and - in the workgroup that finishes last -
//                         gradT(j) = sum over points with piece > j.
//   swept_fixed_kernel    (isdf_eval_swept_at_tstar) replaces prepare + sweep when the minimisers t* are GIVEN.
// Compute-/latency-bound: fp64 throughout, no tile traffic.
#ifndef ISDF_SWEEP_STATS
#define ISDF_SWEEP_STATS 0   // 1: per-wave time stamps for tools/v1_timing.py (developer builds; the debug pointer and the start stamps hold scalar registers)
#endif
#include "isdf_internal.hpp"
#include "dev_mesh.hpp"

namespace isdf {

constexpr int SW_WAVES = 4;
#ifndef ISDF_MESH_WPE
#define ISDF_MESH_WPE 4      // wavefronts per SIMD the workgroup-form mesh kernels are compiled for (register budget 512 / this)
#endif
constexpr int SW_MAX_RANGES = 32;                // intervals kept per point (more set the overflow flag); 5 bits of a task word
constexpr int TASK_STRIDE = 6;                   // doubles per (point, interval) slot: in (lb, ub, seed) / out (t, sdf, grad(3), counters)
constexpr int LPT_TASK = 8;                      // descents of >= this many passes (last step) are dispatched first
// ... mesh robots: descents that lasted at least TWICE the mean of the step before (words[4..6]: sum of the durations in 1.28 us units,
// their count, the threshold the prepare kernel forms from them; wall_clock64 = 100 MHz)

// The trajectory as the kernels of this file see it: durations T[0..N) (LDS copies are padded to a multiple of 4 with a
// huge value, which no time exceeds) and the 6N x 3 column-major coefficients.
template <bool LDS> struct TrajL { const double *T; const double *C; int N; };

// Trajectory::locatePieceIdx (trajectory.hpp:545-563): the same sequential subtraction as dev_math.hpp's traj_locate; the
// LDS form reads four durations per round (independent ds_reads in flight) before the wave-level "anyone still walking?".
template <bool LDS> __device__ __forceinline__ int traj_locate_l(const TrajL<LDS> &tr, double &t) {
    if constexpr (!LDS) { const TrajView v{tr.T, tr.C, tr.N}; return traj_locate(v, t); }
    else {
        int idx = 0;
        bool walking = true;
        for (int k = 0; k < tr.N; k += 4) {
            const double d0 = tr.T[k], d1 = tr.T[k + 1], d2 = tr.T[k + 2], d3_ = tr.T[k + 3];
            walking = walking && (t > d0); if (walking) { t -= d0; idx = k + 1; }
            walking = walking && (t > d1); if (walking) { t -= d1; idx = k + 2; }
            walking = walking && (t > d2); if (walking) { t -= d2; idx = k + 3; }
            walking = walking && (t > d3_); if (walking) { t -= d3_; idx = k + 4; }
            if (!__builtin_amdgcn_ballot_w64(walking)) break;
        }
        if (idx == tr.N) { idx--; t += tr.T[tr.N - 1]; }       // ran off the end: clamp to the last piece
        return idx;
    }
}
template <bool LDS> __device__ __forceinline__ void traj_eval_l(const TrajL<LDS> &tr, int piece, double t, d3 &pos, d3 &vel, d3 &acc, d3 &jer) {
    const TrajView v{tr.T, tr.C, tr.N};
    traj_eval(v, piece, t, pos, vel, acc, jer);
}
// cooperative copy of the trajectory into LDS ([T padded | coeffs]); returns the LDS view.  All threads of the block call it.
__device__ __forceinline__ TrajL<true> traj_stage(double *lds, const double *T, const double *C, int N) {
    const int Npad = (N + 3) & ~3;
    for (int k = threadIdx.x; k < Npad; k += blockDim.x) lds[k] = k < N ? T[k] : 1.7976931348623157e308;
    for (int k = threadIdx.x; k < 18 * N; k += blockDim.x) lds[Npad + k] = C[k];
    __syncthreads();
    return TrajL<true>{lds, lds + Npad, N};
}
__host__ __device__ static inline size_t traj_lds_bytes(int N) { return ((size_t)((N + 3) & ~3) + (size_t)18 * N) * sizeof(double); }
constexpr size_t TRAJ_LDS_MAX = 60 * 1024;      // beyond this (N > ~400) the kernels read the trajectory from global memory

template <bool LDS>
__device__ __forceinline__ void pose_at(const SweptParams &P, const TrajL<LDS> &tr, double t, d3 &x, m3 &R) {
    double tl = t;
    const int i = traj_locate_l(tr, tl);
    d3 v, a, j;
    traj_eval_l(tr, i, tl, x, v, a, j);
    FlatS fs;
    flat_core(P.flat, v, a, fs);
    R = quat_to_rot(flat_quat(fs));
}

// wave-uniform lane -> every lane (two v_readlane; `l` must be uniform)
__device__ __forceinline__ double rlane(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

template <bool LDS>
__global__ __launch_bounds__(1024) void swept_prepare_kernel(const SweptParams P) {
    extern __shared__ double s_dyn[];
    __shared__ int s_n;
    TrajL<LDS> tr{P.T, P.coeffs, P.N};
    if constexpr (LDS) tr = traj_stage(s_dyn, P.T, P.coeffs, P.N);
    // (host-direct step: lastTstar arrives through a staging region only the CPU writes - read by the back-prop kernel, launches later)
    if (P.tstar_stage) for (int i = P.point_begin + (int)threadIdx.x; i < P.point_end; i += (int)blockDim.x) P.tstar[i] = P.tstar_stage[i];
    if (threadIdx.x == 0) {
        double td = 0.0;
        for (int i = 0; i < P.N; i++) td += tr.T[i];           // getTotalDuration, trajectory.hpp:457-466
        double D = *P.traj_duration;
        if (td < 3 * 1e2) { D = td; *P.traj_duration = D; }    // updateTraj, sw_manager.hpp:287-296
        int n = 0;
        for (double t = 0; t < D && n < P.max_coarse; t += 0.2) P.coarse_t[n++] = t;   // choiceTInit :392
        s_n = n;
        *P.n_coarse = n;
        for (int k = 0; k < 8; k++) if (k != 4) P.stats[k] = 0ull;      // [4] = overflow: sticky until the host reads it
        P.words[0] = 0u; P.words[1] = 0u;                               // the scan kernel's task counters (front / back of the list: ONE 64-bit word)
        // mesh robots: "long" = a descent of at least twice last step's mean duration (0: no step yet)
        P.words[6] = P.words[5] ? max(1u, 2u * (P.words[4] / P.words[5])) : 0u;
        P.words[4] = 0u; P.words[5] = 0u;
    }
    __syncthreads();
    // The scan's dispatch order, longest first by last step's records - in point order the launch ends on a tail as long as the rest of
    // it.  Scheduling only.  Mesh robots (a point next to the robot's path scans for a millisecond, most for 50 us): a counting sort
    // into 16 classes (bit length of the measured duration) right here - ~10 us of LDS atomics that a millisecond step does not
    // notice.  Analytic robots (a 0.16 ms step: 10 us would eat the gain): two classes, placed by the previous step's back-prop
    // kernel (backprop_body); this kernel only empties that list's two counters.
    if (threadIdx.x == 0) { P.words[8] = 0u; P.words[9] = 0u; }
    if (P.scan_sort_here) {
        __shared__ unsigned s_cls[16];
        const int np = P.point_end - P.point_begin;
        if (threadIdx.x < 16) s_cls[threadIdx.x] = 0u;
        __syncthreads();
        for (int i = threadIdx.x; i < np; i += blockDim.x) atomicAdd(&s_cls[15 - min(15, 32 - __clz((int)P.scan_ticks[i]))], 1u);
        __syncthreads();
        if (threadIdx.x == 0) { unsigned run = 0u; for (int k = 0; k < 16; k++) { const unsigned h = s_cls[k]; s_cls[k] = run; run += h; } }
        __syncthreads();
        for (int i = threadIdx.x; i < np; i += blockDim.x) P.scan_order_out[atomicAdd(&s_cls[15 - min(15, 32 - __clz((int)P.scan_ticks[i]))], 1u)] = i;
    }
    const int n = s_n;
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
        d3 x; m3 R;
        pose_at(P, tr, P.coarse_t[k], x, R);
        // component-major: the scan's lanes (consecutive coarse times) read consecutive doubles
        double *o = P.coarse_pose + k;
        const size_t st = (size_t)P.max_coarse;
        o[0] = x.x; o[st] = x.y; o[2 * st] = x.z;
#pragma unroll
        for (int q = 0; q < 9; q++) o[(3 + q) * st] = R.a[q];
    }
}

// lane i <- lane i + 1 over the whole wavefront (DPP wave_shl:1, gfx9; the last lane keeps its own value)
__device__ __forceinline__ double wave_shl1(double v) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), 0x130, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// first lane holding the minimum of `v` over the active prefix [0, n); returns the minimum
__device__ __forceinline__ double wave_first_min(double v, int lane, int n, int &arg) {
    double m = (lane < n) ? v : 1.7976931348623157e308;
    // butterfly without LDS round trips (a minimum does not depend on the order it is taken in): halves and rows of the
    // wavefront through gfx950's permlane swaps, the rest through DPP row operations
    {
        const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(m), __double2loint(m), false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(m), __double2hiint(m), false, false);
        m = fmin(__hiloint2double(hi[0], lo[0]), __hiloint2double(hi[1], lo[1]));
    }
    {
        const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(m), __double2loint(m), false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(m), __double2hiint(m), false, false);
        m = fmin(__hiloint2double(hi[0], lo[0]), __hiloint2double(hi[1], lo[1]));
    }
    m = fmin(m, dpp_mov_f64<0x128>(m));      // row_ror:8
    m = fmin(m, dpp_mov_f64<0x141>(m));      // row_half_mirror
    m = fmin(m, dpp_mov_f64<0x4E>(m));       // quad_perm [2,3,0,1]
    m = fmin(m, dpp_mov_f64<0xB1>(m));       // quad_perm [1,0,3,2]
    const unsigned long long eq = __ballot((lane < n) && (v == m));
    arg = eq ? (__ffsll((long long)eq) - 1) : -1;
    return m;
}

// ---- mesh robots: ONE (point / interval) per WORKGROUP.  The wavefronts of the workgroup (two in the scan, one in the descent) run
// the same control flow on the same numbers (so every ballot, every branch and every barrier below is taken by all of them alike);
// where the lanes of a pass evaluate the mesh SDF, the ACTIVE queries are dealt to the wavefronts, 16 per wavefront and round, 4 lanes
// per query (dev_mesh.hpp, quad-cooperative walks) - and the results come back through LDS.  A pass next to the surface (all of the
// hierarchy opened by 25-64 queries) was one wavefront's 1e5 dependent instructions in round 3.
struct MeshWG {
    double *stage;       // this wavefront's [64][4] compacted queries: point, cap (wave-private: no barrier)
    double *res;         // the workgroup's [2 parities][64][5] results: (below, sdf, g.xyz)
    int parity, wave;
    int count;           // queries pending in stage (mesh_wg_push ... mesh_wg_flush)
    MeshFrames frames;   // this wavefront's walk frames (dev_mesh.hpp): a view into the dynamic LDS, as many levels as the hierarchy has
    int n_waves;         // wavefronts of the workgroup (2 in the scan: up to 64 queries per pass; 1 in the descent: at most 14, 19 in a first pass without a seed)
    FlatLds flat;        // small meshes (FLAT form: every wavefront has a task of its own, stage / res are the wavefront's): the LDS copy of the tables
};
constexpr int MF_QUAD = 1, MF_FLAT = 2;      // forms of the cooperative evaluation (template parameter FORM below)
constexpr int MESH_WG_RES = 2 * 64 * 5;
constexpr int mesh_wg_stage(int nw) { return nw * 64 * 4; }
// The queries of a pass go through three steps (every lane of all the workgroup's wavefronts calls each of them):
//   mesh_wg_push   the evaluating lanes' queries join the pending list (compacted into the wavefront's stage); returns the lane's place
//                  in it, or -1 when the lane asks nothing (or the distance lattice answers it: fewer queries, fewer rounds)
//   mesh_wg_flush  the pending list is evaluated, 16 queries per wavefront and round, 4 lanes per query
//   mesh_wg_get    a lane's result: "sdf < cap proven / evaluated" like mesh_sdf_grad_below (cap = +inf: plain query), value, gradient
// A list holds up to 64 queries (stage / res); several pushes may precede one flush - the coarse scan collects the queries of ALL its
// 64-sample windows before it evaluates them: a round costs what the union of its queries' nodes costs, whether it holds 3 queries or 16.
// FLAT: small meshes - the flat evaluation (dev_mesh.hpp) instead of the walks.
__device__ __forceinline__ int mesh_wg_push(MeshWG &X, const DevMesh *m, d3 pr, bool act, double cap) {
    const int lane = (int)__lane_id();
    if (act && mesh_lattice_not_below(m, pr, cap)) act = false;
    const unsigned long long mask = __ballot(act);
    if (!mask) return -1;                                  // (the same in all the workgroup's wavefronts)
    const int idx = X.count + __popcll(mask & ((1ull << lane) - 1ull));
    if (act) { X.stage[4 * idx] = pr.x; X.stage[4 * idx + 1] = pr.y; X.stage[4 * idx + 2] = pr.z; X.stage[4 * idx + 3] = cap; }
    X.count += __popcll(mask);
    return act ? idx : -1;
}
template <int FORM>
__device__ inline void mesh_wg_flush(MeshWG &X, const DevMesh *m) {
    constexpr bool FLAT = FORM == MF_FLAT;
    const int lane = (int)__lane_id();
    const int n_act = X.count;
    X.count = 0;
    if (n_act == 0) return;                                // (the same in all the workgroup's wavefronts)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // FULL wavefronts first (16 queries, 4 lanes each): the queries of a pass are neighbours, so every wavefront that takes part
    // opens nearly the same set of nodes - dealing 25 queries as 7 + 7 + 7 + 4 cost four walks of 28 lanes (counters: lane
    // utilisation 0.48, 1.6x the instructions of round 3), 16 + 9 costs two
    // (res is double-buffered by parity: a wavefront that races ahead writes the OTHER half, which every wavefront finished reading
    // before it passed the barrier below; FLAT: one wavefront, one buffer)
    if constexpr (!FLAT) X.parity ^= 1;
    double *R = X.res + (FLAT ? 0 : (size_t)X.parity * 64 * 5);
    for (int base = 0; base < n_act; base += 16 * X.n_waves) {
        const int q = lane >> 2, r = base + X.wave * 16 + q;
        const bool on = r < n_act;
        if (!__ballot(on)) continue;                       // (wave-uniform; no barrier inside this loop)
        const int rr = on ? r : 0;
        const d3 qp = mk3(X.stage[4 * rr], X.stage[4 * rr + 1], X.stage[4 * rr + 2]);
        const double qcap = X.stage[4 * rr + 3];                  // (a query's own cap: the lanes of a pass may ask different questions)
        double s2 = 0.0;
        d3 g2 = mk3(0, 0, 0);
        bool below;
        if constexpr (FLAT) below = mesh_sdf_grad_below_flat(X.flat, qp, on, qcap, s2, g2);
        else below = mesh_sdf_grad_below_quad(m, X.frames, qp, on, qcap, s2, g2);
        if (on && (lane & 3) == 0) { double *o = R + 5 * r; o[0] = below ? 1.0 : 0.0; o[1] = s2; o[2] = g2.x; o[3] = g2.y; o[4] = g2.z; }
    }
    // (FLAT: the wavefront works alone - its own task, its own stage / res - and the other wavefronts of the workgroup are elsewhere
    // in their own control flow: no workgroup barrier)
    if constexpr (FLAT) flat_lds_sync(); else __syncthreads();
}
template <int FORM>
__device__ __forceinline__ bool mesh_wg_get(const MeshWG &X, int idx, double &sdf, d3 &g) {
    constexpr bool FLAT = FORM == MF_FLAT;
    if (idx < 0) return false;
    const double *o = X.res + (FLAT ? 0 : (size_t)X.parity * 64 * 5) + 5 * idx;
    sdf = o[1]; g = mk3(o[2], o[3], o[4]);
    return o[0] != 0.0;
}
// one pass by itself
template <int FORM>
__device__ inline bool mesh_eval_wg(MeshWG &X, const DevMesh *m, d3 pr, bool act, double cap, double &sdf, d3 &g) {
    const int idx = mesh_wg_push(X, m, pr, act, cap);
    mesh_wg_flush<FORM>(X, m);
    return mesh_wg_get<FORM>(X, idx, sdf, g);
}

// ---- the ONE pose + SDF evaluation both kernels below are built around: pose at `te`, body-frame point (perturbed for a
// gradient sample), shape SDF; pose lanes also form the body-frame point velocity (getSDF_DOTAtTimeStampOptimized :593-662).
// Analytic kinds: called by the lanes that evaluate.  Mesh kind: called by ALL lanes of the workgroup, `act` says which evaluate.
// the pose half: body-frame point of p at time te and, on pose lanes, its body-frame velocity; R comes back for the analytic kinds' own gradient
template <bool LDS>
__device__ __forceinline__ void eval_pose(const SweptParams &P, const TrajL<LDS> &tr, const d3 p, const double te, const bool pose_lane, d3 &pr, d3 &pvel) {
    double tl = te;
    const int pi = traj_locate_l(tr, tl);
    d3 xt, vt, at, jt;
    traj_eval_l(tr, pi, tl, xt, vt, at, jt);
    FlatS fs;
    flat_core(P.flat, vt, at, fs);
    const m3 R = quat_to_rot(flat_quat(fs));
    pr = mat_tmul(R, p - xt);
    if (pose_lane) {
        FlatS2 ft;
        flat_core2(P.flat, vt, at, jt, fs, ft);
        const d3 omg = flat_omg(fs, ft);
        const d3 wx = mk3(0 * pr.x + (-omg.z) * pr.y + omg.y * pr.z, omg.z * pr.x + 0 * pr.y + (-omg.x) * pr.z,
                          (-omg.y) * pr.x + omg.x * pr.y + 0 * pr.z);
        const d3 rv = mat_tmul(R, vt);
        pvel = mk3(-(rv.x + wx.x), -(rv.y + wx.y), -(rv.z + wx.z));
    }
}
template <int KIND, bool LDS, int WG = 0, bool CAPPED = false>      // WG: 0 per lane / wave-cooperative, else the cooperative form (MF_QUAD, MF_FLAT)
__device__ __forceinline__ void eval_lane(const SweptParams &P, const TrajL<LDS> &tr, const d3 p, const double te, const int pert,
                                          const bool pose_lane, const bool own_grad, double &val, d3 &pvel, d3 &g_direct,
                                          MeshWG *X = nullptr, const bool act = true, const double cap = __builtin_inf()) {
    d3 pr;
    eval_pose(P, tr, p, te, pose_lane, pr, pvel);
    if constexpr (KIND != ISDF_SHAPE_MESH) if (pose_lane && own_grad) g_direct = shape_grad<KIND>(P.shape, pr);     // Ball: own gradient
    if (pert >= 1 && pert <= 6) {            // temp(a) -= dx; [temp(a) += 2 dx]  (Shape.hpp:36-54)
        const int a = (pert - 1) >> 1;
        const double dx = 0.000005;
        double c = a == 0 ? pr.x : (a == 1 ? pr.y : pr.z);
        c = c - dx;
        if ((pert - 1) & 1) c = c + 2 * dx;
        if (a == 0) pr.x = c; else if (a == 1) pr.y = c; else pr.z = c;
    } else if (pert >= 7) {                  // Box::getonlyGrad1: forward difference, dx = 0.01 (Shape.hpp:2363-2377)
        const int a = pert - 7;
        if (a == 0) pr.x += 0.01; else if (a == 1) pr.y += 0.01; else pr.z += 0.01;
    }
    if constexpr (WG != 0) {
        // one hierarchy query per evaluating lane gives value AND gradient (the pose lanes keep the gradient)
        d3 gq = mk3(0, 0, 0);
        double v = 1e300;
        const bool below = mesh_eval_wg<WG>(*X, P.shape.mesh, pr, act, cap, v, gq);      // (cap: "sdf >= cap" may be answered without the value)
        if (act && below) val = v;
        if (act && pose_lane) g_direct = gq;
    } else if constexpr (KIND == ISDF_SHAPE_MESH) {
        d3 gq = mk3(0, 0, 0);
        if constexpr (CAPPED) {                                  // (the scan's fine passes: only "val < cap?" and the value when yes)
            double v = 1e300;
            if (mesh_sdf_grad_below_wave(P.shape.mesh, pr, cap, v, gq)) val = v;
        } else val = mesh_sdf_grad_wave(P.shape.mesh, pr, gq);   // the executing lanes walk the hierarchy together (dev_mesh.hpp)
        if (pose_lane) g_direct = gq;
    } else {
        val = shape_sdf<KIND>(P.shape, pr);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// swept_scan_kernel: one wavefront per obstacle point - choiceTInit (:367-445)
// ---------------------------------------------------------------------------------------------------------------
// coarse scan (lanes = coarse times, table poses) -> maximal in-range runs; then the fine scans of the runs, one after the
// other because the running minimum and the seed are carried ACROSS runs (quirk q1).  Leaves, per point, the number of
// intervals and per interval (lb, ub, seed) in the point's task slots.  The workgroup that finishes last turns the interval
// counts into the task list of the descent kernel (exclusive scan in point order: deterministic).
// (WGM: at least 4 wavefronts per SIMD - without the cap the compiler spends 450 registers, one workgroup per CU)
template <int KIND, bool LDS, int WGM>
__device__ __forceinline__ void scan_body(const SweptParams &P, double *s_dyn, const int block) {
    constexpr bool COOP = WGM != 0;      // the evaluating lanes' mesh queries go through mesh_eval_wg (compacted, 4 lanes per query)
    constexpr bool FLAT = WGM == 3;      // ... small meshes: the flat evaluation instead of the walks; one point per WAVEFRONT like the analytic kinds
    constexpr bool MESH = COOP && !FLAT; // one point per WORKGROUP of WGW wavefronts (mesh_eval_wg above: larger mesh robots), else one per wavefront
    constexpr int WGW = FLAT ? 1 : WGM, EW = FLAT ? MF_FLAT : MF_QUAD;
    static_assert(!WGM || KIND == ISDF_SHAPE_MESH, "workgroup-cooperative evaluation is the mesh kind's");
    __shared__ double s_rl[SW_WAVES][SW_MAX_RANGES], s_rr[SW_WAVES][SW_MAX_RANGES];
    __shared__ unsigned s_cnt[SW_WAVES][2], s_base[2];
    __shared__ double s_mesh[FLAT ? SW_WAVES * (64 * 4 + 64 * 5) : (MESH ? mesh_wg_stage(WGW) + MESH_WG_RES : 1)];
    const unsigned long long dbg_t0 = (ISDF_SWEEP_STATS && P.dbg) ? wall_clock64() : 0ull;
    TrajL<LDS> tr{P.T, P.coeffs, P.N};
    if constexpr (LDS) tr = traj_stage(s_dyn, P.T, P.coeffs, P.N);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned long long scan_t0 = KIND == ISDF_SHAPE_MESH ? wall_clock64() : 0ull;
    int pt_rel = MESH ? block : block * SW_WAVES + wave;      // through the longest-first order of the prepare kernel
    if (P.scan_order && pt_rel < P.point_end - P.point_begin) pt_rel = __builtin_amdgcn_readfirstlane(P.scan_order[pt_rel]);
    const int pt = P.point_begin + pt_rel;
    const bool writer = lane == 0 && (!MESH || wave == 0);       // (the four wavefronts of a mesh workgroup hold the same results)
    const int m_lv = MESH ? P.shape.mesh_levels : 1;       // frames: behind the staged trajectory in the dynamic LDS, one set per wavefront
    MeshWG mwg{s_mesh + wave * 64 * 4, s_mesh + (MESH ? mesh_wg_stage(WGW) : 0), 0, wave, 0,
               MeshFrames{reinterpret_cast<unsigned *>(reinterpret_cast<char *>(s_dyn) + (LDS ? traj_lds_bytes(P.N) : 0)) + (MESH ? wave * mesh_frames_words(m_lv) : 0), m_lv}, WGW, FlatLds{}};
    if constexpr (FLAT) {
        // dynamic LDS: [the mesh's table blob | SW_WAVES x this wavefront's slot values]; static: per wavefront [64][4] queries + [64][5] results
        int *blob = reinterpret_cast<int *>(reinterpret_cast<char *>(s_dyn) + (LDS ? ((traj_lds_bytes(P.N) + 15) & ~(size_t)15) : 0));
        flat_lds_stage(P.shape.mesh, blob);
        mwg.stage = s_mesh + wave * (64 * 4 + 64 * 5); mwg.res = mwg.stage + 64 * 4; mwg.wave = 0;
        mwg.flat = flat_lds_view(blob, reinterpret_cast<float *>(blob + P.shape.mesh_flat_words) + wave * flat_om_floats(P.shape.mesh_flat_slots));
    }
    int my_nr = 0;
    if (pt < P.point_end) {
    unsigned long long *const dbg = (ISDF_SWEEP_STATS && P.dbg) ? P.dbg + (size_t)pt * 4 : nullptr;
    if (dbg && writer) { dbg[0] = dbg_t0; dbg[1] = wall_clock64(); }
    const d3 p = mk3(P.points[3 * pt], P.points[3 * pt + 1], P.points[3 * pt + 2]);
    const double D = *P.traj_duration;
    const int n_coarse = *P.n_coarse;
    const double rough_dt = 0.2, dt = 0.02;
    const double inf = 2 * P.safety_hor + 0.1;

    // ---------------- coarse scan -> intervals (:392-421) ----------------
    int n_ranges = 0;
    bool in_range = false;
    double tou_lb = 0.0, range_mindis = 1e9;
    // Coarse samples whose pose is farther from the point than the robot's bounding sphere plus the band cannot be in range
    // (isdf_shape.bound_radius: sdf(q) >= |q| - R for every body-frame q, and |R^T (p - x)| = |p - x|): they skip the rotation
    // and the SDF - a whole 64-sample window does when none of its poses is near, which is most windows of most points.
    const size_t cst = (size_t)P.max_coarse;
    // (the mesh kind's sdf is (1 - 2 w) * distance with libigl's APPROXIMATE winding number, |w error| a few 1e-3: its band is widened by 5 %)
    const double far_r = P.shape.bound_radius + inf * (KIND == ISDF_SHAPE_MESH ? 1.05 : 1.0);
    const double far2 = P.shape.bound_radius > 0.0 ? far_r * far_r : 1.0e300;
    bool carry_din = false;          // (mesh robots) the previous window's last sample is in range by the lattice's bounds
    // the transitions of one 64-sample window (wave-uniform scalar loop): in_mask = its in-range samples, dis / ct = the lanes' values / times
    auto walk_window = [&](const int base, const unsigned long long in_mask, const double dis, const double ct) {
        const int nvalid = min(64, n_coarse - base);
        int b = 0;
        while (b < nvalid) {
            const unsigned long long rest = (b < 64) ? (in_mask >> b) : 0ull;
            if (!in_range) {
                if (!rest) break;                                  // no further entry in this window
                b += __ffsll((long long)rest) - 1;                 // first in-range sample: run entry
                in_range = true;
                range_mindis = rlane(dis, __builtin_amdgcn_readfirstlane(b));
                tou_lb = fmax(0.0, rlane(ct, __builtin_amdgcn_readfirstlane(b)) - rough_dt);
                b++;
            } else {
                const unsigned long long nrest = (~rest) & ((nvalid - b >= 64) ? ~0ull : ((1ull << (nvalid - b)) - 1ull));
                if (!nrest) break;                                 // run continues past this window
                b += __ffsll((long long)nrest) - 1;                // first out-of-range sample: run exit
                in_range = false;
                const double tou_ub = fmin(D, rlane(ct, __builtin_amdgcn_readfirstlane(b)) + rough_dt);
                if (n_ranges < SW_MAX_RANGES) {
                    if (lane == 0) { s_rl[wave][n_ranges] = tou_lb; s_rr[wave][n_ranges] = tou_ub; }
                } else if (writer) {
                    atomicOr((unsigned int *)&P.stats[4], 1u);
                }
                n_ranges++;
                b++;
            }
        }
    };
    if constexpr (COOP) {
        // Mesh robots, cooperative forms.  Which samples need a query does not depend on any query's result (the distance lattice and
        // the bounding sphere decide), so the queries of SEVERAL windows can be evaluated together.  FLAT evaluation: a round costs the
        // same whether it holds 3 queries or 16 - a 20 s trajectory's two windows were two rounds per point (Lthick.obj 1.115 ->
        // 1.074 ms).  Quad WALKS: one window per round, as before - a walk is cheap because its queries are neighbours (they open the
        // same nodes, and the closest-point walk visits nearest-first by its FIRST query's bounds); samples 13 s apart on the trajectory
        // share nothing (measured with groups of 8: Trefoil.obj 4.57 -> 6.03 ms, the 720-face blob 5.34 -> 5.59).  Per window three
        // wave-uniform words are kept (asked lanes, lanes in range without a query, the first place in the list); a group ends after
        // SW_GROUP windows or when the list (64 queries) would overflow.
        constexpr int SW_GROUP = FLAT ? 8 : 1;
        __shared__ unsigned long long s_wm[SW_WAVES][SW_GROUP][2];
        __shared__ int s_wi[SW_WAVES][SW_GROUP][2];
        int g_n = 0;
        auto flush_group = [&]() {
            mesh_wg_flush<EW>(mwg, P.shape.mesh);
            flat_lds_sync();
            for (int w = 0; w < g_n; w++) {
                const int base = s_wi[wave][w][0], lbase = s_wi[wave][w][1];
                const unsigned long long amask = s_wm[wave][w][0], smask = s_wm[wave][w][1];
                const int k = base + lane, kc = min(k, n_coarse - 1);
                const double ct = P.coarse_t[kc];
                double dis = 1e9;
                if ((amask >> lane) & 1ull) {
                    d3 g_unused;
                    double sdf = 1e9;
                    if (mesh_wg_get<EW>(mwg, lbase + __popcll(amask & ((1ull << lane) - 1ull)), sdf, g_unused)) dis = sdf;
                }
                // (a sample the lattice put in range without a query - `skip` below - is never the first of a run: only its being in range is used)
                const unsigned long long in_mask = __ballot((k < n_coarse) && (dis < inf)) | smask;
                walk_window(base, in_mask, dis, ct);
            }
            g_n = 0;
        };
        for (int base = 0; base < n_coarse; base += 64) {
            const int k = base + lane;
            const int kc = min(k, n_coarse - 1);
            const double *o = P.coarse_pose + kc;     // component-major table: coalesced across the lanes
            const d3 dxp = p - mk3(o[0], o[cst], o[2 * cst]);
            const bool near = k < n_coarse && !(dot3(dxp, dxp) > far2);
            unsigned long long amask = 0ull, smask = 0ull;
            int lbase = mwg.count;
            if (__ballot(near)) {
                m3 R;
#pragma unroll
                for (int q = 0; q < 9; q++) R.a[q] = o[(3 + q) * cst];
                const d3 pr = mat_tmul(R, dxp);
                // the scan only asks "dis < inf?" (and keeps dis where the answer is yes): far samples are rejected by the
                // hierarchy's distance cutoff without a full query - or by the distance lattice (mesh_wg_push).  A sample the
                // lattice puts INSIDE the band, right behind another such sample, needs no query either: it is in range, and it
                // cannot be the first sample of a run - the only one whose value is used (:405-408)
                double s_lb, s_ub;
                const bool din = near && mesh_lattice_bounds(P.shape.mesh, pr, s_lb, s_ub) && s_ub < inf * (1.0 - 1e-9);
                const unsigned long long dmask = __ballot(din);
                const bool skip = din && (lane == 0 ? carry_din : (bool)((dmask >> (lane - 1)) & 1ull));
                carry_din = (dmask >> 63) & 1ull;
                const bool ask = near && !skip;
                if (mwg.count + __popcll(__ballot(ask)) > 64) { flush_group(); lbase = 0; }
                const int idx = mesh_wg_push(mwg, P.shape.mesh, pr, ask, inf);
                amask = __ballot(idx >= 0); smask = __ballot(skip);
            } else carry_din = false;
            if (lane == 0) { s_wm[wave][g_n][0] = amask; s_wm[wave][g_n][1] = smask; s_wi[wave][g_n][0] = base; s_wi[wave][g_n][1] = lbase; }
            g_n++;
            if (g_n == SW_GROUP) flush_group();
        }
        flush_group();
    } else
    for (int base = 0; base < n_coarse; base += 64) {
        const int k = base + lane;
        const int kc = min(k, n_coarse - 1);
        const double ct = P.coarse_t[kc];         // travels with the lane: the run boundaries below come out of a register
        const double *o = P.coarse_pose + kc;     // component-major table: coalesced across the lanes
        const d3 dxp = p - mk3(o[0], o[cst], o[2 * cst]);
        double dis = 1e9;
        const bool near = k < n_coarse && !(dot3(dxp, dxp) > far2);
        if (near) {
            m3 R;
#pragma unroll
            for (int q = 0; q < 9; q++) R.a[q] = o[(3 + q) * cst];
            const d3 pr = mat_tmul(R, dxp);
            if constexpr (KIND == ISDF_SHAPE_MESH) {
                d3 g_unused;
                double sdf;
                dis = mesh_sdf_grad_below_wave(P.shape.mesh, pr, inf, sdf, g_unused) ? sdf : 1e9;      // (the lattice answers the far ones: dev_mesh.hpp)
            } else {
                dis = shape_sdf<KIND>(P.shape, pr);
            }
        }
        const unsigned long long in_mask = __ballot((k < n_coarse) && (dis < inf));
        walk_window(base, in_mask, dis, ct);
    }
    n_ranges = min(n_ranges, SW_MAX_RANGES);
    if (dbg && writer) dbg[2] = wall_clock64();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---------------- fine scans (:423-438): lanes = 0.02 s steps, each accumulating its own t by repeated addition
    double range_time_seed = 0.0;
    // (mesh robots, workgroup form: the seed also carries its SDF value and dSDF/dt there - the descent's first pass then knows f(x)
    // and on which side its candidates lie, see descent_body; NaN: no fine sample has been the minimum yet)
    double seed_f = 0.0, seed_gd = __builtin_nan("");
    int n_pass = 0;
    double *const slots = P.task_buf + (size_t)(pt - P.point_begin) * SW_MAX_RANGES * TASK_STRIDE;
    for (int r = 0; r < n_ranges; r++) {
        const double lb = s_rl[wave][r], ub = s_rr[wave][r];
        double t_base = lb;
        for (int guard = 0; guard < (1 << 20); guard++) {
            double te = t_base;
            for (int q = 0; q < 63; q++) if (q < lane) te += dt;
            const bool act = te < ub;
            if (!__ballot(act)) break;          // nothing (left) to scan in this interval
            double val = 1e300;
            d3 pv_unused = mk3(0, 0, 0), g_unused = pv_unused;
            // (mesh robots: only a value BELOW the running minimum can change it - the hierarchy's distance cutoff answers the rest)
            if constexpr (KIND == ISDF_SHAPE_MESH) {
                // Which samples of this pass can hold its minimum?  The distance lattice brackets every sample's SDF (dev_mesh.hpp,
                // mesh_lattice_bounds: a few centimetres); a sample whose lower bound lies above the smallest UPPER bound of the pass
                // cannot be the first minimum, nor tie with it - it needs no query.  An obstacle point the robot passes over keeps
                // the samples around the deepest one, a handful of 64.
                const double te_q = act ? te : lb;
                d3 prq;
                eval_pose(P, tr, p, te_q, COOP, prq, pv_unused);      // (ONE pose evaluation per sample: the lattice and the query see the same point)
                double s_lb, s_ub;
                const bool known = mesh_lattice_bounds(P.shape.mesh, prq, s_lb, s_ub);
                int a_ub;
                const double ub_min = wave_first_min((act && known) ? s_ub : 1.7976931348623157e308, lane, 64, a_ub);
                const bool keep = act && !(known && s_lb > ub_min);
                if constexpr (COOP) {
                    d3 gq = mk3(0, 0, 0);
                    double v = 1e300;
                    if (mesh_eval_wg<EW>(mwg, P.shape.mesh, prq, keep, range_mindis, v, gq) && keep) val = v;
                    if (keep) g_unused = gq;
                } else if (keep) {
                    d3 gq = mk3(0, 0, 0);
                    double v = 1e300;
                    if (mesh_sdf_grad_below_wave(P.shape.mesh, prq, range_mindis, v, gq)) val = v;
                }
            }
            else if (act) eval_lane<KIND, LDS, 0, false>(P, tr, p, te, 0, false, false, val, pv_unused, g_unused, nullptr, true, range_mindis);
            n_pass++;
            const int nv = __popcll(__ballot(act));                  // valid lanes form a prefix (t is increasing)
            int arg;
            const double m = wave_first_min(val, lane, nv, arg);
            if (arg >= 0 && m < range_mindis) {
                range_mindis = m; range_time_seed = rlane(te, __builtin_amdgcn_readfirstlane(arg));
                if constexpr (COOP) { seed_f = m; seed_gd = rlane(dot3(g_unused, pv_unused), __builtin_amdgcn_readfirstlane(arg)); }
            }
            if (nv < 64) break;
            t_base = rlane(te, 63) + dt;
        }
        if (writer) {
            double *o = slots + (size_t)r * TASK_STRIDE; o[0] = lb; o[1] = ub; o[2] = range_time_seed;
            if constexpr (COOP) { o[3] = seed_f; o[4] = seed_gd; }
        }
    }
    if (writer) {
        P.point_nr[pt] = n_ranges | (n_pass << 8);
        // this point's record for the next step's dispatch order of the scan: mesh robots the measured duration (a pass next to the
        // surface costs ten times one far from it), analytic robots the fine passes (a pass costs the same everywhere)
        if (P.scan_ticks) P.scan_ticks[pt - P.point_begin] = KIND == ISDF_SHAPE_MESH ? (unsigned)min((wall_clock64() - scan_t0) >> 7, 0x7FFFFFFFull) : (unsigned)n_pass;
        if (dbg) dbg[3] = (wall_clock64() << 16) | (unsigned long long)(n_pass & 0xFFF) | ((unsigned long long)(n_ranges & 0xF) << 12);
    }
    my_nr = n_ranges;
    }   // pt < point_end

    // ---------------- this workgroup's intervals join the descent kernel's task list.
    // The descent launch lasts as long as its longest wavefront, and wavefronts that start late make that worse; the trajectory
    // moves little between optimizer steps, so the previous step's pass counts say which descents are long: tasks that took
    // >= LPT_TASK passes last step go to the FRONT of the list, the others fill it from the BACK (124 -> 92 us on the bench map).
    // Scheduling only: which wavefront evaluates what has no influence on any result (every task has its own result slot), so
    // the list needs no particular order within the two classes: ONE reservation per workgroup and class (two atomics; round 3
    // built the list in point order in the workgroup that finished last - 15 us of one workgroup between the two launches).
    // (words[0] / words[1]: tasks at the front / at the back - one 64-bit counter, one atomic per workgroup; zeroed by the prepare kernel.)
    {
        unsigned vt = 0u;                                       // bit r: task r of my point was long last step
        if (pt < P.point_end && my_nr > 0 && (!MESH || wave == 0)) {
            const unsigned m = P.point_lmask[pt];
            vt = my_nr >= 32 ? m : (m & ((1u << my_nr) - 1u));
            if (m && lane == 0) P.point_lmask[pt] = 0u;         // consumed (the descent kernel sets it again)
        }
        const int nr_w = (pt < P.point_end && (!MESH || wave == 0)) ? my_nr : 0;
        const unsigned nl = (unsigned)__popc(vt), ns = (unsigned)nr_w - nl;
        if (lane == 0) { s_cnt[wave][0] = nl; s_cnt[wave][1] = ns; }
        __syncthreads();
        if (tid == 0) {
            unsigned tl = 0, ts = 0;
            for (int w = 0; w < (MESH ? WGW : SW_WAVES); w++) { tl += s_cnt[w][0]; ts += s_cnt[w][1]; }
            unsigned long long got = 0ull;
            if (tl | ts) got = atomicAdd(reinterpret_cast<unsigned long long *>(P.words), (unsigned long long)tl | ((unsigned long long)ts << 32));
            s_base[0] = (unsigned)got;
            s_base[1] = (unsigned)(got >> 32);
        }
        __syncthreads();
        unsigned o_l = s_base[0], o_s = s_base[1];
        for (int w = 0; w < wave; w++) { o_l += s_cnt[w][0]; o_s += s_cnt[w][1]; }
        const unsigned cap = (unsigned)(P.point_end - P.point_begin) * SW_MAX_RANGES;      // task_map region of this shard: [0, cap)
        if (lane < nr_w) {
            const unsigned word = ((unsigned)(pt - P.point_begin) << 5) | (unsigned)lane;
            const unsigned below = (1u << lane) - 1u;
            if ((vt >> lane) & 1u) P.task_map[o_l + (unsigned)__popc(vt & below)] = word;
            else P.task_map[cap - 1u - (o_s + (unsigned)__popc(~vt & below))] = word;
        }
    }
}
template <int KIND, bool LDS, int WGM = 0>
__global__ __launch_bounds__(64 * SW_WAVES, WGM ? ISDF_MESH_WPE : 1) void swept_scan_kernel(const SweptParams P) {
    extern __shared__ double s_dyn[];
    scan_body<KIND, LDS, WGM>(P, s_dyn, (int)blockIdx.x);
}

// ---------------------------------------------------------------------------------------------------------------
// swept_descent_kernel: one wavefront per (point, interval) - gradientDescent (:1000-1062) from the interval's seed
// ---------------------------------------------------------------------------------------------------------------
// lanes of one sign-descent pass:
//   0-8    candidates x - tau_k        9-17  candidates x + tau_k          (tau_k = 0.02 / 2^k)
//   18     pose at x (f(x), body-frame point velocity)      19-24  gradient samples at x
//   block m = 1..5, base 25 + 7 (m - 1):  base = pose at y_m, base+1..base+6 = gradient samples at y_m
//   60     f(y_6)
// with y_0 = x, y_(m+1) = clamp(y_m - 0.02 s): the next full steps in the direction s of the previous pass.
constexpr int GD_SPEC = 5;                       // look-ahead blocks
__device__ __forceinline__ int gd_block_base(int m) { return m == 0 ? 18 : 25 + 7 * (m - 1); }

// (Round 4, measured and dropped: scan and descent as ONE launch - the descent workgroups riding behind the scan's, taking every
// interval the moment it is published through task words that are their own ready flags.  Correct, but the fused kernel needs the
// descent's 125 registers for the scan role too (4 instead of 6 wavefronts per SIMD: scan span 75 -> 167 us) and its 1 944 descent
// workgroups do not fit beside each other: 0.43 ms against 0.129 ms for the two launches, profiles/r4_v1_timing_fused_search_dropped.txt.)
template <int KIND, bool LDS, int WGM>
__device__ __forceinline__ void descent_body(const SweptParams &P, double *s_dyn, const unsigned block, const unsigned n_blocks) {
    constexpr bool COOP = WGM != 0;      // mesh queries through mesh_eval_wg, DIRECTED passes
    constexpr bool FLAT = WGM == 3;      // ... small meshes: the flat evaluation, one task per WAVEFRONT
    constexpr bool MESH = COOP && !FLAT; // one task per WORKGROUP of WGW wavefronts (mesh_eval_wg), else one per wavefront
    constexpr int WGW = FLAT ? 1 : WGM, EW = FLAT ? MF_FLAT : MF_QUAD;
    __shared__ double s_mesh[FLAT ? SW_WAVES * (64 * 4 + 64 * 5) : (MESH ? mesh_wg_stage(WGW) + MESH_WG_RES : 1)];
    TrajL<LDS> tr{P.T, P.coeffs, P.N};
    if constexpr (LDS) tr = traj_stage(s_dyn, P.T, P.coeffs, P.N);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool writer = lane == 0 && (!MESH || wave == 0);
    const int m_lv = MESH ? P.shape.mesh_levels : 1;       // frames: behind the staged trajectory in the dynamic LDS, one set per wavefront
    MeshWG mwg{s_mesh + wave * 64 * 4, s_mesh + (MESH ? mesh_wg_stage(WGW) : 0), 0, wave, 0,
               MeshFrames{reinterpret_cast<unsigned *>(reinterpret_cast<char *>(s_dyn) + (LDS ? traj_lds_bytes(P.N) : 0)) + (MESH ? wave * mesh_frames_words(m_lv) : 0), m_lv}, WGW, FlatLds{}};
    if constexpr (FLAT) {
        int *blob = reinterpret_cast<int *>(reinterpret_cast<char *>(s_dyn) + (LDS ? ((traj_lds_bytes(P.N) + 15) & ~(size_t)15) : 0));
        flat_lds_stage(P.shape.mesh, blob);
        mwg.stage = s_mesh + wave * (64 * 4 + 64 * 5); mwg.res = mwg.stage + 64 * 4; mwg.wave = 0;
        mwg.flat = flat_lds_view(blob, reinterpret_cast<float *>(blob + P.shape.mesh_flat_words) + wave * flat_om_floats(P.shape.mesh_flat_slots));
    }
    const unsigned cap = (unsigned)(P.point_end - P.point_begin) * SW_MAX_RANGES;
    const unsigned n_front = (unsigned)__builtin_amdgcn_readfirstlane((int)P.words[0]);
    const unsigned n_tasks = n_front + (unsigned)__builtin_amdgcn_readfirstlane((int)P.words[1]);
    const unsigned lpt_thr = KIND == ISDF_SHAPE_MESH ? (unsigned)__builtin_amdgcn_readfirstlane((int)P.words[6]) : 0u;
    const unsigned n_waves = (unsigned)__builtin_amdgcn_readfirstlane((int)(MESH ? n_blocks : n_blocks * SW_WAVES));
    const double D = *P.traj_duration;
    const int gmode = P.shape.grad_mode;
    const bool fd_central = (KIND != ISDF_SHAPE_MESH) && gmode == ISDF_GRAD_CENTRAL;
    const bool fd_forward = (KIND != ISDF_SHAPE_MESH) && gmode == ISDF_GRAD_BOX_FORWARD;
    const bool own_grad = !fd_central && !fd_forward;
    constexpr bool DIRECTED = COOP;                          // workgroup-cooperative mesh descents (see the task loop; the wave-cooperative walks of small meshes: measured slower with it)
    // this lane's role in a pass: block (0..6) and offset inside it (0 = the pose lane)
    const int gd_blk = lane < 18 ? -1 : (lane < 25 ? 0 : (lane < 60 ? (lane - 25) / 7 + 1 : (lane == 60 ? GD_SPEC + 1 : -1)));
    const int gd_off = lane < 18 ? 0 : (lane < 25 ? lane - 18 : (lane < 60 ? (lane - 25) % 7 : 0));
    for (unsigned task = MESH ? block : block * SW_WAVES + wave, rounds = 0; task < n_tasks && rounds < (1u << 16); task += n_waves, rounds++) {
        const unsigned tm = (unsigned)__builtin_amdgcn_readfirstlane((int)(task < n_front ? P.task_map[task] : P.task_map[cap - 1u - (task - n_front)]));
        const unsigned long long dbg_t0 = ((ISDF_SWEEP_STATS && P.dbg) || KIND == ISDF_SHAPE_MESH) ? wall_clock64() : 0ull;
        // (measured: raising the issue priority of the tasks at the front of the list changes nothing - the walks wait for loads, not for issue slots)
        const int rel = (int)(tm >> 5), r = (int)(tm & 31u);
        const int pt = P.point_begin + rel;
        double *const slot = P.task_buf + ((size_t)rel * SW_MAX_RANGES + r) * TASK_STRIDE;
        const d3 p = mk3(P.points[3 * pt], P.points[3 * pt + 1], P.points[3 * pt + 2]);
        const double lb = slot[0], ub = slot[1];
        const double t_min = fmax(0.0, lb), t_max = fmin(ub, D);
        double x = slot[2], fx = 0.0, prev_x = 10000000.0;
        d3 g_keep = mk3(0, 0, 0);                  // the blocks' shape gradients of the last pass (on their pose lanes)
        int g_src = -1;                            // pose lane of the block whose gradient is the one at the current x (-1: none yet)
        int iter = 0, n_pass = 0;
        int s_spec = 0;                            // direction of the look-ahead blocks (0: none)
        bool grad_only = false;                    // final pass: lanes 18-24 only, the gradient at the final x
        // Mesh robots: DIRECTED passes.  A hierarchy query returns the gradient with the value, so every candidate lane can form
        // dSDF/dt at ITS time as well (the pose and the point velocity are computed by the wavefront anyway): the accepted candidate
        // hands the next iterate its gradient and the sign of dSDF/dt - the next pass knows on which side its nine halving candidates
        // lie before it starts.  Lanes of a directed pass: 0-8 candidates x - tau_k * dir; 9-13 poses at y_2 .. y_6 when the last
        // step was a full step in this direction (y_1 IS candidate 0).  At most 14 hierarchy queries = ONE quad round (the regular
        // pass: 25 = two), no query at x itself, and never a gradient-only pass: 10 -> 6 walks for a descent of five passes.  Same
        // evaluations (a candidate's arithmetic at time t is the pose lane's at x = t), same accept tests, same t*.
        // The first pass is directed as well: the scan's fine pass that found the seed evaluated the very same point at the very same
        // time (swept_scan_kernel leaves f and dSDF/dt with the seed) - only a descent that accepts no step at all still needs the
        // gradient at its seed (one query).
        bool dir_known = false, g_valid = false;
        int dir = 0;                               // sign of dSDF/dt at x
        d3 g_x = mk3(0, 0, 0);                     // shape gradient at x (wave-uniform; g_valid)
        if constexpr (DIRECTED) {
            const double gd0 = slot[4];
            if (gd0 == gd0) { fx = slot[3]; dir = (int)(gd0 > 0) - (int)(gd0 < 0); dir_known = true; }
        }
        for (int guard = 0; guard < 4096; guard++) {
            if (DIRECTED && dir_known) {
                if (dir == 0) {                                           // no candidate set (:1040): the descent stops here
                    iter += 9;
                    if (g_valid) break;
                    grad_only = true; dir_known = false; continue;
                }
                const bool ahead = s_spec != 0 && s_spec == dir;
                double te = x;
                const bool act = lane < 9 || (ahead && lane < 14);
                if (lane < 9) {
                    double tau = 0.02;
                    for (int q = 0; q < 8; q++) if (q < lane) tau = 0.5 * tau;
                    te = x - tau * (dir > 0 ? 1 : -1);
                    te = fmax(fmin(te, t_max), t_min);
                } else if (act) {
                    for (int st = 0; st < GD_SPEC + 1; st++) if (st < lane - 7) te = fmax(fmin(te - 0.02 * dir, t_max), t_min);      // lane 7 + m: y_m
                }
                double val = 1e300;
                d3 pvel = mk3(0, 0, 0), gq = mk3(0, 0, 0);
                if constexpr (COOP) eval_lane<KIND, LDS, EW>(P, tr, p, te, 0, true, true, val, pvel, gq, &mwg, act, fx);
                else if (act) eval_lane<KIND, LDS>(P, tr, p, te, 0, true, true, val, pvel, gq, nullptr, true, fx);
                n_pass++;
                const double gdot = dot3(gq, pvel);
                prev_x = x;
                const unsigned long long accm = __ballot(lane < 9 && ((val - fx) < 0));
                if (!accm) {                                              // no step decreases the SDF
                    iter += 9;
                    if (g_valid) break;
                    grad_only = true; dir_known = false; continue;        // (stopped at its seed: the gradient there)
                }
                int src = __ffsll((long long)accm) - 1;                   // first (largest) step that does
                iter += src + 1;
                x = rlane(te, src);
                fx = rlane(val, src);
                bool full = src == 0, done = false;
                s_spec = full ? dir : 0;
                if (!(iter < 300 && fabs(x - prev_x) > 1e-5)) done = true;
                // marching: while the step at y_m is again the full step in this direction (y_1 on lane 0, y_m on lane 7 + m)
                for (int m = 1; !done && full && ahead && m <= GD_SPEC; m++) {
                    const double gm = rlane(gdot, src);
                    if (((int)(gm > 0) - (int)(gm < 0)) != dir) { s_spec = 0; break; }    // direction changes at y_m: a pass without look-ahead from there
                    const int nb = 8 + m;                                 // y_(m+1)
                    const double fc = rlane(val, nb);
                    if (!((fc - fx) < 0)) { s_spec = 0; break; }          // full step rejected: its halvings are the next pass
                    prev_x = x;
                    iter += 1;
                    x = rlane(te, nb);
                    fx = fc;
                    src = nb;
                    if (!(iter < 300 && fabs(x - prev_x) > 1e-5)) done = true;
                }
                const double gs = rlane(gdot, src);
                dir = (int)(gs > 0) - (int)(gs < 0);
                g_x = mk3(rlane(gq.x, src), rlane(gq.y, src), rlane(gq.z, src));
                g_valid = true;
                if (done) break;
                continue;
            }
            // ---- what this lane evaluates
            double te = 0.0;
            int pert = 0;                 // 0: the body-frame point itself; 1..6: central-difference sample; 7..9: forward difference
            bool act = false, pose_lane = false;
            if (lane < 18) {
                if (!grad_only) {
                    double tau = 0.02;
                    const int k = lane < 9 ? lane : lane - 9;
                    for (int q = 0; q < 8; q++) if (q < k) tau = 0.5 * tau;
                    te = x - tau * (lane < 9 ? 1 : -1);
                    te = fmax(fmin(te, t_max), t_min);
                    act = true;
                    pose_lane = DIRECTED;          // (directed passes: a candidate forms dSDF/dt at its own time as well)
                }
            } else if (gd_blk == 0 || (gd_blk > 0 && !grad_only && s_spec != 0)) {
                // y_blk: gd_blk full steps from x in the direction s_spec, clamped like every candidate (:1046-1047)
                te = x;
                for (int q = 0; q < GD_SPEC + 1; q++) if (q < gd_blk) te = fmax(fmin(te - 0.02 * s_spec, t_max), t_min);
                if (gd_off == 0) { act = true; pose_lane = gd_blk <= GD_SPEC; }
                else if (fd_central) { pert = gd_off; act = true; }
                else if (fd_forward && gd_off <= 3) { pert = gd_off + 6; act = true; }
            }
            double val = 1e300;
            d3 pvel = mk3(0, 0, 0), g_direct = mk3(0, 0, 0);
            // (mesh robots: every test below asks "val < f(x)?" of the lanes other than 18 - the halving candidates directly, the look-ahead
            // blocks through a chain of strictly decreasing values that starts at f(x) - and uses a value only when the answer is yes: f(x),
            // once known, is those queries' cutoff in the hierarchy)
            if constexpr (COOP) eval_lane<KIND, LDS, EW>(P, tr, p, act ? te : x, pert, pose_lane, own_grad, val, pvel, g_direct, &mwg, act,
                                                           (iter > 0 && lane != 18) ? fx : __builtin_inf());
            else if (act) eval_lane<KIND, LDS>(P, tr, p, te, pert, pose_lane, own_grad, val, pvel, g_direct);
            n_pass++;
            // Shape gradient (getonlyGrad1) and dSDF/dt = grad . (body-frame point velocity) of EVERY block at once: the pose lane
            // of a block pulls the six samples to its right across the wavefront (wave_shl DPP: no LDS), forms the gradient with
            // the reference's own operations and its dot product with the point velocity it holds - the serial walk over the
            // blocks below then reads ONE sign per block instead of rebuilding each block's gradient on all lanes
            d3 gq;
            if (fd_central || fd_forward) {
                const double v1 = wave_shl1(val), v2 = wave_shl1(v1), v3 = wave_shl1(v2);
                if (fd_central) {
                    const double v4 = wave_shl1(v3), v5 = wave_shl1(v4), v6 = wave_shl1(v5);
                    const double dx = 0.000005;
                    gq = normalized3(mk3((v2 - v1) / (2 * dx), (v4 - v3) / (2 * dx), (v6 - v5) / (2 * dx)));
                } else {
                    const double dx = 0.01;
                    gq = mk3((v1 - val) / dx, (v2 - val) / dx, (v3 - val) / dx);
                }
            } else gq = g_direct;
            const double gdot = dot3(gq, pvel);
            g_src = 18;                            // the gradient at the current x sits on lane 18
            if (grad_only) { g_keep = gq; break; }
            // sign-descent iterations: the one at x in full, then - while every step is the full step in the direction the
            // look-ahead blocks were laid out for - the ones at y_1, y_2, ...
            if (iter == 0) fx = rlane(val, 18);
            bool finished = false;
            int spec_next = 0;
            for (int m = 0;; m++) {
                const int base = gd_block_base(m);
                if (m > 0) g_src = base;
                const double g = rlane(gdot, base);
                const int sgn = (int)(g > 0) - (int)(g < 0);
                spec_next = 0;
                if (m == 0) {
                    prev_x = x;
                    const bool in_set = sgn > 0 ? (lane < 9) : (sgn < 0 ? (lane >= 9 && lane < 18) : false);
                    const unsigned long long accm = __ballot(in_set && ((val - fx) < 0));
                    bool stop = false, full = false;
                    if (accm) {
                        const int src = __ffsll((long long)accm) - 1;            // first (largest) step that decreases the SDF
                        const int d = sgn > 0 ? src : src - 9;
                        iter += d + 1;
                        x = rlane(te, src);
                        fx = rlane(val, src);
                        g_src = -1;
                        full = d == 0;
                        if (full) spec_next = sgn;
                        if constexpr (DIRECTED) {
                            const double gs = rlane(gdot, src);
                            dir = (int)(gs > 0) - (int)(gs < 0); dir_known = true;
                            g_x = mk3(rlane(gq.x, src), rlane(gq.y, src), rlane(gq.z, src));
                            g_valid = true;
                        }
                    } else {
                        iter += 9;
                        stop = true;
                    }
                    if (!(iter < 300 && !stop && fabs(x - prev_x) > 1e-5)) { finished = true; break; }
                    if (!(full && s_spec != 0 && sgn == s_spec)) break;          // x is not y_1: the next pass starts from it
                } else {
                    // at x = y_m only the full step along s_spec has been evaluated: f(y_(m+1))
                    if (sgn != s_spec) break;                                    // direction changed: regular pass from here
                    const int nb = m < GD_SPEC ? gd_block_base(m + 1) : 60;
                    const double fc = rlane(val, nb);
                    if (!((fc - fx) < 0)) break;                                 // full step rejected: its halvings need a regular pass
                    prev_x = x;
                    iter += 1;
                    x = rlane(te, nb);
                    fx = fc;
                    g_src = -1;
                    spec_next = s_spec;
                    if (!(iter < 300 && fabs(x - prev_x) > 1e-5)) { finished = true; break; }
                    if (m == GD_SPEC) break;
                }
            }
            g_keep = gq;
            s_spec = spec_next;
            if (finished) {
                if (g_src >= 0 || (DIRECTED && g_valid)) break;
                grad_only = true;          // the last step moved x: one more pass for the gradient there
            }
        }
        // the gradient at the final x: on the pose lane of the block that evaluated there (lane 18 after a grad_only pass)
        const int gl = g_src >= 0 ? g_src : 18;
        d3 g1_at_x = mk3(rlane(g_keep.x, gl), rlane(g_keep.y, gl), rlane(g_keep.z, gl));
        if (DIRECTED && g_valid) g1_at_x = g_x;
        // interval result: (t_r, sdf_r, gradient at t_r) - the best interval is picked in order by the back-prop kernel (:739-744)
        if (writer) {
            slot[0] = x; slot[1] = fx; slot[2] = g1_at_x.x; slot[3] = g1_at_x.y; slot[4] = g1_at_x.z;
            slot[5] = __longlong_as_double((long long)(((unsigned long long)n_pass & 0xFFFFull) | ((unsigned long long)iter << 16)));
            // next step: dispatch this interval's descent early.  Analytic robots: by its pass count (a pass costs the same everywhere);
            // mesh robots: by its DURATION (a pass next to the surface costs ten times one far from it; 100 MHz clock)
            bool long_task = n_pass >= LPT_TASK;
            if constexpr (KIND == ISDF_SHAPE_MESH) {
                const unsigned dur = (unsigned)min((wall_clock64() - dbg_t0) >> 7, 0xFFFFFull);      // 1.28 us units
                // (32-bit sum of 1.28 us units: saturating - 2^32 units are 90 minutes of descents per step - and only ever a dispatch hint)
                if (atomicAdd(&P.words[4], dur) > 0xFFFFFFFFu - dur) atomicMax(&P.words[4], 0xFFFFFFFFu);
                atomicAdd(&P.words[5], 1u);
                if (lpt_thr != 0u) long_task = dur >= lpt_thr;                   // (first step: the pass-count rule)
            }
            if (long_task) atomicOr(&P.point_lmask[pt], 1u << r);
            if (ISDF_SWEEP_STATS && P.dbg && task < (unsigned)P.M) {      // developer timing (ISDF_DEBUG_TIMING=1): start, end, passes of this task
                unsigned long long *d = P.dbg + (size_t)P.M * 4 + (size_t)(P.N + 1) * 8 + (size_t)task * 3;
                d[0] = dbg_t0; d[1] = wall_clock64(); d[2] = (unsigned long long)n_pass | ((unsigned long long)(task < n_front) << 32) | ((unsigned long long)(iter & 0xFFFF) << 40);
            }
        }
    }
}

template <int KIND, bool LDS, int WGM = 0>
__global__ __launch_bounds__(64 * SW_WAVES, WGM ? ISDF_MESH_WPE : 1) void swept_descent_kernel(const SweptParams P) {
    extern __shared__ double s_dyn[];
    descent_body<KIND, LDS, WGM>(P, s_dyn, blockIdx.x, gridDim.x);
}
// The minimisers are GIVEN (isdf_eval_swept_at_tstar): one thread per obstacle point evaluates the robot SDF and its
// body-frame gradient at tstar[pt] (getSDFAtTimeStamp :550-556, getGradPrelAtTimeStamp :566-572) and leaves the same record
// the sweep kernel leaves; a negative or NaN tstar means "no interval qualified" (min sdf 10, no penalty; :717).
template <int KIND>
__global__ __launch_bounds__(64) void swept_fixed_kernel(const SweptParams P, const double *tstar_in) {
    const int pt = P.point_begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x == 0) for (int k = 0; k < 8; k++) if (k != 4) P.stats[k] = 0ull;
    if (pt >= P.point_end) return;
    const TrajL<false> tr{P.T, P.coeffs, P.N};
    const d3 p = mk3(P.points[3 * pt], P.points[3 * pt + 1], P.points[3 * pt + 2]);
    const double ts = tstar_in[pt];
    double *o = P.point_partial + (size_t)pt * PARTIAL_STRIDE;
    o[5] = 0.0;
    if (!(ts >= 0.0)) { o[0] = P.tstar[pt]; o[1] = 1e1; o[2] = o[3] = o[4] = 0.0; return; }
    d3 x; m3 R;
    pose_at(P, tr, ts, x, R);
    const d3 pr = mat_tmul(R, p - x);
    d3 g;
    double sdf;
    if constexpr (KIND == ISDF_SHAPE_MESH) sdf = mesh_sdf_grad(P.shape.mesh, pr, g);
    else { sdf = shape_sdf<KIND>(P.shape, pr); g = shape_grad<KIND>(P.shape, pr); }
    o[0] = ts; o[1] = sdf; o[2] = g.x; o[3] = g.y; o[4] = g.z;
}

// Back-prop at t* (back_end_optimizer.hpp:578-646): one THREAD per obstacle point - penalty (mu = 0.01), pose Jacobians,
// flatness reverse mode, the 18 + 1 + 1 doubles the reduction sums.
__device__ __forceinline__ void backprop_point(const SweptParams &P, const int pt) {
    const TrajView tr{P.T, P.coeffs, P.N};
    double *o = P.point_partial + (size_t)pt * PARTIAL_STRIDE;
    const d3 p = mk3(P.points[3 * pt], P.points[3 * pt + 1], P.points[3 * pt + 2]);
    double tstar, min_sdf_star;
    d3 grad_prel;
    unsigned long long counters;
    if (P.direct_records) {           // the fixed-t* kernel left (t*, sdf*, gradient) in the point's record
        tstar = o[0]; min_sdf_star = o[1];
        grad_prel = mk3(o[2], o[3], o[4]);
        counters = (unsigned long long)__double_as_longlong(o[5]);
    } else {
        // the best interval, in interval order with a strict '<' (getSDFofSweptVolume :729-745); no interval: lastTstar stays
        // and the SDF reads 10 (:717, :576-578)
        const int nrw = P.point_nr[pt], nr = nrw & 0xFF;
        const double *slots = P.task_buf + (size_t)(pt - P.point_begin) * SW_MAX_RANGES * TASK_STRIDE;
        tstar = P.tstar[pt]; min_sdf_star = 1e1;
        grad_prel = mk3(0, 0, 0);
        unsigned long long passes = (unsigned long long)(nrw >> 8), iters = 0;
        for (int r = 0; r < nr; r++) {
            const double *sl = slots + (size_t)r * TASK_STRIDE;
            const double f_r = sl[1];
            if (f_r < min_sdf_star) { min_sdf_star = f_r; tstar = sl[0]; grad_prel = mk3(sl[2], sl[3], sl[4]); }
            const unsigned long long c = (unsigned long long)__double_as_longlong(sl[5]);
            passes += c & 0xFFFFull; iters += c >> 16;
        }
        counters = ((unsigned long long)nr & 0xFFull) | ((passes & 0xFFFFull) << 8) | (iters << 24);
    }
    double time_local = tstar;
    const int i = traj_locate(tr, time_local);
    Basis B;
    poly_basis(time_local, B);
    const double *c = P.coeffs + 6 * i;
    const int ld = 6 * P.N;
    const d3 pos = ct_beta(c, ld, B.b0), vel = ct_beta<1>(c, ld, B.b1), acc3 = ct_beta<2>(c, ld, B.b2), jer = ct_beta<3>(c, ld, B.b3),
             sna = ct_beta<4>(c, ld, B.b4);
    FlatS fs;
    flat_core(P.flat, vel, acc3, fs);
    const d4 quat = flat_quat(fs);
    const m3 rot = quat_to_rot(quat);
    double pena = 0.0;
    d3 gradPos = mk3(0, 0, 0), gradVel = gradPos, gradOmg = gradPos;
    d4 gradQuat; gradQuat.w = gradQuat.x = gradQuat.y = gradQuat.z = 0.0;
    double sdf_cost = -1.0, sdf_out_grad = 0.0;
    smoothed_l1(P.safety_hor - min_sdf_star, 0.01, sdf_cost, sdf_out_grad);    // grad_cost_p_sw, mu hard-coded (:851)
    bool active = false;
    if (sdf_cost > 2.2204460492503131e-16) {                                    // costp > DBL_EPSILON (:865)
        active = true;
        const d3 wg = mat_mul(rot, grad_prel);
        gradPos = mk3(P.weight_p * (sdf_out_grad * wg.x), P.weight_p * (sdf_out_grad * wg.y), P.weight_p * (sdf_out_grad * wg.z));
        const d4 sq = quat_step_grad(quat, grad_prel, p - pos);
        gradQuat.w = P.weight_p * (-sdf_out_grad * sq.w); gradQuat.x = P.weight_p * (-sdf_out_grad * sq.x);
        gradQuat.y = P.weight_p * (-sdf_out_grad * sq.y); gradQuat.z = P.weight_p * (-sdf_out_grad * sq.z);
        pena = P.weight_p * sdf_cost;
    }
    d3 gP, gV, gA, gJ;
    flat_backward(P.flat, vel, acc3, jer, gradPos, gradVel, gradQuat, gradOmg, gP, gV, gA, gJ);
    const double gPv[3] = {gP.x, gP.y, gP.z}, gVv[3] = {gV.x, gV.y, gV.z}, gAv[3] = {gA.x, gA.y, gA.z}, gJv[3] = {gJ.x, gJ.y, gJ.z};
#pragma unroll
    for (int d = 0; d < 3; d++)
#pragma unroll
        for (int rr = 0; rr < 6; rr++)
            o[d * 6 + rr] = B.b0[rr] * gPv[d] + B.b1[rr] * gVv[d] + B.b2[rr] * gAv[d] + B.b3[rr] * gJv[d];
    o[18] = (-dot3(gP, vel) + -dot3(gV, acc3) + -dot3(gA, jer) + -dot3(gJ, sna));
    o[19] = pena;
    // a point without an active penalty contributes exact zeros (more than half of the points, all of them in piece 0 when
    // their lastTstar is still 0): the per-piece sums skip it (x + 0.0 == x)
    bool all_zero = !active;
#pragma unroll
    for (int q = 0; q < 19; q++) all_zero = all_zero && (o[q] == 0.0);
    P.point_piece[pt] = all_zero ? -1 : i;
    P.point_stat[pt] = (counters << 1) | (active ? 1ull : 0ull);
    P.tstar[pt] = tstar;
}
__device__ __forceinline__ void backprop_body(const SweptParams &P, const int pt) {
    // Analytic robots: the NEXT step's dispatch order of the scan (this kernel is the step's thread-per-point pass that runs after
    // the scan).  Two classes - a point whose scan made 3 or more fine passes is heavy (17-33 us against 5-13) - as a two-ended list:
    // heavy points fill it from the front, the others from the back; one reservation per wavefront and class (words[8], words[9],
    // emptied by the prepare kernel), requested here and used after the point's own work (their round trip hides behind it).  In
    // point order the scan launch ended on wavefronts that had started 19 us late (profiles/r4_v1_timing_cone.txt).
    const bool build = P.scan_order_out && !P.scan_sort_here;
    const int lane = (int)__lane_id();
    const bool ok = pt < P.point_end;
    bool heavy = false;
    unsigned long long mh = 0ull, ml = 0ull;
    unsigned bh = 0u, bl = 0u;
    if (build) {
        heavy = ok && P.scan_ticks[ok ? pt - P.point_begin : 0] >= 3u;
        mh = __ballot(heavy); ml = __ballot(ok && !heavy);
        if (lane == 0) { if (mh) bh = atomicAdd(&P.words[8], (unsigned)__popcll(mh)); if (ml) bl = atomicAdd(&P.words[9], (unsigned)__popcll(ml)); }
    }
    if (ok) backprop_point(P, pt);
    if (build) {
        const unsigned long long lt = (1ull << lane) - 1ull;
        bh = (unsigned)__builtin_amdgcn_readfirstlane((int)bh); bl = (unsigned)__builtin_amdgcn_readfirstlane((int)bl);
        if (heavy) P.scan_order_out[bh + (unsigned)__popcll(mh & lt)] = pt - P.point_begin;
        else if (ok) P.scan_order_out[(unsigned)(P.point_end - P.point_begin) - 1u - (bl + (unsigned)__popcll(ml & lt))] = pt - P.point_begin;
    }
}

// Deterministic reduction.  Blocks [0, N): gradC rows of piece i and H[i] = sum of gdT over the points whose t* lies in
// piece i; block N: cost and the statistics words.  Every thread adds its strided points in ascending order (the loads of a
// round of RED_U points are issued together - a dependent load per point made this kernel 36 us), the 20 columns then go
// through ONE fixed-shape butterfly per wavefront and the four wave sums are added in wave order.  The workgroup that finishes
// LAST (ticket in stats[6], zeroed by the prepare / fixed kernel of the step) turns H into
// gradT(j) = sum_{i > j} H[i]   ("for (j < i) gradT(j) += gdT", back_end_optimizer.hpp:642-645), in fixed order.
constexpr int RED_U = 8;
// wave-wide sum with DPP lane exchanges (fixed shape: quads, half rows, rows, then the four row totals in row order)
template <int CTRL> __device__ __forceinline__ double dpp_f64(double v) {
    // (every lane of these patterns has a source: the move needs no "old" operand - with one, the value is copied first)
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
    v += dpp_f64<0xB1>(v);        // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);        // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);       // row_half_mirror
    v += dpp_f64<0x140>(v);       // row_mirror: every lane of a row of 16 holds the row total
    return ((rlane(v, 0) + rlane(v, 16)) + rlane(v, 32)) + rlane(v, 48);
}
__device__ __forceinline__ void reduce_body(const SweptParams &P, double *out, double *hist, const int i) {
    __shared__ double s_red[4][20];
    __shared__ unsigned long long s_cnt[4][4];
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double acc[20];
#pragma unroll
    for (int q = 0; q < 20; q++) acc[q] = 0.0;
    unsigned long long n_act = 0, n_rng = 0, n_pass = 0, n_it = 0;
    const bool cost_block = (i == P.N);
    unsigned long long *const rdbg = (ISDF_SWEEP_STATS && P.dbg) ? P.dbg + (size_t)P.M * 4 + (size_t)i * 8 : nullptr;
    if (rdbg && tid == 0) rdbg[0] = wall_clock64();
    for (int base = P.point_begin + tid; base < P.point_end; base += 256 * RED_U) {
        if (cost_block) {
            double pena[RED_U]; unsigned long long st[RED_U];
#pragma unroll
            for (int u = 0; u < RED_U; u++) {
                const int pt = base + u * 256;
                const bool ok = pt < P.point_end;
                pena[u] = ok ? P.point_partial[(size_t)pt * PARTIAL_STRIDE + 19] : 0.0;
                st[u] = ok ? P.point_stat[pt] : 0ull;
            }
#pragma unroll
            for (int u = 0; u < RED_U; u++) {
                if (base + u * 256 < P.point_end) acc[19] += pena[u];
                n_act += st[u] & 1ull; n_rng += (st[u] >> 1) & 0xFFull; n_pass += (st[u] >> 9) & 0xFFFFull; n_it += st[u] >> 25;
            }
        } else {
            int pid[RED_U];
#pragma unroll
            for (int u = 0; u < RED_U; u++) { const int pt = base + u * 256; pid[u] = pt < P.point_end ? P.point_piece[pt] : -1; }
#pragma unroll
            for (int u = 0; u < RED_U; u++)
                if (pid[u] == i) {
                    const double *o = P.point_partial + (size_t)(base + u * 256) * PARTIAL_STRIDE;
#pragma unroll
                    for (int q = 0; q < 19; q++) acc[q] += o[q];
                }
        }
    }
    if (rdbg && tid == 0) rdbg[1] = wall_clock64();
#pragma unroll
    for (int q = 0; q < 20; q++) {
        if (cost_block != (q == 19)) continue;     // wave-uniform
        const double v = wave_sum_dpp(acc[q]);
        if (lane == 0) s_red[wave][q] = v;
    }
    if (cost_block) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            n_act += __shfl_xor(n_act, off, 64); n_rng += __shfl_xor(n_rng, off, 64);
            n_pass += __shfl_xor(n_pass, off, 64); n_it += __shfl_xor(n_it, off, 64);
        }
        if (lane == 0) { s_cnt[wave][0] = n_act; s_cnt[wave][1] = n_rng; s_cnt[wave][2] = n_pass; s_cnt[wave][3] = n_it; }
    }
    __syncthreads();
    if (rdbg && tid == 0) rdbg[2] = wall_clock64();
    if (tid < 20) {
        const int q = tid;
        const double v = ((s_red[0][q] + s_red[1][q]) + s_red[2][q]) + s_red[3][q];
        if (cost_block) { if (q == 19) out[0] = v; }
        else if (q == 18) __hip_atomic_store(&hist[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (q < 18) { const int d = q / 6, r = q - 6 * d; out[1 + P.N + (size_t)d * 6 * P.N + 6 * i + r] = v; }
    }
    if (cost_block && tid == 0) {                  // single writer: no atomics on the statistics words
        P.stats[0] = (unsigned long long)(P.point_end - P.point_begin);
        P.stats[3] = s_cnt[0][0] + s_cnt[1][0] + s_cnt[2][0] + s_cnt[3][0];
        P.stats[2] = s_cnt[0][1] + s_cnt[1][1] + s_cnt[2][1] + s_cnt[3][1];
        P.stats[7] = s_cnt[0][2] + s_cnt[1][2] + s_cnt[2][2] + s_cnt[3][2];
        P.stats[5] = s_cnt[0][3] + s_cnt[1][3] + s_cnt[2][3] + s_cnt[3][3];
    }
    // last workgroup of the N piece blocks: suffix sums
    __syncthreads();
    if (rdbg && tid == 0) rdbg[3] = wall_clock64();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned long long t = cost_block ? 0ull : atomicAdd(&P.stats[6], 1ull);
        s_last = (!cost_block && t == (unsigned long long)(P.N - 1)) ? 1 : 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (rdbg && tid == 0) rdbg[4] = wall_clock64();
    if (s_last) {
        // H is fetched by all threads at once (one agent-scope load each: 40 of them one after the other on a single lane
        // were most of this kernel's 32 us), the running sum itself stays serial and in fixed order
        __shared__ double s_h[256];
        double sfx = 0.0;
        for (int top = P.N; top > 0; top -= 256) {
            const int j = top - 1 - tid;                       // thread 0 holds the highest piece of this round
            if (j >= 0) s_h[tid] = __hip_atomic_load(&hist[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (tid == 0) {
                const int n = min(256, top);
                for (int k = 0; k < n; k++) { out[1 + (top - 1 - k)] = sfx; sfx += s_h[k]; }
            }
            __syncthreads();
        }
    }
    if (rdbg && tid == 0) rdbg[5] = wall_clock64();
}

// (Round 4, measured and dropped: back-prop and sums as ONE launch, the reduction workgroups waiting behind the back-prop workgroups
// for a counter - 29.0 us against 10.3 + 12.7 us for the two launches, whose gap on this system is ~1.5 us.)
__global__ __launch_bounds__(128) void swept_backprop_kernel(const SweptParams P) {
    backprop_body(P, P.point_begin + (int)(blockIdx.x * blockDim.x + threadIdx.x));
}
__global__ __launch_bounds__(256) void swept_reduce_kernel(const SweptParams P, double *out, double *hist) {
    reduce_body(P, out, hist, (int)blockIdx.x);
}

void launch_swept_prepare(const SweptParams &P, hipStream_t stream) {
    const size_t lds = traj_lds_bytes(P.N);
    if (lds <= TRAJ_LDS_MAX) hipLaunchKernelGGL(swept_prepare_kernel<true>, dim3(1), dim3(1024), lds, stream, P);
    else hipLaunchKernelGGL(swept_prepare_kernel<false>, dim3(1), dim3(1024), 0, stream, P);
}
void launch_swept_sweep(const SweptParams &P, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop) {
    const int n = P.point_end - P.point_begin;
    if (n <= 0) return;
    // (larger mesh robots: one point / task per workgroup)
    const bool mesh_wg = P.shape.kind == ISDF_SHAPE_MESH && P.shape.mesh_wg && !P.shape.mesh_flat;
    const dim3 grid(mesh_wg ? n : (n + SW_WAVES - 1) / SW_WAVES), block(64 * SW_WAVES);
    const size_t lds = traj_lds_bytes(P.N);
    const bool in_lds = lds <= TRAJ_LDS_MAX;
    // scan, then descent (one wavefront per task while there are no more tasks than points, which is the rule); the start
    // event rides on the first dispatch, the stop event on the second
#define ISDF_SWEPT_CASE(K) case K: if (in_lds) { hipExtLaunchKernelGGL((swept_scan_kernel<K, true>), grid, block, lds, stream, ev_start, nullptr, 0, P); \
                                                 hipExtLaunchKernelGGL((swept_descent_kernel<K, true>), grid, block, lds, stream, nullptr, ev_stop, 0, P); } \
                                   else { hipExtLaunchKernelGGL((swept_scan_kernel<K, false>), grid, block, 0, stream, ev_start, nullptr, 0, P); \
                                          hipExtLaunchKernelGGL((swept_descent_kernel<K, false>), grid, block, 0, stream, nullptr, ev_stop, 0, P); } break;
    if (P.shape.kind == ISDF_SHAPE_MESH && P.shape.mesh_flat) {
        // small meshes: one point / task per WAVEFRONT (four to a workgroup, like the analytic kinds), the flat evaluation; dynamic LDS = the
        // mesh's table blob (copied once per workgroup) + per wavefront its [16][slots] values; the trajectory stays in global memory
        const size_t lds_flat = (size_t)P.shape.mesh_flat_words * sizeof(int) + (size_t)SW_WAVES * flat_om_floats(P.shape.mesh_flat_slots) * sizeof(float);
        const size_t lds_traj = (traj_lds_bytes(P.N) + 15) & ~(size_t)15;
        // (the trajectory in LDS as well while it leaves room for three workgroups per CU; ISDF_FLAT_TRAJ_LDS=0 / 1 forces)
        static const int tl_env = []{ const char *e = getenv("ISDF_FLAT_TRAJ_LDS"); return e ? atoi(e) : -1; }();
        const bool traj_in_lds = tl_env >= 0 ? tl_env != 0 : lds_traj <= 12 * 1024;
        if (traj_in_lds && lds_traj + lds_flat <= 40 * 1024) {
            hipExtLaunchKernelGGL((swept_scan_kernel<ISDF_SHAPE_MESH, true, 3>), grid, block, lds_traj + lds_flat, stream, ev_start, nullptr, 0, P);
            hipExtLaunchKernelGGL((swept_descent_kernel<ISDF_SHAPE_MESH, true, 3>), grid, block, lds_traj + lds_flat, stream, nullptr, ev_stop, 0, P);
        } else {
            hipExtLaunchKernelGGL((swept_scan_kernel<ISDF_SHAPE_MESH, false, 3>), grid, block, lds_flat, stream, ev_start, nullptr, 0, P);
            hipExtLaunchKernelGGL((swept_descent_kernel<ISDF_SHAPE_MESH, false, 3>), grid, block, lds_flat, stream, nullptr, ev_stop, 0, P);
        }
        return;
    }
    if (mesh_wg) {
        // (Round 6, measured and dropped: a TREE-SPLIT form for launches that do not fill the device - a rank's eighth of the points -
        // with four wavefronts per task, one child of the hierarchy's root each, parts combined in child order (bit-exact).  The queries
        // of a pass are neighbours: what they open lies in ONE child of the root, so one wavefront still walks the whole chain while the
        // others find their subtree's nearest triangle for nothing: rank step of the 720-face blob 2.29 -> 2.22 ms, Trefoil.obj 2.48 ->
        // 3.27, 2.18 -> 3.10 (profiles/r6_c5_shard_ranks.txt).)
        // scan: two wavefronts per point (a fine pass has up to 64 queries: two quad rounds each)
        const dim3 block2(128);
        const size_t fr2 = (size_t)2 * mesh_frames_words(P.shape.mesh_levels) * sizeof(unsigned);      // two wavefronts' frames
        // (the trajectory stays in global memory for these launches: a pass is dominated by the hierarchy walks, and the 6 KB of staged
        // coefficients per workgroup were a workgroup less per CU - Trefoil.obj 9.0 -> 8.1 ms, 720 faces 4.75 -> 4.49 ms)
        // With the distance lattice a fine pass rarely has more than 16 queries left: the second wavefront idles.  A launch of several
        // generations of workgroups (>= 3 x the 4 096 one-wavefront workgroups the device holds) is bound by what its wavefronts
        // issue and runs one-wavefront workgroups (16 611 points: blob 5.75 -> 5.31 ms, Trefoil.obj 4.81 -> 4.51); a smaller one is
        // bound by its longest points and keeps two (7 773 points: 2.73 against 3.00 ms; a rank's eighth of the 16 611: 3.09 / 3.48).
        // ISDF_MESH_SCAN_WAVES = 1 / 2 forces a form.
        static const int sw_env = []{ const char *e = getenv("ISDF_MESH_SCAN_WAVES"); return e ? atoi(e) : 0; }();
        const int sw = (sw_env == 1 || sw_env == 2) ? sw_env : (n >= 3 * 4096 ? 1 : 2);
        if (sw == 1) hipExtLaunchKernelGGL((swept_scan_kernel<ISDF_SHAPE_MESH, false, 1>), grid, dim3(64), fr2 / 2, stream, ev_start, nullptr, 0, P);
        else
        hipExtLaunchKernelGGL((swept_scan_kernel<ISDF_SHAPE_MESH, false, 2>), grid, block2, fr2, stream, ev_start, nullptr, 0, P);
        // The descent's passes are directed (descent_body): at most 14 queries = one quad round on ONE wavefront - one-wavefront
        // workgroups, twice the tasks in flight.  Measured against two wavefronts per task (the second one idle but for a first pass
        // without a usable seed): 16 611 points, blob 9.63 -> 7.78 ms, Trefoil.obj 6.62 -> 5.68; an eighth of them (a rank's shard of
        // an 8-GPU step) 1.98 -> 1.94 / 2.45 -> 2.44; bench map, 7 773 points, 3.71 -> 3.65 ms.
        // (a launch of less than a generation: workgroups for half as many tasks again as points - the excess ones find no task and leave)
        hipExtLaunchKernelGGL((swept_descent_kernel<ISDF_SHAPE_MESH, false, 1>), n < 4096 ? dim3(n + n / 2 + 16) : grid, dim3(64), fr2 / 2, stream, nullptr, ev_stop, 0, P);
        return;
    }
    switch (P.shape.kind) {
    // one instantiation per shape class (each inlines exactly one SDF formula; the run-time switch of the generic kernel costs
    // 272 VGPRs = one wave per SIMD)
    ISDF_SWEPT_CASE(ISDF_SHAPE_TORUS) ISDF_SWEPT_CASE(ISDF_SHAPE_CAPPEDTORUS) ISDF_SWEPT_CASE(ISDF_SHAPE_CAPPEDCONE)
    ISDF_SWEPT_CASE(ISDF_SHAPE_ROUNDEDCONE) ISDF_SWEPT_CASE(ISDF_SHAPE_WIREFRAMEBOX) ISDF_SWEPT_CASE(ISDF_SHAPE_BENDLINEAR)
    ISDF_SWEPT_CASE(ISDF_SHAPE_TWISTBOX) ISDF_SWEPT_CASE(ISDF_SHAPE_BENDBOX) ISDF_SWEPT_CASE(ISDF_SHAPE_TABLE)
    ISDF_SWEPT_CASE(ISDF_SHAPE_TREFOIL) ISDF_SWEPT_CASE(ISDF_SHAPE_SMOOTHDIFFERENCE) ISDF_SWEPT_CASE(ISDF_SHAPE_SMOOTHINTERSECTION)
    ISDF_SWEPT_CASE(ISDF_SHAPE_CSG) ISDF_SWEPT_CASE(ISDF_SHAPE_BOX) ISDF_SWEPT_CASE(ISDF_SHAPE_MESH)
    default: ISDF_SWEPT_CASE(-1)      // Ball
    }
#undef ISDF_SWEPT_CASE
}
// the minimisers are given: replaces prepare + sweep (isdf_eval_swept_at_tstar)
void launch_swept_fixed(const SweptParams &P, const double *tstar_in, hipStream_t stream) {
    const int n = P.point_end - P.point_begin;
    const dim3 grid((unsigned)((max(n, 1) + 63) / 64)), block(64);
#define ISDF_FIXED_CASE(K) case K: hipLaunchKernelGGL(swept_fixed_kernel<K>, grid, block, 0, stream, P, tstar_in); break;
    switch (P.shape.kind) {
    ISDF_FIXED_CASE(ISDF_SHAPE_TORUS) ISDF_FIXED_CASE(ISDF_SHAPE_CAPPEDTORUS) ISDF_FIXED_CASE(ISDF_SHAPE_CAPPEDCONE)
    ISDF_FIXED_CASE(ISDF_SHAPE_ROUNDEDCONE) ISDF_FIXED_CASE(ISDF_SHAPE_WIREFRAMEBOX) ISDF_FIXED_CASE(ISDF_SHAPE_BENDLINEAR)
    ISDF_FIXED_CASE(ISDF_SHAPE_TWISTBOX) ISDF_FIXED_CASE(ISDF_SHAPE_BENDBOX) ISDF_FIXED_CASE(ISDF_SHAPE_TABLE)
    ISDF_FIXED_CASE(ISDF_SHAPE_TREFOIL) ISDF_FIXED_CASE(ISDF_SHAPE_SMOOTHDIFFERENCE) ISDF_FIXED_CASE(ISDF_SHAPE_SMOOTHINTERSECTION)
    ISDF_FIXED_CASE(ISDF_SHAPE_CSG) ISDF_FIXED_CASE(ISDF_SHAPE_BOX) ISDF_FIXED_CASE(ISDF_SHAPE_MESH)
    default: hipLaunchKernelGGL(swept_fixed_kernel<-1>, grid, block, 0, stream, P, tstar_in); break;
    }
#undef ISDF_FIXED_CASE
}
void launch_swept_reduce(const SweptParams &P, double *out, hipStream_t stream) {
    const int n = P.point_end - P.point_begin;
    if (n > 0) hipLaunchKernelGGL(swept_backprop_kernel, dim3((n + 127) / 128), dim3(128), 0, stream, P);
    hipLaunchKernelGGL(swept_reduce_kernel, dim3(P.N + 1), dim3(256), 0, stream, P, out, P.hist);
}

} // namespace isdf
