// Internal host<->kernel parameter blocks (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdint>
#include "dev_shapes.hpp"

namespace isdf {

struct DevGrid {
    int X, Y, Z;
    double res;
    double bmin[3], bmax[3];
    const float *esdf;        // metres, unsigned; z fastest (GridMap3D.h:194)
    const uint8_t *occ;       // 1 = occupied
    const unsigned *bits;     // qualifier bit-grid of the active variant, ZW dwords per z-row: row (x, y), bit z
    int ZW;                   // ceil(Z / 32)
    // the same bits packed along the other two axes, so that a tile can be scanned along the axis its robot is longest on:
    const unsigned *bits_x;   // row (y, z): dword (y * Z + z) * XW + (x >> 5), bit x & 31
    const unsigned *bits_y;   // row (x, z): dword (x * Z + z) * YW + (y >> 5), bit y & 31
    int XW, YW;
};
constexpr int PLAN_CLASSES = 8;          // work classes of the longest-first order
constexpr int PLAN_GROUP_MAX_SAMPLES = 65535;   // samples sorted together: their per-class counts are added in packed 16-bit fields
constexpr int ACC_STRIDE = 8;       // costp, dL/dpos(3), dL/dquat(4) of one pose, before weight_p
constexpr int PARTIAL_STRIDE = 20;  // V1: per-point record = 18 gradC entries (d*6 + r), gdT, pena

// In-kernel exchange of a fused multi-GPU step (set up by csrc/xchg.hip, used by tile_sweep.hip's tail): every rank has a
// BOARD in IPC-mapped uncached memory, [2 step parities][pieces of the launch][XF_ROW] result slots (19 output rows + cost of
// a piece), all-ones while empty.  The workgroup that owns a piece stores its rows straight into every peer's board; the
// peer's workgroup for that piece polls them into its own output.  world <= 1: off.
constexpr int STAGE_G = 4;           // pieces staged per workgroup / per ready flag of a host-direct step
constexpr unsigned long long HOST_FLAG_OVERFLOW = 1ull << 62;
constexpr int XCHG_MAX_WORLD = 16;
constexpr int XF_ROW = 20;
struct XFuse {
    int world, rank, parity;
    int timeout_ticks;                  // bound of a wait for a peer's row, in wall_clock64() ticks (isdf_xchg_timeout_ms)
    unsigned long long parity_stride;   // doubles between the two parities' regions of a board: half the board, FIXED for the
                                        // life of the exchange (a stride derived from the launch's piece count would make
                                        // consecutive steps of different size overlap while a rank is one step ahead)
    double *board[XCHG_MAX_WORLD];
};

struct SweepParams {
    XFuse xf;
    DevGrid grid;
    DevShape shape;
    FlatP flat;
    int variant, K, enable_dyn, enable_pos, enable_cull;
    int n_traj, N;              // all trajectories of a launch have N pieces
    int piece_begin, piece_end; // this rank's range of GLOBAL piece indices (traj-major)
    double bd_half;             // kernel_size * res / 2
    double safety_hor, weight_p, weight_v, weight_omg, weight_theta;
    double vel_sqr_max, omg_sqr_max, theta_max, mu, inv_mu;
    int n_sweep_blocks;         // set by launch_sweep: sweep workgroups of the launch (a fused launch appends one tail workgroup per piece)
    double inv_K;               // 1.0 / K formed on the host (the reference's integralFrac, correctly rounded)
    float occ_thresh;           // (informational; the qualifier bit-grid is built with the double)
    double cull_threshold;      // <= 0: cull disabled
    const double *T;            // n_traj * N
    const double *coeffs;       // n_traj * (6N x 3 col-major)
    double *acc;                // [n_samples_local][ACC_STRIDE] per-pose collision sums written by sweep_kernel
    int *sample_info;           // [n_samples_local][2] exact pairs, active pairs | culled << 31 (statistics)
    double *piece_cost;         // [n_traj * N] slots: a piece's cost once its block has published it, all-ones while empty (self-resetting)
    double *out;                // n_traj * (1 + N + 18N)
    unsigned long long *stats;  // [0]=units [1]=culled [2]=pairs [3]=grad pairs [4]=overflow
    // host-direct step (isdf_eval / the callback on one GPU, fused launch only).  Outputs: `out` points into host-mapped
    // memory - every piece's workgroup stores its rows there and releases them (system scope) before its cost; the
    // trajectory's collector then stores host_flag[b] = seq (| overflow bit), the only thing the host waits for.  Inputs: either the host has written them straight into P.T / P.coeffs in
    // device memory through the PCIe BAR (host_T == null), or - no large BAR - the first workgroups copy them from
    // host-mapped memory into `stage` ([piece][19] = T, c(6x3)) and raise stage_flags[piece / STAGE_G] = seq, for which every
    // consumer waits.
    const double *host_T, *host_coeffs;   // device-visible addresses of the pinned inputs (null: inputs are P.T / P.coeffs)
    double *stage;
    unsigned long long *stage_flags;
    unsigned long long seq;
    unsigned long long *host_flag;
    // Dispatch order of a fused single-launch step (scheduling only; results cannot depend on it): rank in the launch
    // (workgroup b, wave w: b + w * n_sweep_blocks) -> local sample, -1 = none.  The tail workgroups of step k build the order
    // for step k + 1 from the work classes the sweep waves of step k - 1 left (plan_piece, tile_sweep.hip).
    void *poses;                // non-fused launches: [local samples] pose records (tile_sweep.hip SweepPose) written by pose_kernel
    const int *sample_map;
    unsigned char *plan_cls_out;        // this step's work classes, one per local sample (sweep waves)
    const unsigned char *plan_cls_in;   // the classes of the step before -> plan_lr_out / plan_hist_out (tail workgroups)
    unsigned short *plan_lr_out, *plan_hist_out;
    const unsigned short *plan_lr_in, *plan_hist_in;   // the records made one step ago -> plan_map_out
    int *plan_map_out;                  // the next step's order
    int plan_zone;                      // resident launch: workgroups that get the heaviest samples (one per CU); 0: workgroups of four neighbours of the sorted order, heaviest first
    int plan_group;                     // pieces sorted together (the whole launch, or one trajectory of a batch)
    int dbg_flags;              // developer switches (ISDF_DEBUG_FLAGS): 1 = skip the exact pass body
    unsigned long long *dbg;    // developer timing buffer (ISDF_DEBUG_TIMING=1), null on the product path
    // mesh robots: the exact pass as its own launch over a QUEUE of 16-voxel blocks (tile_sweep.hip mesh_exact_kernel); mq_items null: in the sweep kernel
    unsigned *mq_entries;       // [mq_cap][16] ring entries (x | y << 10 | z << 20 inside the tile)
    int2 *mq_items;             // [mq_cap] (local sample, voxels of the block)
    double *mq_res;             // [mq_cap][10] per block: the 8 collision sums, exact pairs, active pairs
    int *mq_sample_items;       // [local samples][mq_kmax] the item numbers of a sample's blocks, in block order
    int *mq_sample_n;           // [local samples] blocks of the sample | culled << 30
    unsigned *mq_count;         // device counters: [0] items queued this step, [1] items claimed by the exact launch (zeroed by pose_kernel)
    unsigned mq_cap;
    int mq_kmax;
};

// one optimizer step of the integral sweep = sweep, tail, in order, on `stream`
// ev_start / ev_stop (nullable): HIP events attached to the kernel dispatch itself (hipExtLaunchKernelGGL), i.e. its begin and
// end timestamps - the same interval rocprofv3 reports for the kernel
size_t sweep_pose_bytes();      // sizeof(SweepPose)
bool sweep_can_fuse(const SweepParams &P);
int sweep_resident_blocks(const SweepParams &P, int n_cus);      // small launch, analytic shape with identity body offset: sweep + tail in one launch
void launch_sweep(const SweepParams &P, hipStream_t stream, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr, bool fused = false);
void launch_stats(const SweepParams &P, hipStream_t stream);   // on demand, not part of the step
void launch_build_bits(const DevGrid &G, int use_esdf, double thresh, unsigned *out, hipStream_t stream);
void launch_build_bits_xy(const DevGrid &G, int use_esdf, double thresh, unsigned *out_x, unsigned *out_y, hipStream_t stream);
void launch_tail(const SweepParams &P, hipStream_t stream, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);

// ---- V1 (swept volume) ----
struct SweptParams {
    DevShape shape;
    FlatP flat;
    int N;
    int M, point_begin, point_end;
    double safety_hor, weight_p;
    const double *T;            // N
    const double *coeffs;       // 6N x 3 col-major
    const double *points;       // M x 3
    double *tstar;              // M (in/out), may be null -> internal scratch
    const double *tstar_stage;  // host-direct steps: the caller's lastTstar as the CPU stored it (a region the GPU never writes); the prepare
                                // kernel copies it into tstar.  null: tstar already holds it
    double *traj_duration;      // device scalar: persistent "traj_duration" state (sw_manager.hpp:287-296)
    // per-step scratch
    double *coarse_t;           // [max_coarse]
    double *coarse_pose;        // [max_coarse][12] = pos(3) + R(9)
    int *n_coarse;              // device scalar
    int max_coarse;
    double *point_partial;      // [M][PARTIAL_STRIDE] (gdC 18, gdT, pena)
    int *point_piece;           // [M]
    int *point_nr;              // [M] scan kernel: intervals | scan passes << 8
    double *task_buf;           // [M][32][6] per (point, interval): in (lb, ub, seed), out (t, sdf, gradient, counters)
    unsigned *task_map;         // [<= 32 M] task list of the descent: point (relative to point_begin) << 5 | interval
    unsigned *scan_ticks;       // [M] this step's scan record of point i (mesh robots: its duration in 1.28 us units; analytic: its fine passes); null: none kept
    const int *scan_order;      // [M] this step's dispatch order of the scan, longest first by LAST step's records (null: point order)
    int *scan_order_out;        // [M] where an order is written: by the prepare kernel for THIS step (scan_sort_here, mesh robots), else by the back-prop kernel for the NEXT
    int scan_sort_here;
    unsigned *point_lmask;      // [M] bit r: the descent of interval r was long last step (set by the descent kernel, consumed by the scan's list builder)
    unsigned *words;            // 32 words: [8], [9] the two ends of the next scan order (analytic robots); [0] tasks at the front of the list (long last step) [2] tasks at its back: zeroed by the prepare kernel, reserved by the
                                // scan kernel's workgroups
    int direct_records;         // 1: the records in point_partial were written by the fixed-t* kernel (no task slots)
    unsigned long long *point_stat;   // [M] active | intervals << 1 | passes << 9 | descent iterations << 25 (statistics)
    double *hist;               // [N] per-piece sum of gdT
    unsigned long long *dbg;    // developer timing buffer (ISDF_DEBUG_TIMING=1): 4 words per point, null on the product path
    unsigned long long *stats;  // [0]=units [2]=intervals [3]=active points [4]=overflow [5]=descent iterations [6]=reduce ticket [7]=passes
};
void launch_swept_prepare(const SweptParams &P, hipStream_t stream);
void launch_swept_sweep(const SweptParams &P, hipStream_t stream, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
void launch_swept_reduce(const SweptParams &P, double *out, hipStream_t stream);      // back-prop + sums (writes every entry of `out`)
void launch_swept_fixed(const SweptParams &P, const double *tstar_in, hipStream_t stream);   // minimisers given: replaces prepare + sweep

} // namespace isdf
