// Private host-side context of the C ABI (not installed; include/isdf_accel.h keeps isdf_ctx opaque).
#pragma once
#include "isdf_internal.hpp"
#include "minco_host.hpp"
#include <string>
#include <vector>

using namespace isdf;

struct ProfEvent { hipEvent_t a, b, c, d; };   // start/stop of the dominant kernel, start/stop of the one after it
struct isdf_xchg;
struct isdf_ctx {
    isdf_config cfg;
    int device = 0;
    std::string err;
    // grid
    DevGrid grid{};
    float *d_esdf = nullptr;
    int mesh_info[16] = {0};                    // isdf_mesh_info: what isdf_set_shape found and decided about the installed mesh
    const double *v1_tstar_stage = nullptr;     // set by the host-direct V1 step for ONE eval_device_impl call (SweptParams::tstar_stage)
    double *d_esdf_stage = nullptr; size_t esdf_stage_cap = 0;        // isdf_esdf_sample's staging (points | values | gradients): grows only, no allocation per call
    float *d_esdf_bricks = nullptr; size_t bricks_cap = 0; bool bricks_stale = true;     // the ESDF as 2 x 2 x 2-cell bricks with apron, one 128-byte line each (map_build.hip: scattered points)
    uint8_t *d_occ = nullptr;
    unsigned *d_bits = nullptr; size_t bits_cap = 0; bool bits_dirty = true;
    bool have_geom = false;
    // shape
    DevShape shape{};
    isdf_shape shape_host{};
    bool have_shape = false;
    DevMesh *d_mesh = nullptr; int mesh_depth = 0;     // levels of the mesh robot's hierarchy
    double *d_mesh_tri = nullptr;
    float *d_mesh_trif = nullptr;
    int *d_fwn_child = nullptr;
    float *d_fwn_box = nullptr, *d_fwn_boxq = nullptr;
    double *d_fwn_triq = nullptr;
    int *d_mesh_flat = nullptr;          // the flat slot table of a small mesh (DevMesh::flat)
    float *d_mesh_dl = nullptr;          // the mesh kind's distance lattice (DevMesh::dl)
    double *d_shape_grid = nullptr;      // ISDF_SHAPE_GRID: the sampled lattice
    void *d_pose = nullptr; size_t pose_cap = 0;      // pose records of a non-fused integral step (bytes)
    // points (V1)
    double *d_points = nullptr;
    int M = 0;
    double *d_tstar = nullptr;       // internal lastTstar when the caller passes none
    // shard
    int rank = 0, world = 1;
    // per-step scratch
    double *d_acc = nullptr; int *d_sample_info = nullptr; size_t sample_cap = 0;
    SweepParams last_P{}; bool have_last_P = false;
    double last_exact_ms = 0.0;
    double *d_piece_cost = nullptr; size_t piece_cap = 0;
    double *d_in = nullptr; size_t in_cap = 0;        // host-API staging: T | coeffs
    double *d_out = nullptr; size_t out_cap = 0;
    std::vector<double> h_out;
    unsigned long long *d_stats = nullptr;
    isdf_stats last_stats{};
    bool stats_cached = false;
    // V1 scratch
    double *d_traj_duration = nullptr;
    double *d_coarse_t = nullptr, *d_coarse_pose = nullptr;
    int *d_n_coarse = nullptr;
    double *d_point_partial = nullptr; int *d_point_piece = nullptr; unsigned long long *d_point_stat = nullptr; int point_cap = 0;
    int *d_point_nr = nullptr; double *d_task_buf = nullptr; unsigned *d_task_map = nullptr, *d_v1_words = nullptr, *d_point_lmask = nullptr;
    unsigned *d_scan_ticks = nullptr; int *d_scan_order = nullptr;     // V1: a step's scan records (class | rank per point) / the dispatch order of the NEXT step's scan built from them
    long long scan_order_b = -1, scan_order_e = -1; unsigned long long scan_order_epoch = 0, points_epoch = 1;   // ... and the shard of which points it is for
    double *d_hist = nullptr; size_t hist_cap = 0;
    // profiling
    bool fuse_small = true;     // developer switch ISDF_NO_FUSE=1: always sweep + tail as two launches
    // developer / fallback switches, ALL read once per ctx in isdf_create (so a process can hold ctxs of either kind):
    // ISDF_NO_HOST_DIRECT=1 (isdf_eval and the callback take the copy path), ISDF_NO_BAR_WRITES=1 (host-direct steps fetch their
    // inputs from host-mapped memory instead of receiving them through the PCIe BAR), ISDF_NO_LPT=1 (plain dispatch order)
    bool env_no_direct = false, env_no_bar = false, env_no_lpt = false;
    int last_host_path = 0;     // isdf_host_path(): how the last host-array step crossed PCIe
    bool prof_on = false, prof_secondary = false; int prof_every = 1; long long prof_tick = 0;
    std::vector<ProfEvent> prof_events;
    size_t prof_used = 0;
    hipStream_t stream = nullptr;   // stream of the host API
    int *d_sample_map = nullptr; size_t sample_map_n = 0;      // developer override of the tile sweep's dispatch order
    // longest-first order of fused single-launch steps: two generations of work classes and of orders, the geometry they belong to
    unsigned char *d_plan_cls = nullptr; int *d_plan_map = nullptr; unsigned short *d_plan_lr = nullptr, *d_plan_hist = nullptr;
    long long plan_ns = -1, plan_geo = -1; int plan_nb = -1; int plan_k = 0; int plan_cur = 0; int n_cus = 0; size_t plan_ns_cap = 0, plan_np_cap = 0;
    unsigned long long *d_dbg = nullptr; size_t dbg_cap = 0; size_t dbg_used = 0;   // ISDF_DEBUG_TIMING=1 (developer tool)
    // full objective callback (costFunctionLmbm): MINCO on the host, sweeps on the device
    isdf_host::MincoS3 minco; bool have_traj = false; double rho = 0.0;
    std::vector<double> cb_T, cb_gdC, cb_gdT, cb_gradP, cb_gradT;
    double *h_eval_pin = nullptr; size_t eval_pin_cap = 0;   // pinned staging of isdf_eval: [inputs | outputs | statistics]
    double *h_eval_pin_dev = nullptr;                        // ... as the devices see it
    // multi-device ctx, isdf_eval: the sum kernel writes the step's outputs, the statistics words and a completion word straight
    // into the pinned buffer (no download commands, no stream synchronisation: the calling thread spins on the word)
    double *mh_out = nullptr; unsigned long long *mh_words = nullptr; unsigned long long mh_seq = 0; unsigned *d_msum_blocks = nullptr;
    bool env_multi_no_hostout = false;
    double *h_pin = nullptr; size_t pin_cap = 0;      // pinned staging: [T | coeffs | out_a | out_b]
    double *d_cb = nullptr; size_t cb_cap = 0;        // device twin of the staging buffer
    // host-direct steps (isdf_eval / isdf_cost_function on one GPU when the step is one fused launch): pinned, device-mapped
    // [inputs | outputs | one flag per trajectory]; the launch reads the inputs and writes the outputs over PCIe itself
    double *h_dir = nullptr, *h_dir_dev = nullptr; size_t dir_cap = 0; size_t dir_in = 0, dir_out = 0, dir_flags = 0;
    double *d_stage = nullptr; size_t stage_cap = 0;
    unsigned long long *d_stage_flags = nullptr; size_t stage_flags_cap = 0;
    unsigned long long dir_seq = 0;
    unsigned long long host_steps = 0, host_late = 0, host_late_spins = 0, host_late_mark = 0;      // isdf_host_info: host-mapped result hand-overs / those whose rows landed after the flag
    int bar_state = 0;          // 0: untested, 1: the host can write device memory through the PCIe BAR (verified), -1: it cannot
    bool dir_pending = false; int dir_nb = 0, dir_n = 0; bool cb_direct = false;
    double *h_v1_pin = nullptr, *h_v1_pin_dev = nullptr; size_t v1_pin_cap = 0;     // host-direct swept-volume step: [out | statistics | flag | lastTstar], device-mapped
    double last_parts[4] = {0, 0, 0, 0};
    std::vector<double> cb_x; double cb_energy = 0.0; int cb_n_out = 1; bool cb_pending = false;
    // device half of the callback (csrc/minco_dev.hip): MINCO, energy, adjoint and chain rule in two small kernels either side
    // of the sweeps - a callback moves n doubles down and n + 5 up.  minco_mode 0: wherever it is faster (isdf_host.hip
    // cb_device_minco), 1 (ISDF_HOST_MINCO=1): the host's band LU, bitwise the reference's elimination order, 2
    // (ISDF_DEVICE_MINCO=1): on the device whenever N <= CB_MAX_N
    int minco_mode = 0; int last_minco_path = 0;          // last_minco_path: 1 = the last callback ran MINCO on the device
    double *d_cbdev = nullptr; size_t cbdev_cap = 0;      // [x | ends(18) | u | energy block]
    double *h_cbres = nullptr, *h_cbres_dev = nullptr; size_t cbres_cap = 0;   // pinned, device-mapped: [x staging | cost, g, parts | flag]
    unsigned long long cb_seq = 0; bool cb_dev = false, cb_post_queued = false; bool cb_ends_dirty = true;
    double cb_ends[18] = {0};
    // front end (csrc/frontend.hip): attitude kernels of the robot, inflated bit-packed occupancy, breadth-first order tables
    struct FrontEnd {
        isdf_frontend_config cfg{}; int xk = 0, yk = 0; double margin = 0.0; bool built = false;
        unsigned *d_rows = nullptr, *d_bits = nullptr; double *d_rot = nullptr;
        unsigned short *d_seq = nullptr; int *d_seq_len = nullptr; int seq_stride = 0;
        void *d_row_list = nullptr; int *d_row_ptr = nullptr; int n_row_list = 0;      // non-empty rows per attitude
        unsigned *d_cspace = nullptr;                                                    // 4 dwords per voxel
        // the A* (isdf_frontend_astar_search): the table on the host, the breadth-first attitude orders, the last path
        unsigned *h_cspace = nullptr; bool h_cspace_valid = false, h_cspace_pinned = false;   // pinned when the host lets us (16 B per voxel), else pageable
        std::vector<unsigned short> h_seq; std::vector<int> h_seq_len;
        std::vector<double> path_xyz, path_rp;
    } fe;
    struct isdf_xchg *xchg = nullptr;           // peer-to-peer exchange of the multi-GPU path (csrc/xchg.hip)
    isdf_progress_fn progress = nullptr;        // isdf_set_progress: the optimizer drivers' progress / cancel hook
    void *progress_instance = nullptr;
    size_t progress_stride = 0;                 // batch: trajectory t's hook gets (char *)progress_instance + t * stride
    double xchg_timeout_ms = 2000.0;             // bound of the exchange's waits (isdf_xchg_set_timeout_ms), device wall clock
    // ONE host process driving SEVERAL devices (isdf_create_multi): this ctx is the lead (shard 0 of n on devices[0]) and owns
    // one plain ctx per further device (shard r of n).  Once-per-plan state set on the lead is replicated; a step launches every
    // shard on its own device's stream from the calling thread and sums the shards' packed outputs on the lead (one kernel
    // reading the peers' buffers over xGMI in rank order, or RCCL's all-reduce) - SURVEY 8(b) "Threading".
    std::vector<isdf_ctx *> peers;
    bool is_peer = false;                       // owned by a lead: not handed to the caller
    hipEvent_t mev_in = nullptr, mev_done = nullptr;      // lead: inputs ready on the caller's stream / the step's sum has run; peer: shard finished
    // mesh robots on the tile sweep: the queue of 64-voxel blocks between the scan launch and the exact launch
    unsigned *d_mq_entries = nullptr, *d_mq_count = nullptr;
    void *d_mq_items = nullptr;
    double *d_mq_res = nullptr;
    int *d_mq_sample_items = nullptr, *d_mq_sample_n = nullptr;
    size_t mq_cap = 0, mq_samples_cap = 0;
    int mq_kmax = 0;
    bool multi_pull = false;                              // lead: the peers read the lead's inputs in place (peer access both ways)
    bool msum_recorded = false;                           // lead: mev_done has been recorded at least once
    int multi_collective = 0;                   // ISDF_MULTI_*: how the shards' outputs are summed
    void *rccl_lib = nullptr; void *rccl_comm = nullptr;  // RCCL by dlopen (only when asked for): this device's communicator
    double *d_mpart = nullptr; size_t mpart_cap = 0;     // every shard of a multi-device step writes [packed outputs | 8 statistics as doubles] here
    double *d_mstage = nullptr; size_t mstage_cap = 0;   // lead, staged mode: the peers' parts copied next to each other
};
namespace isdf { struct XFuse; }
// xchg.hip: fills the in-kernel exchange block of a fused step when isdf_xchg_fuse is on (returns false: not requested;
// *err != ISDF_OK: requested but impossible for this launch)
bool isdf_xchg_fill(isdf_ctx *c, isdf::XFuse *xf, size_t pieces, int *err);
bool isdf_xchg_fuse_on(const isdf_ctx *c);     // isdf_xchg_fuse(ctx, 1) is in force
int isdf_reserve_sweep_buffers(isdf_ctx *c, long long total_pieces);   // isdf_host.hip: scratch of the integral sweep (grows only)
void isdf_xchg_reset_board(isdf_ctx *c);  // xchg.hip: empties this rank's board again (after an overflow)
int isdf_reset_result_slots(isdf_ctx *c); // isdf_host.hip: drains the device and empties every self-resetting slot again (after an overflow)
void isdf_xchg_release(isdf_ctx *c);          // xchg.hip: closes the peer mappings, frees the mailbox (isdf_destroy)
void isdf_frontend_release(isdf_ctx *c);      // frontend.hip: frees the tables (isdf_destroy)
int isdf_mesh_lattice_build(isdf_ctx *c, isdf::DevMesh *hm, const double lo[3], const double hi[3], int n, float s_range_out[2]);
int isdf_mesh_surface_valid(isdf_ctx *c, const double *d_tri, int nF, double extent, double tau_limit, int *valid_out, float defect_out[2]);      // shape_eval.hip: exact winding number 0 / 1 on both sides of every face      // shape_eval.hip: the mesh kind's distance lattice (DevMesh::dl)

// every setter of once-per-plan state ends with this: the same call on every owned peer ctx (isdf_create_multi)
#define ISDF_REPLICATE(ctx, call)                                                                  \
    do {                                                                                           \
        for (isdf_ctx *p_ : (ctx)->peers) {                                                        \
            const int r_ = (call);                                                                 \
            if (r_ != ISDF_OK) { (ctx)->err = "device " + std::to_string(p_->device) + ": " + p_->err; return r_; } \
        }                                                                                          \
    } while (0)

#define HIPCHK(ctx, call)                                                                          \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                        \
            return ISDF_ERR_HIP;                                                                   \
        }                                                                                          \
    } while (0)

int isdf_fail(isdf_ctx *c, int code, const char *msg);      // records the message, returns code
int isdf_ensure_doubles(isdf_ctx *c, double **p, size_t *cap, size_t need);
