// Concurrent optimisation of a batch of trajectories on one shared map (BASELINE.json configs[2]: "batch of 128 random
// start/goal trajectories optimized concurrently").  Every trajectory runs the reference's L-BFGS driver
// (csrc/lbfgs_host.hpp == lbfgs::lbfgs_optimize, lbfgs.hpp:480-835) in its own host thread on its own objective
// (costFunctionLmbm, back_end_optimizer.hpp:358-430) with its own boundary states; the callbacks of a round of trajectories are
// evaluated TOGETHER as one batched integral sweep.  Rounds are GREEDY (a round starts as soon as a share of the trajectories
// that can submit is waiting) and several are in flight, each on its own stream, the sweeps ordered by events because they
// share the ctx's scratch - so one round's uploads, downloads and host work run under another round's sweep.
// MINCO (tau -> T, setParameters, energy; propogateGrad, chain rule) runs ON THE DEVICE, one workgroup per trajectory of the
// round, in the two small kernels of csrc/minco_dev.hip either side of the batched sweep (log-depth junction-state form,
// csrc/minco_pcr.hpp): a round moves n doubles per trajectory down and n + 5 up, and the trajectories' host threads only run
// the L-BFGS vector updates.  isdf_set_minco_mode(ctx, 1) keeps the round-4 form - MINCO on the trajectories' host threads
// (csrc/minco_host.hpp, the reference's band LU pivot for pivot; ~10 us per solve on a CPU core).  (The device form of rounds
// 2-4 walked that 6N-pivot chain with one wavefront per trajectory - 140 + 105 us per round - and was removed.)
// Which trajectories share a round has no influence on results: the iterates are bitwise those of isdf_optimize_lbfgs on each
// trajectory alone.
#include "isdf_ctx.hpp"
#include "lbfgs_host.hpp"
#include "minco_dev.hpp"
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>

namespace {

struct Batch;
struct Worker {
    Batch *batch = nullptr;
    int id = 0;
    isdf_host::LbfgsResult result;
    isdf_host::MincoS3 minco;
    std::vector<double> T, gdC, gdT, gradP, gradT;
    double energy = 0.0;
};

struct Batch {
    isdf_ctx *ctx = nullptr;
    int n_traj = 0, N = 0, n = 0;
    double rho = 0.0;
    size_t ostride = 0;
    std::vector<Worker> workers;
    // round state (guarded by m)
    std::mutex m;
    std::condition_variable cv_coord;
    std::unique_ptr<std::condition_variable[]> cv_worker;      // one per trajectory: a delivered round wakes its own trajectories only
    std::vector<char> submitted, finished, ready;     // ready[t]: the round trajectory t joined has been evaluated
    int n_submitted = 0, n_finished = 0, n_inflight = 0;
    double min_share = 0.25;      // greedy rounds: a round starts when this share of the live trajectories waits (1.0: all of them)
    // GROUP rounds (the default): the trajectories are dealt into `groups` fixed groups; a group's round starts when ALL its live
    // members wait, so every round is full, and while one group's sweep runs the other groups' host threads do their L-BFGS updates
    // (greedy rounds ran a third full on average: 14 000 evaluations in 450 rounds).  Measured on two boxes, 128 x 40 pieces x 30
    // iterations, device MINCO (profiles/r5_callback_bench.txt): 4 groups with three rounds in flight 75 ms on both (= 93 % of the
    // batched sweep's own rate), 3 groups / two in flight 75-83, 2 groups 74-95, 5-6 groups 97-103, one group 120, greedy 80-105; with
    // MINCO on the host threads 75-89.  0 (ISDF_BATCH_GROUPS=0): greedy rounds.
    int groups = 4;
    std::vector<int> sub_g, live_g, busy_g;
    int error = ISDF_OK;
    // staging: slot t holds [T(N) | coeffs(18N)]; out slot t holds the sweep's [cost | gradT | gradC]
    size_t in_per = 0;
    std::vector<double> stage_in, stage_out;
    // device MINCO: slot t of stage_x holds x (n), of stage_res [cost | g (n) | 4 cost parts]
    bool dev = false;
    size_t rs = 0;
    std::vector<double> stage_x, stage_res, ends;
    long long rounds_run = 0, evals_run = 0;
    isdf_progress_fn progress = nullptr;      // isdf_set_progress (copied from the ctx when the batch starts)
    void *progress_instance = nullptr;
    size_t progress_stride = 0;
};

// the objective of one trajectory; blocks until the coordinator has evaluated the round it joined
double worker_evaluate(void *instance, const double *x, double *g, const int n) {
    Worker &w = *(Worker *)instance;
    Batch &b = *w.batch;
    // serial part on this trajectory's host thread (tau -> T, MINCO, energy), the sweep in the round, then the adjoint here
    const int N = b.N;
    if (b.dev) {
        std::memcpy(b.stage_x.data() + (size_t)w.id * b.n, x, (size_t)n * sizeof(double));
        {
            std::unique_lock<std::mutex> lk(b.m);
            b.submitted[w.id] = 1;
            b.n_submitted++;
            if (b.groups > 0) b.sub_g[w.id % b.groups]++;
            b.cv_coord.notify_one();
            b.cv_worker[w.id].wait(lk, [&] { return b.ready[w.id] != 0; });
            b.ready[w.id] = 0;
            if (b.error != ISDF_OK) return INFINITY;
        }
        const double *res = b.stage_res.data() + (size_t)w.id * b.rs;
        std::memcpy(g, res + 1, (size_t)n * sizeof(double));
        return res[0];
    }
    for (int i = 0; i < N; i++) w.T[i] = isdf_host::tau_to_T(x[i]);
    w.minco.set_parameters(x + N, w.T.data());
    w.energy = w.minco.energy(w.gdC.data(), w.gdT.data());
    double *in = b.stage_in.data() + (size_t)w.id * b.in_per;
    std::memcpy(in, w.T.data(), (size_t)N * sizeof(double));
    std::memcpy(in + N, w.minco.c.data(), (size_t)18 * N * sizeof(double));
    {
        std::unique_lock<std::mutex> lk(b.m);
        b.submitted[w.id] = 1;
        b.n_submitted++;
        if (b.groups > 0) b.sub_g[w.id % b.groups]++;
        b.cv_coord.notify_one();
        b.cv_worker[w.id].wait(lk, [&] { return b.ready[w.id] != 0; });
        b.ready[w.id] = 0;
        if (b.error != ISDF_OK) return INFINITY;
    }
    const double *o = b.stage_out.data() + (size_t)w.id * b.ostride;
    double cost = w.energy + o[0];
    for (int i = 0; i < N; i++) w.gdT[i] += o[1 + i];
    for (int i = 0; i < 18 * N; i++) w.gdC[i] += o[1 + N + i];
    w.minco.propagate_grad(w.gdC.data(), w.gdT.data(), w.gradP.data(), w.gradT.data());
    double tsum = 0.0;
    for (int i = 0; i < N; i++) tsum += w.T[i];
    cost += b.rho * tsum;
    for (int i = 0; i < N; i++) g[i] = isdf_host::grad_T_to_tau(x[i], w.gradT[i] + b.rho);
    for (int i = 0; i < 3 * (N - 1); i++) g[N + i] = w.gradP[i];
    return cost;
}

}  // namespace

extern "C" int isdf_optimize_lbfgs_batch(isdf_ctx *c, int n_traj, int N, const double *heads_pva, const double *tails_pva, double rho,
                                         double *x_inout, const isdf_lbfgs_params *p, isdf_lbfgs_result *results, double *wall_ms_out) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    if (n_traj < 1 || N < 1 || !heads_pva || !tails_pva || !x_inout || !p || !results) return isdf_fail(c, ISDF_ERR_INVALID_ARG, "bad batch arguments");
    if (c->cfg.variant == ISDF_V1_SWEPT) return isdf_fail(c, ISDF_ERR_UNSUPPORTED, "the batched optimizer drives the integral sweep (V2/V3)");
    if (c->world > 1) return isdf_fail(c, ISDF_ERR_STATE, "sharded ctx: the batched optimizer would optimise on this rank's partial sums (shard the BATCH across ranks instead: one ctx per rank, isdf_set_shard(0, 1))");
    HIPCHK(c, hipSetDevice(c->device));
    Batch b;
    b.ctx = c; b.n_traj = n_traj; b.N = N; b.n = N + 3 * (N - 1); b.rho = rho;
    b.progress = c->progress; b.progress_instance = c->progress_instance; b.progress_stride = c->progress_stride;
    b.ostride = isdf_out_stride(N);
    b.workers.resize(n_traj);
    b.submitted.assign(n_traj, 0); b.finished.assign(n_traj, 0); b.ready.assign(n_traj, 0);
    b.cv_worker.reset(new std::condition_variable[n_traj]);
    if (c->minco_mode != 1 && N <= isdf::CB_MAX_N) b.min_share = 0.5;       // device MINCO: trajectories come back sooner, fuller rounds pay (measured 80 ms against 88 at 0.25)
    if (const char *e = getenv("ISDF_BATCH_GROUPS")) { const int v = atoi(e); if (v >= 0 && v <= 16) b.groups = v; }
    if (b.groups > n_traj) b.groups = n_traj;
    if (b.groups > 0) {
        b.sub_g.assign(b.groups, 0); b.live_g.assign(b.groups, 0); b.busy_g.assign(b.groups, 0);
        for (int t = 0; t < n_traj; t++) b.live_g[t % b.groups]++;
    }
    if (const char *e = getenv("ISDF_BATCH_MIN_SHARE")) { const double v = atof(e); if (v > 0.0 && v <= 1.0) b.min_share = v; }
    b.in_per = (size_t)19 * N;
    b.dev = c->minco_mode != 1 && N <= isdf::CB_MAX_N;
    b.rs = (size_t)1 + (size_t)b.n + 4;
    if (b.dev) {
        b.stage_x.assign((size_t)n_traj * b.n, 0.0); b.stage_res.assign((size_t)n_traj * b.rs, 0.0); b.ends.assign((size_t)n_traj * 18, 0.0);
        for (int t = 0; t < n_traj; t++) {
            std::memcpy(b.ends.data() + (size_t)18 * t, heads_pva + 9 * t, 9 * sizeof(double));
            std::memcpy(b.ends.data() + (size_t)18 * t + 9, tails_pva + 9 * t, 9 * sizeof(double));
        }
    }
    c->last_minco_path = b.dev ? 1 : 0;
    for (int t = 0; t < n_traj; t++) {
        Worker &w = b.workers[t];
        w.batch = &b; w.id = t;
        w.minco.set_conditions(heads_pva + 9 * t, tails_pva + 9 * t, N);
        w.T.assign(N, 0.0); w.gdC.assign((size_t)18 * N, 0.0); w.gdT.assign(N, 0.0);
        w.gradP.assign((size_t)3 * (N > 1 ? N - 1 : 1), 0.0); w.gradT.assign(N, 0.0);
    }
    b.stage_in.assign((size_t)n_traj * b.in_per, 0.0); b.stage_out.assign((size_t)n_traj * b.ostride, 0.0);
    // device + pinned buffers for a full batch per slot: N_SLOTS rounds are in flight, each on its own stream
    constexpr int N_SLOTS = 4;          // rounds in flight at most (slots_used of them are used)
    struct Slot {
        double *h_pin = nullptr, *d_buf = nullptr;
        std::vector<int> active; int group = -1; hipStream_t stream = nullptr; hipEvent_t done = nullptr, sweep_done = nullptr; bool busy = false;
    } slots[N_SLOTS];
    const size_t in_all = (size_t)n_traj * b.in_per, out_all = (size_t)n_traj * b.ostride;
    // device MINCO: further device regions behind [T | coeffs | sweep out] - x + ends of the round (compacted), junction states,
    // energy block, results; the pinned buffer then holds [x + ends of the round | ... | results | overflow word]
    const size_t xe_all = (size_t)n_traj * (b.n + 18), u_all = (size_t)n_traj * 6 * (N + 1), res_all = (size_t)n_traj * b.rs;
    const size_t off_xe = in_all + out_all, off_u = off_xe + xe_all, off_e = off_u + u_all, off_res = off_e + out_all;
    const size_t dev_all = b.dev ? off_res + res_all : in_all + out_all;
    const size_t pin_res = b.dev ? std::max(xe_all, in_all + out_all) : 0;        // results of a device round in the pinned buffer
    const size_t word_at = b.dev ? pin_res + res_all : in_all + out_all;
    auto release = [&] {
        for (Slot &q : slots) {
            if (q.h_pin) (void)hipHostFree(q.h_pin);
            if (q.d_buf) (void)hipFree(q.d_buf);
            if (q.done) (void)hipEventDestroy(q.done); if (q.sweep_done) (void)hipEventDestroy(q.sweep_done);
            if (q.stream) (void)hipStreamDestroy(q.stream);
        }
    };
    for (Slot &q : slots) {
        // pinned: [T | coeffs of the round | sweep out | overflow word]; device: [T | coeffs | sweep out]
        if (hipHostMalloc((void **)&q.h_pin, (word_at + 1) * sizeof(double), hipHostMallocDefault) != hipSuccess ||
            hipMalloc(&q.d_buf, dev_all * sizeof(double)) != hipSuccess ||
            hipStreamCreate(&q.stream) != hipSuccess ||
            hipEventCreateWithFlags(&q.done, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&q.sweep_done, hipEventDisableTiming) != hipSuccess) { release(); return isdf_fail(c, ISDF_ERR_HIP, "batch buffers: allocation failed"); }
    }
    { int rc0 = isdf_reserve_sweep_buffers(c, (long long)n_traj * N); if (rc0) { release(); return rc0; } }
    (void)hipDeviceSynchronize();

    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> threads;
    threads.reserve(n_traj);
    for (int t = 0; t < n_traj; t++) {
        threads.emplace_back([&b, p, x_inout, t] {
            Worker &w = b.workers[t];
            isdf_host::Lbfgs opt;
            opt.param.mem_size = p->mem_size; opt.param.past = p->past; opt.param.max_iterations = p->max_iterations;
            opt.param.max_linesearch = p->max_linesearch; opt.param.weak_wolfe = p->weak_wolfe; opt.param.reference_patches = p->reference_patches;
            opt.param.g_epsilon = p->g_epsilon; opt.param.delta = p->delta; opt.param.min_step = p->min_step; opt.param.max_step = p->max_step;
            opt.param.f_dec_coeff = p->f_dec_coeff; opt.param.s_curv_coeff = p->s_curv_coeff; opt.param.cautious_factor = p->cautious_factor;
            opt.param.machine_prec = p->machine_prec; opt.param.dir_norm_cap = p->dir_norm_cap;
            opt.evaluate = worker_evaluate;
            opt.instance = &w;
            // the caller's progress / cancel hook, per trajectory (isdf_set_progress): called on this trajectory's host thread
            opt.progress = b.progress;
            opt.progress_instance = b.progress ? (void *)((char *)b.progress_instance + (size_t)t * b.progress_stride) : nullptr;
            w.result = opt.minimize(x_inout + (size_t)t * b.n, b.n);
            std::unique_lock<std::mutex> lk(b.m);
            b.finished[t] = 1;
            b.n_finished++;
            if (b.groups > 0) b.live_g[t % b.groups]--;
            b.cv_coord.notify_one();
        });
    }
    // coordinator: greedy rounds, N_SLOTS in flight
    int rc = ISDF_OK;
    hipEvent_t last_sweep = nullptr;          // the sweeps share the ctx's scratch: each waits for the one issued before it
    auto issue = [&](Slot &q) {               // q.active holds the trajectories of the round; their staging slots are quiescent
        const int na = (int)q.active.size();
        if (b.dev) {
            // compacted [x of the round | their boundary states] down, pre -> sweep -> post on this slot's stream, results up
            double *hx = q.h_pin, *he = q.h_pin + (size_t)na * b.n;
            for (int k = 0; k < na; k++) {
                std::memcpy(hx + (size_t)k * b.n, b.stage_x.data() + (size_t)q.active[k] * b.n, (size_t)b.n * sizeof(double));
                std::memcpy(he + (size_t)k * 18, b.ends.data() + (size_t)q.active[k] * 18, 18 * sizeof(double));
            }
            isdf::CbDev P{};
            P.N = N; P.nb = na; P.n_out = 1; P.res_stride = (int)b.rs;
            P.x = q.d_buf + off_xe; P.ends = q.d_buf + off_xe + (size_t)na * b.n;
            P.T = q.d_buf; P.coeffs = q.d_buf + (size_t)na * N; P.sweep = q.d_buf + in_all;
            P.u = q.d_buf + off_u; P.epart = q.d_buf + off_e;
            P.rho = b.rho; P.res = q.d_buf + off_res; P.flag = nullptr; P.seq = 0; P.stats = nullptr;
            if (rc == ISDF_OK && hipMemcpyAsync(q.d_buf + off_xe, q.h_pin, (size_t)na * (b.n + 18) * sizeof(double), hipMemcpyHostToDevice, q.stream) != hipSuccess) rc = ISDF_ERR_HIP;
            if (rc == ISDF_OK) isdf::launch_cb_pre(P, q.stream);
            if (rc == ISDF_OK && last_sweep && hipStreamWaitEvent(q.stream, last_sweep, 0) != hipSuccess) rc = ISDF_ERR_HIP;
            if (rc == ISDF_OK) rc = isdf_eval_device(c, na, N, P.T, P.coeffs, q.d_buf + in_all, nullptr, q.stream);
            if (rc == ISDF_OK && hipEventRecord(q.sweep_done, q.stream) != hipSuccess) rc = ISDF_ERR_HIP;
            last_sweep = q.sweep_done;
            if (rc == ISDF_OK) isdf::launch_cb_post(P, q.stream);
            if (rc == ISDF_OK && hipGetLastError() != hipSuccess) rc = ISDF_ERR_HIP;
            if (rc == ISDF_OK && hipMemcpyAsync(q.h_pin + pin_res, q.d_buf + off_res, (size_t)na * b.rs * sizeof(double), hipMemcpyDeviceToHost, q.stream) != hipSuccess) rc = ISDF_ERR_HIP;
            if (rc == ISDF_OK && hipMemcpyAsync(q.h_pin + word_at, c->d_stats + 4, sizeof(double), hipMemcpyDeviceToHost, q.stream) != hipSuccess) rc = ISDF_ERR_HIP;
            if (rc == ISDF_OK && hipEventRecord(q.done, q.stream) != hipSuccess) rc = ISDF_ERR_HIP;
            q.busy = true;
            b.rounds_run++;
            b.evals_run += na;
            return;
        }
        double *hT = q.h_pin, *hC = q.h_pin + (size_t)na * N;          // compacted [T of all | coeffs of all]
        for (int k = 0; k < na; k++) {
            const double *in = b.stage_in.data() + (size_t)q.active[k] * b.in_per;
            std::memcpy(hT + (size_t)k * N, in, (size_t)N * sizeof(double));
            std::memcpy(hC + (size_t)k * 18 * N, in + N, (size_t)18 * N * sizeof(double));
        }
        double *dT = q.d_buf, *dC = q.d_buf + (size_t)na * N, *dO = q.d_buf + in_all;
        if (rc == ISDF_OK && hipMemcpyAsync(q.d_buf, q.h_pin, (size_t)na * b.in_per * sizeof(double), hipMemcpyHostToDevice, q.stream) != hipSuccess) rc = ISDF_ERR_HIP;
        if (rc == ISDF_OK && last_sweep && hipStreamWaitEvent(q.stream, last_sweep, 0) != hipSuccess) rc = ISDF_ERR_HIP;
        if (rc == ISDF_OK) rc = isdf_eval_device(c, na, N, dT, dC, dO, nullptr, q.stream);
        if (rc == ISDF_OK && hipEventRecord(q.sweep_done, q.stream) != hipSuccess) rc = ISDF_ERR_HIP;
        last_sweep = q.sweep_done;
        if (rc == ISDF_OK && hipMemcpyAsync(q.h_pin + in_all, dO, (size_t)na * b.ostride * sizeof(double), hipMemcpyDeviceToHost, q.stream) != hipSuccess) rc = ISDF_ERR_HIP;
        if (rc == ISDF_OK && hipMemcpyAsync(q.h_pin + word_at, c->d_stats + 4, sizeof(double), hipMemcpyDeviceToHost, q.stream) != hipSuccess) rc = ISDF_ERR_HIP;
        if (rc == ISDF_OK && hipEventRecord(q.done, q.stream) != hipSuccess) rc = ISDF_ERR_HIP;
        q.busy = true;
        b.rounds_run++;
        b.evals_run += na;
    };
    auto deliver = [&](Slot &q) {             // waits for the round, hands the results to its trajectories
        if (rc == ISDF_OK && hipEventSynchronize(q.done) != hipSuccess) rc = ISDF_ERR_HIP;
        const int na = (int)q.active.size();
        if (rc == ISDF_OK) {
            unsigned long long ovf = 0;
            std::memcpy(&ovf, q.h_pin + word_at, sizeof(ovf));
            if (ovf) { rc = ISDF_ERR_OVERFLOW; (void)hipMemset(c->d_stats + 4, 0, sizeof(unsigned long long)); (void)isdf_reset_result_slots(c); }
        }
        for (int k = 0; k < na && rc == ISDF_OK; k++) {
            if (b.dev) std::memcpy(b.stage_res.data() + (size_t)q.active[k] * b.rs, q.h_pin + pin_res + (size_t)k * b.rs, b.rs * sizeof(double));
            else std::memcpy(b.stage_out.data() + (size_t)q.active[k] * b.ostride, q.h_pin + in_all + (size_t)k * b.ostride, b.ostride * sizeof(double));
        }
        std::unique_lock<std::mutex> lk(b.m);
        if (rc != ISDF_OK) b.error = rc;
        for (int t : q.active) b.ready[t] = 1;
        b.n_inflight -= na;
        if (q.group >= 0) b.busy_g[q.group] = 0;
        q.busy = false;
        if (rc != ISDF_OK) { for (int t = 0; t < b.n_traj; t++) b.cv_worker[t].notify_one(); }     // an error ends every wait
        else for (int t : q.active) b.cv_worker[t].notify_one();
    };
    const int round_cap = n_traj;
    int slots_used = 3;
    if (const char *e = getenv("ISDF_BATCH_SLOTS")) { const int v = atoi(e); if (v >= 1 && v <= N_SLOTS) slots_used = v; }
    int head_slot = 0, tail_slot = 0, n_busy = 0;      // ring: rounds are delivered in the order they were issued
    for (;;) {
        if (n_busy == slots_used) { deliver(slots[tail_slot]); tail_slot = (tail_slot + 1) % slots_used; n_busy--; }
        bool take = false, done = false;
        int take_group = -1;
        {
            std::unique_lock<std::mutex> lk(b.m);
            for (;;) {
                if (b.groups > 0) {
                    for (int g = 0; g < b.groups && take_group < 0; g++)
                        if (!b.busy_g[g] && b.live_g[g] > 0 && b.sub_g[g] == b.live_g[g]) take_group = g;
                    if (take_group >= 0) { take = true; break; }
                } else {
                // trajectories in flight cannot submit: the share refers to those that can
                const int free_live = b.n_traj - b.n_finished - b.n_inflight;
                const int need = std::max(1, (int)std::ceil(b.min_share * free_live));
                if (b.n_submitted > 0 && b.n_submitted >= std::min(need, free_live)) { take = true; break; }
                }
                if (n_busy > 0) {
                    if (rc != ISDF_OK || hipEventQuery(slots[tail_slot].done) != hipErrorNotReady) break;   // finished (or failed): deliver it - that is where new work comes from
                    b.cv_coord.wait_for(lk, std::chrono::microseconds(20));
                    continue;
                }
                if (b.n_traj - b.n_finished == 0) { done = true; break; }
                b.cv_coord.wait(lk);
            }
            if (take) {
                Slot &q = slots[head_slot];
                // at most round_cap trajectories per round: a herd that submits together is split into several rounds whose
                // MINCO kernels run side by side and whose sweeps follow each other without a gap
                q.active.clear();
                q.group = take_group;
                if (take_group >= 0) {
                    for (int t = take_group; t < n_traj; t += b.groups) if (b.submitted[t]) { q.active.push_back(t); b.submitted[t] = 0; }
                    b.busy_g[take_group] = 1; b.sub_g[take_group] = 0;
                } else
                for (int t = 0; t < n_traj && (int)q.active.size() < round_cap; t++) if (b.submitted[t]) { q.active.push_back(t); b.submitted[t] = 0; }
                b.n_inflight += (int)q.active.size();
                b.n_submitted -= (int)q.active.size();
            }
        }
        if (done) break;
        if (take) { issue(slots[head_slot]); head_slot = (head_slot + 1) % slots_used; n_busy++; }
        else { deliver(slots[tail_slot]); tail_slot = (tail_slot + 1) % slots_used; n_busy--; }
    }
    for (auto &th : threads) th.join();
    const auto t1 = std::chrono::steady_clock::now();
    release();
    const double wall = std::chrono::duration<double, std::milli>(t1 - t0).count();
    for (int t = 0; t < n_traj; t++) {
        const isdf_host::LbfgsResult &r = b.workers[t].result;
        results[t].f = r.f; results[t].status = r.status; results[t].iterations = r.iterations; results[t].evaluations = r.evaluations;
        results[t].wall_ms = wall; results[t].reserved = (int32_t)b.rounds_run;
    }
    if (wall_ms_out) *wall_ms_out = wall;
    if (b.error != ISDF_OK) return isdf_fail(c, b.error, "a device round of the batched optimizer failed");
    return ISDF_OK;
}
