// The one exchange step of the multi-GPU path: the sum of the ranks' packed [cost | gradT | gradC] vectors.
// RCCL's all-reduce costs ~20-30 us for these 6-50 KB on one xGMI node - about as long as the whole optimizer step - so the
// library also offers a ONE-SHOT peer-to-peer form for the ranks of one node: every rank stores its vector straight into a
// mailbox slot on every peer (xGMI peer stores through IPC-mapped, uncached device memory), raises a sequence flag there,
// waits for its own mailbox to fill, and adds the slots in RANK ORDER - one kernel, no ring, and every rank forms bit for bit
// the same sum.  Mailboxes are double-buffered by sequence parity (a rank can only be one exchange ahead of a peer, because
// finishing exchange k needs every peer's flag k).  Every wait is bounded: on a time-out the status word is set and the
// caller falls back to RCCL (implicit-sdf-planner_amd/parallel.py checks the path against RCCL before using it).
#include "isdf_ctx.hpp"
#include "isdf_internal.hpp"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>

namespace isdf {

// Every wait is bounded by the DEVICE WALL CLOCK (wall_clock64(): a constant-rate counter, 100 MHz on gfx950; the rate is read
// from hipDeviceAttributeWallClockRate when the exchange is created), not by a spin count: the bound is the same number of
// milliseconds on every box, the caller can read it (isdf_xchg_timeout_ms) and set it (isdf_xchg_set_timeout_ms).
//   T  = the time-out: how much later than this rank a peer may START the exchange (data wait, push-counter wait)
//   2T + T/4 from this rank's start: the verdict round's bound (a peer whose data arrived started before t0 + T, its own
//        data wait ends before its start + T, so its verdict is here before t0 + 2T)
//   the other blocks of a launch give up only while block 0 has not STARTED (4T + T from their own start), see below
constexpr double XCHG_DEFAULT_TIMEOUT_MS = 2000.0;     // generous: ranks are separate processes (a first step loads code objects, a host thread may be descheduled); a peer later than this is a failure, not jitter

struct XchgParams {
    int rank, world;
    size_t slot_doubles;                  // capacity of one slot
    unsigned long long seq;               // this exchange (1, 2, ...)
    unsigned long long t_wait;            // T in wall-clock ticks
    double *peer_slots[XCHG_MAX_WORLD];   // base of rank r's mailbox data: [2 parities][world slots][slot_doubles]
    unsigned long long *peer_flags[XCHG_MAX_WORLD];   // rank r's flags: [2 parities][world] "data of exchange seq has arrived", then
                                                      // [2 parities][world] VERDICT words (seq << 1 | "my wait timed out")
    unsigned *status;                     // device word: != 0 after a time-out
    unsigned long long *pushes;           // device counter of completed pushes (world - 1 per exchange), local to this rank
    unsigned long long *decision;         // device word (seq << 2 | state): the rank's ONE decision for exchange seq
};

// states of the decision word for one exchange.  It moves NONE -> STARTED -> OK | BAD (block 0) or NONE -> BAD (another block
// whose hang guard expired before block 0 ever started); every transition is a compare-and-swap, so the rank has ONE verdict:
// whoever loses the race adopts the winner's.
constexpr unsigned long long XD_STARTED = 1ull, XD_OK = 2ull, XD_BAD = 3ull;

__device__ __forceinline__ unsigned long long xd_load(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
}
// Move the decision word of exchange `seq` to `to` and return the state in force afterwards.  A FINAL state of this exchange is
// never changed (the caller adopts it).  from_none_only (the other blocks' hang guard): a word that block 0 has already moved to
// STARTED is left alone and STARTED is returned.
__device__ __forceinline__ unsigned long long xd_move(unsigned long long *p, unsigned long long seq, unsigned long long to, bool from_none_only) {
    unsigned long long cur = xd_load(p);
    for (;;) {
        if ((cur >> 2) == seq) {
            const unsigned long long st = cur & 3ull;
            if (st >= XD_OK || st == to || from_none_only) return st;
        }
        const unsigned long long want = (seq << 2) | to;
        if (__hip_atomic_compare_exchange_strong(p, &cur, want, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) return to;
    }
}

// grid: one block per PEER pushes (blockIdx.x < world - 1), then every block takes part in the wait + rank-order sum
__global__ __launch_bounds__(1024) void xchg_allreduce_kernel(XchgParams X, double *__restrict__ buf, size_t count) {
    const int par = (int)(X.seq & 1ull);
    const int nb = gridDim.x;
    const unsigned long long t0 = wall_clock64();
    // ---- push: this rank's vector into slot [par][rank] of every peer, then the flag
    for (int pb = blockIdx.x; pb < X.world - 1; pb += nb) {
        const int peer = pb < X.rank ? pb : pb + 1;
        double *dst = X.peer_slots[peer] + ((size_t)par * X.world + X.rank) * X.slot_doubles;
        for (size_t i = threadIdx.x; i < count; i += blockDim.x) __builtin_nontemporal_store(buf[i], dst + i);
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_store(X.peer_flags[peer] + (size_t)par * X.world + X.rank, X.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_fetch_add(X.pushes, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
    // ---- wait (bounded) + verdict round: BLOCK 0 decides for the whole rank.  Lane r of its first wavefront polls peer r's flag
    // in MY mailbox, lane `world` this rank's own push counter (the sum below overwrites buf: every push of THIS rank, possibly
    // by another block, must have read it first) - all at once, one round trip instead of one per peer.
    __shared__ int s_ok;
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        const int lane = threadIdx.x;
        // block 0 is running: from here on its own clock bounds the exchange (<= 2T + T/4), the other blocks stop timing
        unsigned long long st = XD_STARTED;
        if (lane == 0) st = xd_move(X.decision, X.seq, XD_STARTED, false);
        st = __shfl(st, 0);
        bool ok = st != XD_BAD;             // another block gave up before this one started: the rank's verdict is already "bad"
        const unsigned long long d_data = t0 + X.t_wait;
        if (ok && lane < X.world && lane != X.rank) {
            const unsigned long long *mine = X.peer_flags[X.rank] + (size_t)par * X.world;
            while (__hip_atomic_load(mine + lane, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != X.seq) {
                __builtin_amdgcn_s_sleep(2);
                if (wall_clock64() > d_data) { ok = false; break; }
            }
        } else if (ok && lane == X.world) {
            const unsigned long long want = X.seq * (unsigned long long)(X.world - 1);
            while (__hip_atomic_load(X.pushes, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(2);
                if (wall_clock64() > d_data) { ok = false; break; }
            }
        }
        bool all_ok = __ballot(!ok) == 0ull;
        // ---- verdict round: a time-out is one-sided by nature (THIS mailbox was late; the late peer itself may have received
        // everything), and ranks that disagree about a step would walk different optimizer paths.  So every rank tells every
        // peer what its wait saw - one more 8-byte store per peer and one more poll, inside this kernel, instead of a host-side
        // collective after every step - and the exchange counts as good only if EVERY rank's wait was.  A verdict that does
        // not arrive in time is a bad one.  A rank whose own wait failed has nothing to learn from the peers' verdicts: it
        // posts "bad" and leaves (its kernel lasts T, not 3T).
        unsigned long long *const verdict_own = X.peer_flags[X.rank] + (size_t)(2 + par) * X.world;
        if (lane < X.world) {
            const unsigned long long word = (X.seq << 1) | (all_ok ? 0ull : 1ull);
            __hip_atomic_store(X.peer_flags[lane] + (size_t)(2 + par) * X.world + X.rank, word, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            ok = all_ok;
            if (all_ok) {
                const unsigned long long d_verdict = t0 + 2ull * X.t_wait + (X.t_wait >> 2);
                unsigned long long v;
                while (((v = __hip_atomic_load(verdict_own + lane, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM)) >> 1) != X.seq) {
                    __builtin_amdgcn_s_sleep(2);
                    if (wall_clock64() > d_verdict) { v = 1ull; break; }
                }
                ok = (v & 1ull) == 0ull;
            }
        } else ok = true;
        all_ok = __ballot(!ok) == 0ull;
        if (lane == 0) {
            // the rank's ONE decision for exchange seq (release: the other blocks read the mailbox after it)
            const unsigned long long fin = xd_move(X.decision, X.seq, all_ok ? XD_OK : XD_BAD, false);
            if (fin != XD_OK) atomicExch(X.status, 1u);
            s_ok = fin == XD_OK ? 1 : 0;
        }
    } else if (blockIdx.x != 0 && threadIdx.x == 0) {
        // The other blocks wait for block 0's decision.  They time ONLY block 0's absence: once the word says STARTED, block 0's
        // own clock ends the exchange within 2T + T/4, so a block here cannot give up while block 0 goes on to decide "ok" (a
        // finite cost over a gradient slice that was never summed).  If block 0 has not even started 5T after this block did
        // (not resident, faulted), this block moves the word NONE -> BAD: block 0, should it ever run, adopts that verdict, posts
        // it to the peers and writes the NaN.  A second, far longer guard covers a block 0 that started and then vanished.
        const unsigned long long d_absent = t0 + 5ull * X.t_wait, d_vanished = t0 + 16ull * X.t_wait;
        unsigned long long fin = 0ull;
        for (;;) {
            const unsigned long long cur = xd_load(X.decision);
            const bool this_seq = (cur >> 2) == X.seq;
            if (this_seq && (cur & 3ull) >= XD_OK) { fin = cur & 3ull; break; }
            const unsigned long long now = wall_clock64();
            if ((!this_seq && now > d_absent) || now > d_vanished) {
                fin = xd_move(X.decision, X.seq, XD_BAD, /*from_none_only=*/now <= d_vanished);
                if (fin == XD_STARTED) continue;            // block 0 started in this very moment: keep waiting for it
                if (fin != XD_OK) atomicExch(X.status, 1u);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        s_ok = fin == XD_OK ? 1 : 0;
    }
    __syncthreads();
    if (!s_ok) {
        // somebody's wait expired, and by the verdict round EVERY rank knows: all of them leave a NaN cost (and the sticky
        // status word, which tells this NaN from a diverged iterate's) and all of them redo the step on the group's all-reduce
        if (blockIdx.x == 0 && threadIdx.x == 0) buf[0] = __longlong_as_double(0x7FF8000000000000ll);
        return;
    }
    __threadfence_system();
    // ---- sum in rank order (own contribution straight from buf); every rank performs the same additions
    const double *slots = X.peer_slots[X.rank] + (size_t)par * X.world * X.slot_doubles;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)nb * blockDim.x) {
        double s = 0.0;
        for (int r = 0; r < X.world; r++) s += (r == X.rank) ? buf[i] : __builtin_nontemporal_load(slots + (size_t)r * X.slot_doubles + i);
        buf[i] = s;
    }
}

}  // namespace isdf

using namespace isdf;

struct isdf_xchg {
    int rank = 0, world = 1;
    size_t slot_doubles = 0;
    void *own = nullptr;                        // this rank's mailbox (uncached device memory): data, flags, then the board
    size_t own_bytes = 0, flags_offset = 0;
    size_t board_offset = 0, board_doubles = 0; // board of the in-kernel exchange (isdf_internal.hpp XFuse): 2 parities
    bool fuse_on = false;
    unsigned long long fuse_seq = 0;
    void *peer_base[XCHG_MAX_WORLD] = {nullptr};
    bool opened[XCHG_MAX_WORLD] = {false};
    unsigned *d_status = nullptr;               // [status word | pad | 64-bit push counter | 64-bit decision word]
    unsigned long long seq = 0;
    double ticks_per_ms = 1e5;                  // wall_clock64() rate of this device (hipDeviceAttributeWallClockRate, kHz): 100 MHz on gfx950
    bool connected = false;
};

static void xchg_free(isdf_ctx *c) {
    isdf_xchg *x = c->xchg;
    if (!x) return;
    for (int r = 0; r < x->world; r++)
        if (x->opened[r] && x->peer_base[r]) (void)hipIpcCloseMemHandle(x->peer_base[r]);
    if (x->own) (void)hipFree(x->own);
    if (x->d_status) (void)hipFree(x->d_status);
    delete x;
    c->xchg = nullptr;
}
void isdf_xchg_release(isdf_ctx *c) { xchg_free(c); }

extern "C" int isdf_xchg_create(isdf_ctx *c, int rank, int world, size_t max_doubles, void *handle_out64) {
    if (c && (!c->peers.empty() || c->is_peer)) return isdf_fail(c, ISDF_ERR_STATE, "a multi-device ctx (isdf_create_multi) exchanges by itself");
    if (!c) return ISDF_ERR_INVALID_ARG;
    if (world < 2 || world > XCHG_MAX_WORLD || rank < 0 || rank >= world || max_doubles == 0 || !handle_out64) return isdf_fail(c, ISDF_ERR_INVALID_ARG, "bad exchange setup");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    HIPCHK(c, hipSetDevice(c->device));
    xchg_free(c);
    isdf_xchg *x = new isdf_xchg();
    c->xchg = x;
    x->rank = rank; x->world = world;
    {
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device) == hipSuccess && khz > 0) x->ticks_per_ms = (double)khz;
        else (void)hipGetLastError();
    }
    x->slot_doubles = (max_doubles + 63) / 64 * 64;
    const size_t data_bytes = (size_t)2 * world * x->slot_doubles * sizeof(double);
    x->flags_offset = (data_bytes + 255) / 256 * 256;
    // board: XF_ROW slots per piece and parity; a packed vector of max_doubles holds at most max_doubles / 19 pieces
    x->board_offset = (x->flags_offset + (size_t)4 * world * sizeof(unsigned long long) + 255) / 256 * 256;      // data flags + verdict words
    x->board_doubles = (size_t)2 * XF_ROW * (max_doubles / 19 + 1);
    x->own_bytes = x->board_offset + x->board_doubles * sizeof(double);
    // uncached (fine-grained) device memory: peers' stores over xGMI become visible to this device's loads without cache games
    if (hipExtMallocWithFlags(&x->own, x->own_bytes, hipDeviceMallocUncached) != hipSuccess) { xchg_free(c); return isdf_fail(c, ISDF_ERR_HIP, "uncached mailbox allocation failed"); }
    HIPCHK(c, hipMemset(x->own, 0, x->own_bytes));
    HIPCHK(c, hipMemset((char *)x->own + x->board_offset, 0xFF, x->board_doubles * sizeof(double)));    // every slot empty
    HIPCHK(c, hipMalloc(&x->d_status, 32));
    HIPCHK(c, hipMemset(x->d_status, 0, 32));
    HIPCHK(c, hipDeviceSynchronize());
    hipIpcMemHandle_t h;
    if (hipIpcGetMemHandle(&h, x->own) != hipSuccess) { xchg_free(c); return isdf_fail(c, ISDF_ERR_HIP, "hipIpcGetMemHandle failed"); }
    std::memcpy(handle_out64, &h, 64);
    return ISDF_OK;
}

extern "C" int isdf_xchg_connect(isdf_ctx *c, const void *handles) {
    if (!c || !handles) return ISDF_ERR_INVALID_ARG;
    isdf_xchg *x = c->xchg;
    if (!x) return isdf_fail(c, ISDF_ERR_STATE, "isdf_xchg_create has not been called");
    HIPCHK(c, hipSetDevice(c->device));
    for (int r = 0; r < x->world; r++) {
        if (r == x->rank) { x->peer_base[r] = x->own; continue; }
        hipIpcMemHandle_t h;
        std::memcpy(&h, (const char *)handles + (size_t)64 * r, 64);
        void *p = nullptr;
        if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess || !p) {
            (void)hipGetLastError();
            return isdf_fail(c, ISDF_ERR_HIP, "hipIpcOpenMemHandle failed (peer memory not reachable)");
        }
        x->peer_base[r] = p; x->opened[r] = true;
    }
    x->connected = true;
    return ISDF_OK;
}

extern "C" int isdf_xchg_allreduce(isdf_ctx *c, double *d_buf, size_t count, void *stream) {
    if (!c || !d_buf) return ISDF_ERR_INVALID_ARG;
    isdf_xchg *x = c->xchg;
    if (!x || !x->connected) return isdf_fail(c, ISDF_ERR_STATE, "exchange not connected");
    if (count > x->slot_doubles) return isdf_fail(c, ISDF_ERR_INVALID_ARG, "vector longer than the exchange was created for");
    XchgParams X{};
    X.rank = x->rank; X.world = x->world; X.slot_doubles = x->slot_doubles; X.seq = ++x->seq; X.t_wait = (unsigned long long)(c->xchg_timeout_ms * x->ticks_per_ms); X.status = x->d_status; X.pushes = (unsigned long long *)((char *)x->d_status + 8); X.decision = (unsigned long long *)((char *)x->d_status + 16);
    for (int r = 0; r < x->world; r++) {
        X.peer_slots[r] = (double *)x->peer_base[r];
        X.peer_flags[r] = (unsigned long long *)((char *)x->peer_base[r] + x->flags_offset);
    }
    const int blocks = std::max(x->world - 1, (int)std::min<size_t>(8, (count + 1023) / 1024));
    hipLaunchKernelGGL(xchg_allreduce_kernel, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, X, d_buf, count);
    HIPCHK(c, hipGetLastError());
    return ISDF_OK;
}

// In-kernel exchange of fused steps (tile_sweep.hip, XFuse): on = 1 makes every following isdf_eval_device a COMPLETE
// multi-GPU step - its output is the full result on every rank, no isdf_xchg_allreduce afterwards.  All ranks must switch
// together and evaluate in lock-step (same launches in the same order).
extern "C" int isdf_xchg_fuse(isdf_ctx *c, int on) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    isdf_xchg *x = c->xchg;
    if (!x || !x->connected) return isdf_fail(c, ISDF_ERR_STATE, "exchange not connected");
    if (on && (x->rank != c->rank || x->world != c->world)) return isdf_fail(c, ISDF_ERR_STATE, "isdf_set_shard and the exchange disagree on rank / world");
    x->fuse_on = on != 0;
    return ISDF_OK;
}

bool isdf_xchg_fuse_on(const isdf_ctx *c) { return c->xchg && c->xchg->fuse_on; }

bool isdf_xchg_fill(isdf_ctx *c, isdf::XFuse *xf, size_t pieces, int *err) {
    *err = ISDF_OK;
    xf->world = 0;
    isdf_xchg *x = c->xchg;
    if (!x || !x->fuse_on) return false;
    if (!x->connected || x->rank != c->rank || x->world != c->world) { *err = isdf_fail(c, ISDF_ERR_STATE, "in-kernel exchange: shard and exchange disagree"); return true; }
    if ((size_t)2 * XF_ROW * pieces > x->board_doubles) { *err = isdf_fail(c, ISDF_ERR_INVALID_ARG, "in-kernel exchange: more pieces than the exchange was created for"); return true; }
    xf->world = x->world; xf->rank = x->rank; xf->parity = (int)(++x->fuse_seq & 1ull);
    xf->timeout_ticks = (int)std::min(2.0e9, c->xchg_timeout_ms * x->ticks_per_ms);
    xf->parity_stride = (unsigned long long)(x->board_doubles / 2);
    for (int r = 0; r < x->world; r++) xf->board[r] = (double *)((char *)x->peer_base[r] + x->board_offset);
    return true;
}

// After a bounded wait expired (overflow flag / NaN cost) a producer may still publish late into a slot nobody takes any
// more; the next step of that parity would consume the stale value.  Called by the host whenever it has seen the overflow
// word: the stream is drained and this rank's board is emptied again.  (The peers do the same when they see theirs; the
// in-kernel exchange should be switched off after an overflow - parallel.py does.)
void isdf_xchg_reset_board(isdf_ctx *c) {
    isdf_xchg *x = c->xchg;
    if (!x || !x->own) return;
    (void)hipDeviceSynchronize();
    (void)hipMemset((char *)x->own + x->board_offset, 0xFF, x->board_doubles * sizeof(double));
    (void)hipDeviceSynchronize();
}

// The bound T of every wait of the exchange, in milliseconds of the device's wall clock (default 2 000): how much later than this
// rank a peer may start an exchange before this rank calls it failed.  A failed exchange kernel ends after T on the rank whose
// wait expired and at once on a late rank; a healthy one waits at most 2.25 T for the verdicts.
extern "C" double isdf_xchg_timeout_ms(isdf_ctx *c) { return c ? c->xchg_timeout_ms : XCHG_DEFAULT_TIMEOUT_MS; }
extern "C" int isdf_xchg_set_timeout_ms(isdf_ctx *c, double ms) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    if (!(ms >= 1.0 && ms <= 10000.0)) return isdf_fail(c, ISDF_ERR_INVALID_ARG, "exchange time-out must be 1 ... 10 000 ms");
    c->xchg_timeout_ms = ms;
    return ISDF_OK;
}

extern "C" int isdf_xchg_status(isdf_ctx *c) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    isdf_xchg *x = c->xchg;
    if (!x) return ISDF_ERR_STATE;
    unsigned s = 0;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpy(&s, x->d_status, sizeof(s), hipMemcpyDeviceToHost));      // synchronises with the exchanges issued so far
    return s ? 1 : 0;
}

extern "C" int isdf_xchg_destroy(isdf_ctx *c) {
    if (!c) return ISDF_ERR_INVALID_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    (void)hipDeviceSynchronize();
    xchg_free(c);
    return ISDF_OK;
}
