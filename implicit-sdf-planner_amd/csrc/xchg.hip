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

constexpr long long XCHG_SPIN_LIMIT = 1ll << 22;       // x ~64 ns sleep + one flag load: ~0.5 s

struct XchgParams {
    int rank, world;
    size_t slot_doubles;                  // capacity of one slot
    unsigned long long seq;               // this exchange (1, 2, ...)
    double *peer_slots[XCHG_MAX_WORLD];   // base of rank r's mailbox data: [2 parities][world slots][slot_doubles]
    unsigned long long *peer_flags[XCHG_MAX_WORLD];   // rank r's flags: [2 parities][world] "data of exchange seq has arrived", then
                                                      // [2 parities][world] VERDICT words (seq << 1 | "my wait timed out")
    unsigned *status;                     // device word: != 0 after a time-out
    unsigned long long *pushes;           // device counter of completed pushes (world - 1 per exchange), local to this rank
    unsigned long long *decision;         // device word (seq << 1 | bad): block 0's verdict for exchange seq, read by this rank's other blocks
};

// grid: one block per PEER pushes (blockIdx.x < world - 1), then every block takes part in the wait + rank-order sum
__global__ __launch_bounds__(1024) void xchg_allreduce_kernel(XchgParams X, double *__restrict__ buf, size_t count) {
    const int par = (int)(X.seq & 1ull);
    const int nb = gridDim.x;
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
    // by another block, must have read it first) - all at once, one round trip instead of one per peer.  The other blocks have
    // no timer of their own: they wait for block 0's published decision (a per-block time-out could let ONE block give up
    // while block 0 - the only writer of the NaN cost - did not: a finite cost over a gradient slice that was never summed).
    __shared__ int s_ok;
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        const int lane = threadIdx.x;
        bool ok = true;
        if (lane < X.world && lane != X.rank) {
            const unsigned long long *mine = X.peer_flags[X.rank] + (size_t)par * X.world;
            long long spins = 0;
            while (__hip_atomic_load(mine + lane, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != X.seq) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > XCHG_SPIN_LIMIT) { ok = false; break; }
            }
        } else if (lane == X.world) {
            const unsigned long long want = X.seq * (unsigned long long)(X.world - 1);
            long long spins = 0;
            while (__hip_atomic_load(X.pushes, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > XCHG_SPIN_LIMIT) { ok = false; break; }
            }
        }
        bool all_ok = __ballot(!ok) == 0ull;
        // ---- verdict round: a time-out is one-sided by nature (THIS mailbox was late; the late peer itself may have received
        // everything), and ranks that disagree about a step would walk different optimizer paths.  So every rank tells every
        // peer what its wait saw - one more 8-byte store per peer and one more poll, inside this kernel, instead of a host-side
        // collective after every step - and the exchange counts as good only if EVERY rank's wait was.  A verdict that does
        // not arrive in time is a bad one.
        unsigned long long *const verdict_own = X.peer_flags[X.rank] + (size_t)(2 + par) * X.world;
        if (lane < X.world) {
            const unsigned long long word = (X.seq << 1) | (all_ok ? 0ull : 1ull);
            __hip_atomic_store(X.peer_flags[lane] + (size_t)(2 + par) * X.world + X.rank, word, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            long long spins = 0;
            unsigned long long v;
            while (((v = __hip_atomic_load(verdict_own + lane, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM)) >> 1) != X.seq) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > XCHG_SPIN_LIMIT) { v = 1ull; break; }
            }
            ok = (v & 1ull) == 0ull;
        } else ok = true;
        all_ok = __ballot(!ok) == 0ull;
        if (lane == 0) {
            if (!all_ok) atomicExch(X.status, 1u);
            s_ok = all_ok ? 1 : 0;
            // the rank's ONE decision for exchange seq, for its other blocks (release: they read the mailbox after it)
            __hip_atomic_store(X.decision, (X.seq << 1) | (all_ok ? 0ull : 1ull), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else if (blockIdx.x != 0 && threadIdx.x == 0) {
        // bounded like block 0's own waits (three of them, one after the other) plus slack: if block 0 of this launch never ran to
        // its decision (not resident, faulted), the other blocks must not hang the stream - they take the failure path and say so
        unsigned long long v;
        long long spins = 0;
        while (((v = __hip_atomic_load(X.decision, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) >> 1) != X.seq) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 4 * XCHG_SPIN_LIMIT) { v = 1ull; atomicExch(X.status, 1u); break; }
        }
        s_ok = (v & 1ull) == 0ull ? 1 : 0;
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
    X.rank = x->rank; X.world = x->world; X.slot_doubles = x->slot_doubles; X.seq = ++x->seq; X.status = x->d_status; X.pushes = (unsigned long long *)((char *)x->d_status + 8); X.decision = (unsigned long long *)((char *)x->d_status + 16);
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
    xf->world = x->world; xf->rank = x->rank; xf->parity = (int)(++x->fuse_seq & 1ull); xf->pad = 0;
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
