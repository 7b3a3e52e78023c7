"""Multi-GPU plumbing of the hot path: one process per GPU, constraint points sharded across ranks
with no data-path collective, then ONE all-reduce (RCCL over xGMI; gloo in the CPU tests) of the packed
[cost | gradT(N) | gradC(6N x 3)] vector per optimizer step.  torch.distributed is plumbing only."""
import os


def shard_range(total, rank, world):
    """Contiguous balanced partition — identical to isdf_set_shard's rule in csrc/isdf_host.hip."""
    q, r = divmod(total, world)
    b = rank * q + min(rank, r)
    return b, b + q + (1 if rank < r else 0)


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_process_group(backend, device=None):
    import torch.distributed as dist
    rank, _, world = env_rank_world()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    kw = {}
    if device is not None and backend == "nccl":
        kw["device_id"] = device
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def allreduce_packed(out_tensor):
    """Sum the ranks' partial [cost | gradT | gradC] buffers in place (a single collective per step)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(out_tensor, op=dist.ReduceOp.SUM)
    return out_tensor


def unpack_out(out, N):
    """(cost, gradT[N], gradC[18N col-major]) views of one trajectory's packed output."""
    return out[0], out[1:1 + N], out[1 + N:1 + 19 * N]


class _DeviceDoubles:
    """Zero-copy view of a raw device pointer for torch (__cuda_array_interface__)."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


def cost_function_allreduced(engine, x):
    """The full objective callback on a sharded engine (engine.set_shard(rank, world) done): MINCO + this rank's share of
    the sweeps, ONE all-reduce of the partial sums over RCCL, then the rest of the callback - the same (cost, g) on every
    rank.  Runs on torch's current stream so that the collective is ordered after the sweeps."""
    import torch
    stream = torch.cuda.current_stream().cuda_stream
    ptr, count = engine.cost_function_launch(x, stream)
    part = torch.as_tensor(_DeviceDoubles(ptr, count), device=torch.device("cuda", torch.cuda.current_device()))
    allreduce_packed(part)
    return engine.cost_function_finish(stream)
