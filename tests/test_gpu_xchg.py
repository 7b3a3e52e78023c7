"""One-shot peer-to-peer exchange of the multi-GPU path (csrc/xchg.hip) with several processes sharing the one GPU of the
test box: IPC mapping, sequence flags, double buffering and the rank-order sum are the same code that runs between the
GPUs of a node; what a single GPU cannot show is xGMI itself (parallel.XgmiExchange therefore verifies the path against the
group's all-reduce at start-up on the real node and stays on RCCL if anything disagrees)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _said(stdout, tag):
    """the lines a worker printed from its failure report on (stdout: a cut stderr tail once hid the only text there was)"""
    lines = stdout.splitlines()
    first = next((i for i, l in enumerate(lines) if tag in l and "FAILED" in l), None)
    return "\n".join(lines[first:first + 60]) if first is not None else "\n".join(lines[-20:])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 4])
def test_xchg_processes_sharing_one_gpu(product_lib, world):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "native", "xchg_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    if not (r.returncode == 0 and "XCHG WORKERS OK" in r.stdout):
        print(_said(r.stdout, "[xchg_worker]"))          # the workers' own report, whole (an assertion's repr is cut)
    assert r.returncode == 0 and "XCHG WORKERS OK" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])


def test_xchg_failure_is_agreed_in_the_kernel(product_lib):
    """A bounded wait that expires on ONE rank is known to ALL ranks when the exchange kernel ends (verdict round, csrc/xchg.hip):
    no per-step host collective, and the callback's redo on the group's all-reduce is entered by every rank together."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "native", "xchg_fail_worker.py")]
    import time
    t0 = time.perf_counter()
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    wall = time.perf_counter() - t0
    said = "\n".join(l for l in r.stdout.splitlines() if "[xchg_fail_worker]" in l or "Error" in l or "assert" in l)
    if r.returncode != 0:
        said += "\n" + _said(r.stdout, "[xchg_fail_worker]")
    print(f"xchg failure test: {wall:.1f} s wall\n{said}")         # (-s / the failure report show the workers' own lines, the kernel's duration among them)
    assert r.returncode == 0 and "XCHG FAILURE AGREED OK" in r.stdout, (said, r.stdout[-3000:], r.stderr[-3000:])
    assert "rank 0: failed exchange took" in r.stdout and "rank 1: failed exchange took" in r.stdout
