import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_terminal_summary(terminalreporter):
    """The measured device-vs-oracle deviations of the mesh x V1 tests (tests/common.py::_spread_log), in the pytest log itself."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        import common
    except Exception:       # noqa: BLE001
        return
    if common.SPREAD_LINES:
        terminalreporter.write_sep("-", "mesh x V1: measured deviations (device vs oracle)")
        for l in common.SPREAD_LINES:
            terminalreporter.write_line(l)


@pytest.fixture(scope="session", autouse=True)
def _torch_first():
    """PyTorch bundles its own HIP runtime; when the product library (linked against /opt/rocm's) initialises the device
    first, torch's later initialisation reports "No HIP GPUs are available".  The tests that hand torch tensors to the library
    (exchange workers, bench) import torch first anyway; make the order deterministic for the whole session."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.zeros(1, device="cuda")
    except Exception:       # noqa: BLE001 - no torch / no GPU: nothing to order
        pass


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure).  Built by __graft_entry__.build(); rebuilt here if missing."""
    import __graft_entry__ as g
    lib = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(lib):
        import subprocess
        subprocess.check_call(["make", "liboracle.so"], cwd=os.path.join(ROOT, "oracle"))
    return g.load_oracle()


@pytest.fixture(scope="session")
def product_lib(pkg):
    """libisdf_accel.so; compiled on demand (hipcc cross-compiles without a GPU)."""
    if not os.path.exists(pkg.capi.LIB_PATH):
        import subprocess
        subprocess.check_call(["make", "-j8"], cwd=os.path.join(ROOT, "implicit-sdf-planner_amd", "csrc"))
    return pkg.capi.load_library()
