"""The drop-in boundary as a C++ caller sees it (timed inside the compiled reference-side binding)."""
import os
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def native_boundary_ms(pkg, eng, N, T, cm, x0, eval_too=True):
    """isdf_eval and the lmbm_evaluate_t callback as a C++ caller sees them: timed inside the compiled reference-side
    binding (tests/native/binding_stub.cpp, built here with g++) - ctypes marshalling costs as much as the call itself.
    Returns (eval_ms, callback_ms) medians, or None when no compiler is at hand."""
    import ctypes as C
    try:
        out = os.path.join(tempfile.mkdtemp(), "libbinding_stub.so")
        libdir = os.path.dirname(pkg.capi.LIB_PATH)
        subprocess.check_call(["g++", "-O2", "-std=c++14", "-fPIC", "-shared", "-I", os.path.join(ROOT, "include"),
                               "-I", os.path.join(ROOT, "oracle", "ref_build", "eigen_shim"),
                               os.path.join(ROOT, "tests", "native", "binding_stub.cpp"), "-o", out,
                               "-L", libdir, "-lisdf_accel", f"-Wl,-rpath,{libdir}"], stderr=subprocess.DEVNULL)
        stub = C.CDLL(out)
    except Exception:
        return None
    dp = C.POINTER(C.c_double)
    stub.stub_time_eval.argtypes = [C.c_void_p, C.c_int, dp, dp, C.c_int, dp]
    stub.stub_time_callback.argtypes = [C.c_void_p, C.c_int, dp, C.c_int, dp, dp]
    lat = np.zeros(1000 if eval_too else 4)
    if stub.stub_time_eval(eng.h, N, T.ctypes.data_as(dp), cm.ctypes.data_as(dp), lat.size, lat.ctypes.data_as(dp)) != 0:
        return None
    cb = None
    if x0 is not None:
        cbl = np.zeros(500); cst = np.zeros(1)
        stub.stub_time_callback(eng.h, x0.size, x0.ctypes.data_as(dp), cbl.size, cbl.ctypes.data_as(dp), cst.ctypes.data_as(dp))
        cb = float(np.median(cbl)) * 1e-3
    return float(np.median(lat)) * 1e-3, cb
