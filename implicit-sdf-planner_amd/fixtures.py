"""Host mirror of the library's readers for the reference's own input files (include/isdf_accel.h, csrc/fixtures.hip):
ASCII .pcd maps, Wavefront .obj robots through the Generalshape constructor's transform, the flat yaml configs.  No device
is needed for any of these."""
import ctypes as C
import os

import numpy as np

from . import capi

_dp = C.POINTER(C.c_double)


def read_pcd(path, lib=None):
    """float32 [n, 3]: the points of an ASCII .pcd as pcl::io::loadPCDFile<pcl::PointXYZ> yields them."""
    lib = lib or capi.load_library()
    n = lib.isdf_read_pcd(os.fsencode(path), None, 0)
    if n < 0:
        raise ValueError(f"isdf_read_pcd({path}): status {n}")
    xyz = np.zeros((n, 3), dtype=np.float32)
    got = lib.isdf_read_pcd(os.fsencode(path), xyz.ctypes.data_as(C.POINTER(C.c_float)), n)
    assert got == n
    return xyz


def read_obj(path, lib=None):
    """(V float64 [nV, 3], F int32 [nF, 3]) of a Wavefront .obj as igl::read_triangle_mesh yields them."""
    lib = lib or capi.load_library()
    nV = C.c_int(0); nF = C.c_int(0)
    rc = lib.isdf_read_obj(os.fsencode(path), None, 0, None, 0, C.byref(nV), C.byref(nF))
    if rc != 0:
        raise ValueError(f"isdf_read_obj({path}): status {rc}")
    V = np.zeros((nV.value, 3)); F = np.zeros((nF.value, 3), dtype=np.int32)
    rc = lib.isdf_read_obj(os.fsencode(path), V.ctypes.data_as(_dp), nV.value, F.ctypes.data_as(C.POINTER(C.c_int32)), nF.value, C.byref(nV), C.byref(nF))
    if rc != 0:
        raise ValueError(f"isdf_read_obj({path}): status {rc}")
    return V, F


def poly_rotation(poly_params, lib=None):
    lib = lib or capi.load_library()
    pp = np.ascontiguousarray(poly_params, dtype=np.float64); R = np.zeros(9)
    assert lib.isdf_poly_rotation(pp.ctypes.data_as(_dp), R.ctypes.data_as(_dp)) == 0
    return R.reshape(3, 3)


def body_transform(V, poly_params, lib=None):
    """Generalshape's constructor transform (Shape.cpp:37-49) of the vertices; returns (V', trans, Rotate)."""
    lib = lib or capi.load_library()
    pp = np.ascontiguousarray(poly_params, dtype=np.float64)
    W = np.ascontiguousarray(V, dtype=np.float64).copy(); t = np.zeros(3); R = np.zeros(9)
    assert lib.isdf_body_transform(pp.ctypes.data_as(_dp), W.ctypes.data_as(_dp), W.shape[0], t.ctypes.data_as(_dp), R.ctypes.data_as(_dp)) == 0
    return W, t, R.reshape(3, 3)


def load_yaml_config(path, lib=None):
    lib = lib or capi.load_library()
    p = capi.IsdfPlanConfig()
    rc = lib.isdf_load_yaml_config(os.fsencode(path), C.byref(p))
    if rc != 0:
        raise ValueError(f"isdf_load_yaml_config({path}): status {rc}")
    return p


def shape_from_config(plan, package_dir, lib=None, max_vertices=1 << 16, max_faces=1 << 17):
    """isdf_shape for the robot a config names (registry class with its body offset, or the mesh through the transform).
    The returned struct keeps the vertex / face arrays alive (attribute _keep)."""
    lib = lib or capi.load_library()
    s = capi.IsdfShape()
    V = np.zeros((max_vertices, 3)); F = np.zeros((max_faces, 3), dtype=np.int32)
    rc = lib.isdf_shape_from_config(C.byref(s), C.byref(plan), os.fsencode(package_dir), V.ctypes.data_as(_dp), max_vertices,
                                    F.ctypes.data_as(C.POINTER(C.c_int32)), max_faces)
    if rc != 0:
        raise ValueError(f"isdf_shape_from_config: status {rc}")
    s._keep = (V, F)
    return s
