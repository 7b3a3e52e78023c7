"""ctypes mirror of include/isdf_accel.h (the C-ABI drop-in boundary).

Host-side plumbing only: structure layouts, enum values and the loader of libisdf_accel.so.
The library is the product; there is NO CPU fallback — if the shared object is missing, loading fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libisdf_accel.so")

# enums (keep in sync with include/isdf_accel.h)
ISDF_OK = 0
ISDF_ERR_INVALID_ARG, ISDF_ERR_NO_DEVICE, ISDF_ERR_HIP, ISDF_ERR_STATE, ISDF_ERR_OVERFLOW, ISDF_ERR_UNSUPPORTED = -1, -2, -3, -4, -5, -6
V1_SWEPT, V2_OCC_TILE, V3_ESDF_TILE = 1, 2, 3
GRID_OCCUPANCY, GRID_ESDF = 0, 1
U8, F32, F64 = 0, 1, 2
(SHAPE_TORUS, SHAPE_CAPPEDTORUS, SHAPE_CAPPEDCONE, SHAPE_ROUNDEDCONE, SHAPE_WIREFRAMEBOX, SHAPE_BENDLINEAR,
 SHAPE_TWISTBOX, SHAPE_BENDBOX, SHAPE_TABLE, SHAPE_TREFOIL, SHAPE_SMOOTHDIFFERENCE, SHAPE_SMOOTHINTERSECTION,
 SHAPE_CSG, SHAPE_BOX, SHAPE_BALL, SHAPE_MESH) = range(16)
GRAD_DEFAULT, GRAD_CENTRAL, GRAD_BOX_FORWARD, GRAD_ANALYTIC_BALL = 0, 1, 2, 3

SHAPE_NAMES = ["Torus", "Cappedtorus", "CappedCone", "RoundedCone", "WireframeBox", "BendLinear", "TwistBox",
               "BendBox", "Table", "Trefoil", "SmoothDifference", "SmoothIntersection", "CSG", "Box", "Ball", "Mesh"]


class IsdfShape(C.Structure):
    _fields_ = [("kind", C.c_int32), ("grad_mode", C.c_int32), ("params", C.c_double * 16),
                ("trans", C.c_double * 3), ("rotate", C.c_double * 9), ("bound_radius", C.c_double),
                ("bbox_center", C.c_double * 3), ("bbox_half", C.c_double * 3),
                ("mesh_vertices", C.POINTER(C.c_double)), ("mesh_faces", C.POINTER(C.c_int32)),
                ("n_vertices", C.c_int32), ("n_faces", C.c_int32)]


class IsdfConfig(C.Structure):
    _fields_ = [("device", C.c_int32), ("variant", C.c_int32), ("kernel_size", C.c_int32),
                ("integral_intervs", C.c_int32), ("enable_dyn", C.c_int32), ("enable_pos", C.c_int32),
                ("enable_cull", C.c_int32), ("reserved0", C.c_int32),
                ("safety_hor", C.c_double), ("weight_p", C.c_double), ("weight_v", C.c_double),
                ("weight_omg", C.c_double), ("weight_theta", C.c_double), ("vmax", C.c_double),
                ("omgmax", C.c_double), ("thetamax", C.c_double), ("smoothing_eps", C.c_double),
                ("occ_thresh", C.c_double), ("vehicle_mass", C.c_double), ("grav_acc", C.c_double),
                ("horiz_drag", C.c_double), ("vert_drag", C.c_double), ("paras_drag", C.c_double),
                ("speed_eps", C.c_double)]


class IsdfStats(C.Structure):
    _fields_ = [("n_units", C.c_int64), ("n_units_culled", C.c_int64), ("n_pairs", C.c_int64),
                ("n_grad_pairs", C.c_int64), ("overflow", C.c_int32), ("reserved", C.c_int32)]


class IsdfLbfgsParams(C.Structure):
    _fields_ = [("mem_size", C.c_int32), ("past", C.c_int32), ("max_iterations", C.c_int32), ("max_linesearch", C.c_int32),
                ("weak_wolfe", C.c_int32), ("reference_patches", C.c_int32),
                ("g_epsilon", C.c_double), ("delta", C.c_double), ("min_step", C.c_double), ("max_step", C.c_double),
                ("f_dec_coeff", C.c_double), ("s_curv_coeff", C.c_double), ("cautious_factor", C.c_double),
                ("machine_prec", C.c_double), ("dir_norm_cap", C.c_double)]


class IsdfFrontendConfig(C.Structure):
    _fields_ = [("kernel_size", C.c_int32), ("reserved", C.c_int32), ("kernel_max_roll", C.c_double), ("kernel_max_pitch", C.c_double),
                ("kernel_ang_res", C.c_double), ("front_end_safeh", C.c_double)]


def frontend_config(kernel_size=13, max_roll=45.0, max_pitch=45.0, ang_res=9.0, safeh=0.0):
    """The shipped configs' front-end values (config_*.yaml: kernel_max_roll 45, kernel_ang_res 9, front_end_safeh 0)."""
    c = IsdfFrontendConfig()
    c.kernel_size = int(kernel_size); c.kernel_max_roll = float(max_roll); c.kernel_max_pitch = float(max_pitch)
    c.kernel_ang_res = float(ang_res); c.front_end_safeh = float(safeh)
    return c


class IsdfAstarResult(C.Structure):
    """isdf_astar_result (include/isdf_accel.h)."""
    _fields_ = [("success", C.c_int32), ("n_path", C.c_int32), ("expansions", C.c_int64), ("checks", C.c_int64),
                ("cspace_ms", C.c_double), ("table_ms", C.c_double), ("search_ms", C.c_double)]


class IsdfPlanConfig(C.Structure):
    """isdf_plan_config: what a plan needs from the reference's yaml files (include/isdf_accel.h)."""
    _fields_ = [("sweep", IsdfConfig), ("frontend", IsdfFrontendConfig), ("occupancy_resolution", C.c_double),
                ("sta_threshold", C.c_int32), ("threads_num", C.c_int32), ("rho", C.c_double), ("inittime", C.c_double),
                ("momentum", C.c_double), ("traj_parlength", C.c_double), ("poly_params", C.c_double * 6),
                ("offset_aabb", C.c_double * 3), ("box", C.c_double * 3), ("map_bound", C.c_double * 6),
                ("inputdata", C.c_char * 256), ("pcdmapname", C.c_char * 128)]


class IsdfLbfgsResult(C.Structure):
    _fields_ = [("f", C.c_double), ("wall_ms", C.c_double), ("status", C.c_int32), ("iterations", C.c_int32),
                ("evaluations", C.c_int32), ("reserved", C.c_int32)]


EVALUATE_FN = C.CFUNCTYPE(C.c_double, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int)
PROGRESS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.c_double, C.c_int, C.c_int)   # isdf_progress_fn


# every symbol include/isdf_accel.h declares (tests check the .so exports all of them)
EXPORTED_SYMBOLS = [
    "isdf_config_default", "isdf_shape_default", "isdf_shape_from_name", "isdf_create", "isdf_destroy",
    "isdf_last_error", "isdf_abi_version", "isdf_set_grid", "isdf_set_shape", "isdf_set_points",
    "isdf_set_shard", "isdf_eval", "isdf_eval_device", "isdf_eval_swept_at_tstar", "isdf_eval_swept_at_tstar_host", "isdf_out_stride", "isdf_profile_enable",
    "isdf_profile_read", "isdf_profile_read_secondary", "isdf_get_stats",
    "isdf_set_trajectory", "isdf_num_variables", "isdf_pack_variables", "isdf_unpack_variables",
    "isdf_cost_function", "isdf_cost_function_lmbm", "isdf_cost_parts",
    "isdf_cost_function_launch", "isdf_cost_function_finish", "isdf_set_minco_mode", "isdf_minco_path",
    "isdf_lbfgs_params_default", "isdf_lbfgs_minimize", "isdf_optimize_lbfgs", "isdf_optimize_lbfgs_batch",
    "isdf_set_pointcloud", "isdf_generate_esdf", "isdf_get_grid", "isdf_gather_points", "isdf_get_points", "isdf_shape_eval",
    "isdf_esdf_sample", "isdf_esdf_sample_device", "isdf_esdf_sample_scattered", "isdf_esdf_sample_scattered_device",
    "isdf_frontend_build", "isdf_frontend_get_shape_kernels", "isdf_frontend_get_map_kernel", "isdf_frontend_check", "isdf_frontend_cspace",
    "isdf_frontend_astar_search", "isdf_frontend_astar_path", "isdf_host_info",
    "isdf_read_pcd", "isdf_read_obj", "isdf_poly_rotation", "isdf_body_transform", "isdf_plan_config_default", "isdf_load_yaml_config",
    "isdf_shape_from_config",
    "isdf_host_path", "isdf_mesh_atan2f", "isdf_create_multi", "isdf_multi_info", "isdf_set_shape_grid", "isdf_set_shape_sampled",
    "isdf_xchg_create", "isdf_xchg_connect", "isdf_xchg_allreduce", "isdf_xchg_fuse", "isdf_xchg_status", "isdf_xchg_destroy",
    "isdf_xchg_timeout_ms", "isdf_xchg_set_timeout_ms", "isdf_lbfgs_minimize_progress", "isdf_set_progress", "isdf_mesh_info",
]

HOST_PATH_COPY, HOST_PATH_DIRECT_MAPPED, HOST_PATH_DIRECT_BAR, HOST_PATH_DEVICE_CALLBACK = 0, 1, 2, 3
MINCO_AUTO, MINCO_HOST, MINCO_DEVICE = 0, 1, 2
MULTI_NONE, MULTI_PEER_SUM, MULTI_STAGED, MULTI_RCCL = 0, 1, 2, 3

SDF_WITH_GRAD_FN = C.CFUNCTYPE(C.c_double, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double))   # isdf_sdf_with_grad_fn
SHAPE_GRID, GRAD_GRID = 16, 4

_lib = None


def load_library(path=None):
    """dlopen libisdf_accel.so (built by __graft_entry__.build()).  Raises if it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("ISDF_ACCEL_LIB") or LIB_PATH      # ISDF_ACCEL_LIB: developer A/B of two builds on one box
    if not os.path.exists(p):
        raise RuntimeError(f"{p} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
                           "There is no CPU fallback for the product path.")
    lib = C.CDLL(p)
    dp = C.POINTER(C.c_double)
    lib.isdf_config_default.argtypes = [C.POINTER(IsdfConfig)]
    lib.isdf_config_default.restype = None
    lib.isdf_shape_default.argtypes = [C.POINTER(IsdfShape), C.c_int]
    lib.isdf_shape_from_name.argtypes = [C.POINTER(IsdfShape), C.c_char_p]
    lib.isdf_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(IsdfConfig)]
    lib.isdf_destroy.argtypes = [C.c_void_p]
    lib.isdf_create_multi.argtypes = [C.POINTER(C.c_void_p), C.POINTER(IsdfConfig), C.POINTER(C.c_int), C.c_int]
    lib.isdf_multi_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.isdf_last_error.argtypes = [C.c_void_p]
    lib.isdf_last_error.restype = C.c_char_p
    lib.isdf_set_grid.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, dp, dp, C.c_double, C.c_int]
    lib.isdf_set_shape.argtypes = [C.c_void_p, C.POINTER(IsdfShape)]
    lib.isdf_set_points.argtypes = [C.c_void_p, dp, C.c_int]
    lib.isdf_set_shape_grid.argtypes = [C.c_void_p, dp, C.c_int, C.c_int, C.c_int, dp, C.c_double, C.c_double, dp, dp]
    lib.isdf_set_shape_sampled.argtypes = [C.c_void_p, SDF_WITH_GRAD_FN, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, dp, dp]
    lib.isdf_set_shard.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.isdf_eval.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(dp), C.POINTER(dp), dp,
                              C.POINTER(dp), C.POINTER(dp), dp]
    lib.isdf_eval_device.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.isdf_eval_swept_at_tstar.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.isdf_eval_swept_at_tstar_host.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, dp, dp, dp]
    lib.isdf_out_stride.argtypes = [C.c_int]
    lib.isdf_out_stride.restype = C.c_size_t
    lib.isdf_profile_enable.argtypes = [C.c_void_p, C.c_int]
    lib.isdf_profile_read.argtypes = [C.c_void_p, C.POINTER(C.c_int), dp]
    lib.isdf_profile_read_secondary.argtypes = [C.c_void_p, dp]
    lib.isdf_get_stats.argtypes = [C.c_void_p, C.POINTER(IsdfStats)]
    lib.isdf_host_path.argtypes = [C.c_void_p]
    lib.isdf_mesh_atan2f.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_longlong, C.POINTER(C.c_float)]
    lib.isdf_set_trajectory.argtypes = [C.c_void_p, C.c_int, dp, dp, C.c_double]
    lib.isdf_num_variables.argtypes = [C.c_void_p]
    lib.isdf_pack_variables.argtypes = [C.c_void_p, dp, dp, dp]
    lib.isdf_unpack_variables.argtypes = [C.c_void_p, dp, dp, dp]
    lib.isdf_cost_function.argtypes = [C.c_void_p, dp, dp, C.c_int, dp]
    lib.isdf_cost_function_lmbm.argtypes = [C.c_void_p, dp, dp, C.c_int]
    lib.isdf_cost_function_lmbm.restype = C.c_double
    lib.isdf_cost_parts.argtypes = [C.c_void_p, dp]
    lib.isdf_cost_function_launch.argtypes = [C.c_void_p, dp, C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    lib.isdf_cost_function_finish.argtypes = [C.c_void_p, dp, dp, C.c_void_p]
    lib.isdf_set_minco_mode.argtypes = [C.c_void_p, C.c_int]
    lib.isdf_minco_path.argtypes = [C.c_void_p]
    ip = C.POINTER(C.c_int)
    lib.isdf_set_pointcloud.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_longlong, dp, dp, C.c_double, C.c_int, ip]
    lib.isdf_generate_esdf.argtypes = [C.c_void_p]
    lib.isdf_get_grid.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, ip, dp, dp]
    lib.isdf_gather_points.argtypes = [C.c_void_p, dp, C.c_int, dp, dp, ip]
    lib.isdf_get_points.argtypes = [C.c_void_p, dp, C.c_int]
    lib.isdf_shape_eval.argtypes = [C.c_void_p, dp, C.c_int, dp, dp]
    lib.isdf_esdf_sample.argtypes = [C.c_void_p, dp, C.c_longlong, dp, dp]
    lib.isdf_esdf_sample_device.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.isdf_esdf_sample_scattered.argtypes = [C.c_void_p, dp, C.c_longlong, dp, dp]
    lib.isdf_esdf_sample_scattered_device.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.isdf_frontend_build.argtypes = [C.c_void_p, C.POINTER(IsdfFrontendConfig)]
    lib.isdf_frontend_get_shape_kernels.argtypes = [C.c_void_p, C.c_void_p, ip]
    lib.isdf_frontend_get_map_kernel.argtypes = [C.c_void_p, C.c_void_p, ip]
    lib.isdf_xchg_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p]
    lib.isdf_xchg_connect.argtypes = [C.c_void_p, C.c_void_p]
    lib.isdf_xchg_allreduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.isdf_xchg_fuse.argtypes = [C.c_void_p, C.c_int]
    lib.isdf_xchg_status.argtypes = [C.c_void_p]
    lib.isdf_xchg_destroy.argtypes = [C.c_void_p]
    lib.isdf_xchg_timeout_ms.argtypes = [C.c_void_p]
    lib.isdf_xchg_timeout_ms.restype = C.c_double
    lib.isdf_xchg_set_timeout_ms.argtypes = [C.c_void_p, C.c_double]
    lib.isdf_frontend_cspace.argtypes = [C.c_void_p, C.c_void_p, dp]
    lib.isdf_host_info.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    lib.isdf_frontend_astar_search.argtypes = [C.c_void_p, dp, dp, C.POINTER(IsdfAstarResult)]
    lib.isdf_frontend_astar_path.argtypes = [C.c_void_p, C.c_int, dp, dp, dp]
    lib.isdf_frontend_check.argtypes = [C.c_void_p, C.c_int, C.c_void_p, dp, dp, C.c_void_p, dp, dp, C.c_void_p]
    lib.isdf_lbfgs_params_default.argtypes = [C.POINTER(IsdfLbfgsParams)]
    lib.isdf_lbfgs_params_default.restype = None
    lib.isdf_lbfgs_minimize.argtypes = [EVALUATE_FN, C.c_void_p, dp, C.c_int, C.POINTER(IsdfLbfgsParams), C.POINTER(IsdfLbfgsResult)]
    lib.isdf_lbfgs_minimize_progress.argtypes = [EVALUATE_FN, PROGRESS_FN, C.c_void_p, dp, C.c_int, C.POINTER(IsdfLbfgsParams), C.POINTER(IsdfLbfgsResult)]
    lib.isdf_mesh_info.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    lib.isdf_set_progress.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    lib.isdf_optimize_lbfgs_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, dp, dp, C.c_double, dp, C.POINTER(IsdfLbfgsParams),
                                              C.POINTER(IsdfLbfgsResult), dp]
    lib.isdf_optimize_lbfgs.argtypes = [C.c_void_p, dp, C.c_int, C.POINTER(IsdfLbfgsParams), C.POINTER(IsdfLbfgsResult)]
    lib.isdf_read_pcd.argtypes = [C.c_char_p, C.POINTER(C.c_float), C.c_longlong]
    lib.isdf_read_pcd.restype = C.c_longlong
    lib.isdf_read_obj.argtypes = [C.c_char_p, dp, C.c_int, C.POINTER(C.c_int32), C.c_int, ip, ip]
    lib.isdf_poly_rotation.argtypes = [dp, dp]
    lib.isdf_body_transform.argtypes = [dp, dp, C.c_int, dp, dp]
    lib.isdf_plan_config_default.argtypes = [C.POINTER(IsdfPlanConfig)]
    lib.isdf_plan_config_default.restype = None
    lib.isdf_load_yaml_config.argtypes = [C.c_char_p, C.POINTER(IsdfPlanConfig)]
    lib.isdf_shape_from_config.argtypes = [C.POINTER(IsdfShape), C.POINTER(IsdfPlanConfig), C.c_char_p, dp, C.c_int, C.POINTER(C.c_int32), C.c_int]
    if path is None:
        _lib = lib
    return lib
