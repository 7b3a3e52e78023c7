"""Host-side mirror of the reference's cost-callback plumbing over the C-ABI (include/isdf_accel.h).

``Engine`` plays the role TrajOptimizer's members play around the two sweeps
(src/planner_algorithm/include/planner_algorithm/back_end_optimizer.hpp:59-62, :386-405, :667-725):
setParam -> Engine(cfg), setGridMap -> set_grid, setEnvironment/initShape -> set_shape,
parallel_points -> set_points, and eval() == addSaftyPenaOnSweptVolumeParallel / addTimeIntPenaltyParallel
with the same ACCUMULATE semantics.  All arithmetic happens in libisdf_accel.so on the GPU; this file only
marshals numpy arrays / device pointers.  No CPU fallback exists: errors raise IsdfError.
"""
import ctypes as C
import numpy as np

from . import capi

_dp = C.POINTER(C.c_double)


class IsdfError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"isdf error {code}: {msg}")
        self.code = code


def _p(a):
    return a.ctypes.data_as(_dp)


def lbfgs_params(lib, **kw):
    p = capi.IsdfLbfgsParams()
    lib.isdf_lbfgs_params_default(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise TypeError(f"unknown L-BFGS parameter {k}")
        setattr(p, k, v)
    return p


def lbfgs_minimize(fun, x0, lib=None, progress=None, **params):
    """Driver on an arbitrary Python callback fun(x) -> (f, g) (host only; used by the CPU tests).
    progress(x, g, fx, step, k, ls) -> truthy cancels (isdf_lbfgs_minimize_progress: the reference's lbfgs_progress_t)."""
    lib = lib or capi.load_library()
    x = np.ascontiguousarray(x0, dtype=np.float64).copy()
    n = x.size

    def tramp(_inst, xp, gp, nn):
        xv = np.ctypeslib.as_array(xp, shape=(nn,))
        f, g = fun(xv.copy())
        np.ctypeslib.as_array(gp, shape=(nn,))[:] = g
        return float(f)
    cb = capi.EVALUATE_FN(tramp)
    p = lbfgs_params(lib, **params)
    r = capi.IsdfLbfgsResult()
    if progress is None:
        rc = lib.isdf_lbfgs_minimize(cb, None, _p(x), n, C.byref(p), C.byref(r))
    else:
        def ptramp(_inst, xp, gp, fx, step, k, ls):
            return 1 if progress(np.ctypeslib.as_array(xp, shape=(n,)).copy(), np.ctypeslib.as_array(gp, shape=(n,)).copy(), fx, step, k, ls) else 0
        pcb = capi.PROGRESS_FN(ptramp)
        rc = lib.isdf_lbfgs_minimize_progress(cb, pcb, None, _p(x), n, C.byref(p), C.byref(r))
    if rc != capi.ISDF_OK:
        raise IsdfError(rc, "isdf_lbfgs_minimize")
    return x, {"f": r.f, "status": r.status, "iterations": r.iterations, "evaluations": r.evaluations, "wall_ms": r.wall_ms}


class Engine:
    def __init__(self, cfg, lib=None, devices=None):
        """devices: None = one device (cfg.device); a list = ONE ctx over those devices (isdf_create_multi), used like any other."""
        self.lib = lib or capi.load_library()
        self.cfg = cfg
        h = C.c_void_p()
        if devices is None:
            rc = self.lib.isdf_create(C.byref(h), C.byref(cfg))
        else:
            dv = (C.c_int * len(devices))(*[int(d) for d in devices])
            rc = self.lib.isdf_create_multi(C.byref(h), C.byref(cfg), dv, len(devices))
        if rc != capi.ISDF_OK:
            msg = self.lib.isdf_last_error(None)
            raise IsdfError(rc, (msg or b"").decode())
        self.h = h
        self._keep = []

    def close(self):
        if getattr(self, "h", None):
            self.lib.isdf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != capi.ISDF_OK:
            msg = self.lib.isdf_last_error(self.h)
            raise IsdfError(rc, (msg or b"").decode())

    # ---- once-per-plan state
    def set_grid(self, vox, origin, res, kind, bmax=None):
        dt = {np.dtype(np.uint8): capi.U8, np.dtype(np.float32): capi.F32, np.dtype(np.float64): capi.F64}
        vox = np.ascontiguousarray(vox)
        o = np.asarray(origin, dtype=np.float64)
        bm = None if bmax is None else np.asarray(bmax, dtype=np.float64)
        self._check(self.lib.isdf_set_grid(self.h, vox.ctypes.data_as(C.c_void_p), dt[vox.dtype], vox.shape[0],
                                           vox.shape[1], vox.shape[2], _p(o), None if bm is None else _p(bm),
                                           float(res), kind))

    def set_shape(self, shape):
        self._keep.append(shape)
        self._check(self.lib.isdf_set_shape(self.h, C.byref(shape)))

    def set_shape_grid(self, cells, grid_min, nres, bound_radius=0.0, bbox=None):
        """ISDF_SHAPE_GRID: cells [nx, ny, nz, 4] = (unit gradient xyz, distance) per lattice node (BasicShape::num_sdf_map)."""
        cells = np.ascontiguousarray(cells, dtype=np.float64)
        nx, ny, nz, four = cells.shape
        assert four == 4
        gm = np.ascontiguousarray(grid_min, dtype=np.float64)
        bc = bh = None
        if bbox is not None:
            bc = np.ascontiguousarray(bbox[0], dtype=np.float64); bh = np.ascontiguousarray(bbox[1], dtype=np.float64)
        self._check(self.lib.isdf_set_shape_grid(self.h, _p(cells), nx, ny, nz, _p(gm), float(nres), float(bound_radius),
                                                 None if bc is None else _p(bc), None if bh is None else _p(bh)))

    def set_shape_sampled(self, fn, nd, nres, bound_radius=0.0):
        """fn(p[3]) -> (distance, gradient[3]): any host shape's getSDFwithGrad1; tabulated like BasicShape::initShape."""
        def tramp(_u, pp, gp):
            d, g = fn(np.array([pp[0], pp[1], pp[2]]))
            gp[0], gp[1], gp[2] = float(g[0]), float(g[1]), float(g[2])
            return float(d)
        cb = capi.SDF_WITH_GRAD_FN(tramp)
        self._check(self.lib.isdf_set_shape_sampled(self.h, cb, None, float(nd[0]), float(nd[1]), float(nd[2]), float(nres), float(bound_radius), None, None))

    def set_points(self, pts):
        pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 3)
        self._check(self.lib.isdf_set_points(self.h, _p(pts), pts.shape[0]))

    # ---- map products built on the device
    def set_pointcloud(self, xyz, res, sta_threshold=1, bmin=None, bmax=None):
        """xyz: n x 3 float32 (pcl::PointXYZ).  Returns the grid dimensions."""
        pts = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        dims = (C.c_int * 3)()
        bm0 = None if bmin is None else np.asarray(bmin, dtype=np.float64)
        bm1 = None if bmax is None else np.asarray(bmax, dtype=np.float64)
        self._check(self.lib.isdf_set_pointcloud(self.h, pts.ctypes.data_as(C.POINTER(C.c_float)), pts.shape[0],
                                                 None if bm0 is None else _p(bm0), None if bm1 is None else _p(bm1),
                                                 float(res), int(sta_threshold), dims))
        return tuple(dims)

    def generate_esdf(self):
        self._check(self.lib.isdf_generate_esdf(self.h))

    def get_grid(self, kind):
        dims = (C.c_int * 3)()
        o = np.zeros(3); bm = np.zeros(3)
        self._check(self.lib.isdf_get_grid(self.h, kind, None, capi.F32, dims, _p(o), _p(bm)))
        shape = tuple(dims)
        if kind == capi.GRID_ESDF:
            out = np.zeros(shape, dtype=np.float32); dt = capi.F32
        else:
            out = np.zeros(shape, dtype=np.uint8); dt = capi.U8
        self._check(self.lib.isdf_get_grid(self.h, kind, out.ctypes.data_as(C.c_void_p), dt, dims, _p(o), _p(bm)))
        return out, o, bm

    def gather_points(self, waypoints, half, offset=None):
        W = np.ascontiguousarray(np.asarray(waypoints, dtype=np.float64).reshape(-1, 3))
        h = np.asarray(half, dtype=np.float64) * np.ones(3)
        off = None if offset is None else np.asarray(offset, dtype=np.float64)
        M = C.c_int(0)
        self._check(self.lib.isdf_gather_points(self.h, _p(W) if W.size else None, W.shape[0], _p(h),
                                                None if off is None else _p(off), C.byref(M)))
        return M.value

    def get_points(self):
        M = self.lib.isdf_get_points(self.h, None, 0)
        if M < 0:
            self._check(M)
        out = np.zeros((M, 3))
        if M:
            rc = self.lib.isdf_get_points(self.h, _p(out), M)
            if rc < 0:
                self._check(rc)
        return out

    def shape_eval(self, p_rel, want_grad=True):
        """(sdf[n], grad[n, 3]) of the installed shape at body-frame points (getonlySDF / getonlyGrad1)."""
        p_rel = np.ascontiguousarray(p_rel, dtype=np.float64).reshape(-1, 3)
        n = p_rel.shape[0]
        sdf = np.zeros(n); grad = np.zeros((n, 3)) if want_grad else None
        self._check(self.lib.isdf_shape_eval(self.h, _p(p_rel), n, _p(sdf), _p(grad) if want_grad else None))
        return sdf, grad

    def esdf_sample(self, xyz, want_grad=True, scattered=False):
        """(value[n], grad[n, 3]) of the environment ESDF at world points (GridMap3D::getSDFValue / getSDFValueWithGrad);
        scattered: through the bricked copy (points in no particular order) - the same bits."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
        n = xyz.shape[0]
        val = np.zeros(n); grad = np.zeros((n, 3)) if want_grad else None
        fn = self.lib.isdf_esdf_sample_scattered if scattered else self.lib.isdf_esdf_sample
        self._check(fn(self.h, _p(xyz), n, _p(val), _p(grad) if want_grad else None))
        return val, grad

    def esdf_sample_device(self, d_xyz, n, d_value, d_grad=0, stream=0, scattered=False):
        """device pointers (ints), asynchronous on `stream`"""
        fn = self.lib.isdf_esdf_sample_scattered_device if scattered else self.lib.isdf_esdf_sample_device
        self._check(fn(self.h, C.c_void_p(d_xyz), n, C.c_void_p(d_value), C.c_void_p(d_grad), C.c_void_p(stream)))

    # ---- front end: pose feasibility by kernel convolution (SweptVolumeManager::checkKernelValue)
    def frontend_build(self, fe_cfg):
        self._fe_cfg = fe_cfg
        self._check(self.lib.isdf_frontend_build(self.h, C.byref(fe_cfg)))

    def frontend_shape_kernels(self):
        """ByteShapeKernel::map of every attitude (reference byte layout): uint8 [xk * yk, k * k * ceil(k / 8)]."""
        d = (C.c_int * 3)()
        self._check(self.lib.isdf_frontend_get_shape_kernels(self.h, None, d))
        out = np.zeros((d[0] * d[1], d[2]), dtype=np.uint8)
        self._check(self.lib.isdf_frontend_get_shape_kernels(self.h, out.ctypes.data_as(C.c_void_p), d))
        return out

    def frontend_map_kernel(self):
        """The array generateMapKernel produces (reference byte layout): uint8 [X + 2h, Y + 2h, ceil((Z + 2h) / 8)]."""
        d = (C.c_int * 3)()
        self._check(self.lib.isdf_frontend_get_map_kernel(self.h, None, d))
        out = np.zeros((d[0], d[1], d[2]), dtype=np.uint8)
        self._check(self.lib.isdf_frontend_get_map_kernel(self.h, out.ctypes.data_as(C.c_void_p), d))
        return out

    def frontend_check(self, index, father_roll, father_pitch):
        """(ok[n] uint8, child_roll[n], child_pitch[n], kernel_index[n]) of AstarGetSucc's per-neighbour test."""
        index = np.ascontiguousarray(index, dtype=np.int32).reshape(-1, 3)
        n = index.shape[0]
        fr = np.ascontiguousarray(np.broadcast_to(father_roll, (n,)), dtype=np.float64)
        fp = np.ascontiguousarray(np.broadcast_to(father_pitch, (n,)), dtype=np.float64)
        ok = np.zeros(n, dtype=np.uint8); cr = np.zeros(n); cp = np.zeros(n); ki = np.zeros(n, dtype=np.int32)
        self._check(self.lib.isdf_frontend_check(self.h, n, index.ctypes.data_as(C.c_void_p), _p(fr), _p(fp), ok.ctypes.data_as(C.c_void_p),
                                                 _p(cr), _p(cp), ki.ctypes.data_as(C.c_void_p)))
        return ok, cr, cp, ki

    def frontend_cspace(self, download=True):
        """(free_mask uint32 [X, Y, Z, 4 * ceil(attitudes / 128)] or None, kernel ms): bit (i * yk + j) of a voxel's mask = attitude fits."""
        dims = (C.c_int * 3)()
        o = np.zeros(3); bm = np.zeros(3)
        self._check(self.lib.isdf_get_grid(self.h, capi.GRID_OCCUPANCY, None, capi.U8, dims, _p(o), _p(bm)))
        kd = (C.c_int * 3)()
        self._check(self.lib.isdf_frontend_get_shape_kernels(self.h, None, kd))
        nw = 4 * ((kd[0] * kd[1] + 127) // 128)
        out = np.zeros((dims[0], dims[1], dims[2], nw), dtype=np.uint32) if download else None
        ms = C.c_double(0.0)
        self._check(self.lib.isdf_frontend_cspace(self.h, out.ctypes.data_as(C.c_void_p) if download else None, C.byref(ms)))
        return out, ms.value

    def host_info(self):
        """isdf_host_info: {"handovers", "late", "late_polls"} of the host-mapped result hand-overs of this ctx."""
        a = (C.c_int64 * 8)()
        self._check(self.lib.isdf_host_info(self.h, a))
        return {"handovers": int(a[0]), "late": int(a[1]), "late_polls": int(a[2])}

    def frontend_astar(self, start, goal):
        """AstarPathSearcher::AstarPathSearch + getPath + getastarSE3Path (front_end_Astar.hpp:238-403) over the device-built
        configuration space.  Returns (xyz (n, 3), roll/pitch degrees (n, 2), rot (n, 3, 3), IsdfAstarResult); the three arrays
        are None when success == 0."""
        s = np.ascontiguousarray(start, dtype=np.float64); g = np.ascontiguousarray(goal, dtype=np.float64)
        r = capi.IsdfAstarResult()
        self._check(self.lib.isdf_frontend_astar_search(self.h, _p(s), _p(g), C.byref(r)))
        if not r.success:
            return None, None, None, r
        n = r.n_path
        xyz = np.zeros((n, 3)); rp = np.zeros((n, 2)); rot = np.zeros((n, 3, 3))
        got = self.lib.isdf_frontend_astar_path(self.h, n, _p(xyz), _p(rp), _p(rot))
        if got != n:
            raise RuntimeError(f"isdf_frontend_astar_path returned {got}, the search said {n}")
        return xyz, rp, rot, r

    # ---- one-shot peer-to-peer exchange of the multi-GPU path (csrc/xchg.hip); see parallel.XgmiExchange
    def xchg_create(self, rank, world, max_doubles):
        h = (C.c_ubyte * 64)()
        self._check(self.lib.isdf_xchg_create(self.h, int(rank), int(world), int(max_doubles), h))
        return bytes(h)

    def xchg_connect(self, handles_bytes):
        buf = (C.c_ubyte * len(handles_bytes)).from_buffer_copy(handles_bytes)
        self._check(self.lib.isdf_xchg_connect(self.h, buf))

    def xchg_allreduce(self, d_ptr, count, stream=0):
        self._check(self.lib.isdf_xchg_allreduce(self.h, C.c_void_p(d_ptr), int(count), C.c_void_p(stream)))

    def xchg_fuse(self, on):
        """on: every following eval_device is a complete multi-GPU step (exchange inside the fused launch)."""
        self._check(self.lib.isdf_xchg_fuse(self.h, 1 if on else 0))

    def xchg_status(self):
        return int(self.lib.isdf_xchg_status(self.h))

    def xchg_timeout_ms(self):
        """bound of every wait of the exchange, milliseconds of the device wall clock"""
        return float(self.lib.isdf_xchg_timeout_ms(self.h))

    def xchg_set_timeout_ms(self, ms):
        self._check(self.lib.isdf_xchg_set_timeout_ms(self.h, float(ms)))

    def xchg_destroy(self):
        self._check(self.lib.isdf_xchg_destroy(self.h))

    def set_shard(self, rank, world):
        self._check(self.lib.isdf_set_shard(self.h, rank, world))

    # ---- per-step, host buffers (drop-in semantics: accumulate)
    def eval(self, T_list, coeffs_list, tstar=None, accumulate_into=None):
        """T_list[b]: N_b durations; coeffs_list[b]: 18*N_b doubles, column-major 6N x 3.
        Returns (cost[n_traj], [gradT_b], [gradC_b]); pass accumulate_into=(cost, gTs, gCs) to add into existing
        buffers exactly like the reference's += on cost / gradT / gradC."""
        n = len(T_list)
        Ts = [np.ascontiguousarray(t, dtype=np.float64) for t in T_list]
        Cs = [np.ascontiguousarray(c, dtype=np.float64).reshape(-1) for c in coeffs_list]
        Ns = (C.c_int * n)(*[t.size for t in Ts])
        for t, c in zip(Ts, Cs):
            if c.size != 18 * t.size:
                raise ValueError("coeffs must hold 18*N doubles (6N x 3 column-major)")
        if accumulate_into is None:
            cost = np.zeros(n)
            gTs = [np.zeros(t.size) for t in Ts]
            gCs = [np.zeros(18 * t.size) for t in Ts]
        else:
            cost, gTs, gCs = accumulate_into
        arr = lambda xs: (_dp * n)(*[_p(x) for x in xs])
        self._check(self.lib.isdf_eval(self.h, n, Ns, arr(Ts), arr(Cs), _p(cost), arr(gTs), arr(gCs),
                                       None if tstar is None else _p(tstar)))
        return cost, gTs, gCs

    def eval_single(self, T, coeffs_colmajor, tstar=None):
        cost, gT, gC = self.eval([T], [coeffs_colmajor], tstar=tstar)
        return float(cost[0]), gT[0], gC[0]

    # ---- per-step, device-resident (async on `stream`); arguments are raw device pointers (ints)
    def eval_device(self, n_traj, N, d_T, d_coeffs, d_out, d_tstar=0, stream=0):
        self._check(self.lib.isdf_eval_device(self.h, n_traj, N, C.c_void_p(d_T), C.c_void_p(d_coeffs),
                                              C.c_void_p(d_out), C.c_void_p(d_tstar), C.c_void_p(stream)))

    def eval_swept_at_tstar(self, N, d_T, d_coeffs, d_out, d_tstar, stream=0):
        """The swept-volume sweep's back-prop at GIVEN minimisers (device pointers; a negative t* = no interval)."""
        self._check(self.lib.isdf_eval_swept_at_tstar(self.h, N, C.c_void_p(d_T), C.c_void_p(d_coeffs), C.c_void_p(d_out),
                                                      C.c_void_p(d_tstar), C.c_void_p(stream)))

    def eval_swept_at_tstar_host(self, T, coeffs_colmajor, tstar):
        """Host arrays in, (cost, gradT, gradC) of the swept-volume sweep evaluated AT the given minimisers out."""
        T = np.ascontiguousarray(T, dtype=np.float64); N = T.size
        Cc = np.ascontiguousarray(coeffs_colmajor, dtype=np.float64).reshape(-1)
        ts = np.ascontiguousarray(tstar, dtype=np.float64)
        cost = np.zeros(1); gT = np.zeros(N); gC = np.zeros(18 * N)
        self._check(self.lib.isdf_eval_swept_at_tstar_host(self.h, N, _p(T), _p(Cc), _p(ts), _p(cost), _p(gT), _p(gC)))
        return float(cost[0]), gT, gC

    def out_stride(self, N):
        return int(self.lib.isdf_out_stride(N))

    # ---- full objective callback (TrajOptimizer::costFunctionLmbm)
    def set_trajectory(self, N, head_pva, tail_pva, rho):
        """head/tail: 3x3 arrays whose COLUMNS are position, velocity, acceleration (Eigen::Matrix3d of setConditions)."""
        h = np.ascontiguousarray(np.asarray(head_pva, dtype=np.float64).T).reshape(-1)   # column-major
        t = np.ascontiguousarray(np.asarray(tail_pva, dtype=np.float64).T).reshape(-1)
        self._check(self.lib.isdf_set_trajectory(self.h, int(N), _p(h), _p(t), float(rho)))
        self._N = int(N)

    def num_variables(self):
        return int(self.lib.isdf_num_variables(self.h))

    def pack_variables(self, T, waypoints):
        """T: N durations, waypoints: (N-1) x 3 -> x = [tau | xi]."""
        T = np.ascontiguousarray(T, dtype=np.float64)
        W = np.ascontiguousarray(np.asarray(waypoints, dtype=np.float64).reshape(-1))
        x = np.zeros(self.num_variables())
        self._check(self.lib.isdf_pack_variables(self.h, _p(T), _p(W) if W.size else None, _p(x)))
        return x

    def unpack_variables(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        T = np.zeros(self._N)
        cm = np.zeros(18 * self._N)
        self._check(self.lib.isdf_unpack_variables(self.h, _p(x), _p(T), _p(cm)))
        return T, cm

    def cost_function(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        g = np.zeros_like(x)
        cost = C.c_double(0)
        self._check(self.lib.isdf_cost_function(self.h, _p(x), _p(g), x.size, C.byref(cost)))
        return cost.value, g

    def cost_function_launch(self, x, stream=0):
        """Multi-GPU form, first half: returns (device pointer, count) of this rank's partial sums (to be all-reduced)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        self._cb_x = x
        ptr = C.c_void_p()
        cnt = C.c_size_t(0)
        self._check(self.lib.isdf_cost_function_launch(self.h, _p(x), x.size, C.c_void_p(stream), C.byref(ptr), C.byref(cnt)))
        return ptr.value, cnt.value

    def cost_function_finish(self, stream=0):
        g = np.zeros_like(self._cb_x)
        cost = C.c_double(0)
        self._check(self.lib.isdf_cost_function_finish(self.h, _p(g), C.byref(cost), C.c_void_p(stream)))
        return cost.value, g

    def set_minco_mode(self, mode):
        """capi.MINCO_AUTO (default): wherever it is faster; capi.MINCO_HOST: the host's band LU; capi.MINCO_DEVICE: the device kernels."""
        self._check(self.lib.isdf_set_minco_mode(self.h, int(mode)))

    def minco_path(self):
        """Where the last callback ran MINCO: 1 = device, 0 = host."""
        return int(self.lib.isdf_minco_path(self.h))

    def cost_parts(self):
        p = np.zeros(4)
        self._check(self.lib.isdf_cost_parts(self.h, _p(p)))
        return {"energy": p[0], "swept": p[1], "integral": p[2], "time": p[3]}

    # ---- optimizer driver (lbfgs::lbfgs_optimize behind the callback)
    def set_progress(self, fn, n_traj=1):
        """fn(traj, x, g, fx, step, k, ls) -> truthy cancels that trajectory (isdf_set_progress; None removes the hook).  In the batch
        driver the hook runs on the trajectories' host threads: trajectory t's instance is the t-th int of an array (stride 4)."""
        if fn is None:
            self._progress_keep = None
            self._check(self.lib.isdf_set_progress(self.h, None, None, 0))
            return
        n = self.num_variables()
        ids = (C.c_int * max(1, int(n_traj)))(*range(max(1, int(n_traj))))

        def tramp(inst, xp, gp, fx, step, k, ls):
            t = C.cast(inst, C.POINTER(C.c_int))[0]
            return 1 if fn(t, np.ctypeslib.as_array(xp, shape=(n,)).copy(), np.ctypeslib.as_array(gp, shape=(n,)).copy(), fx, step, k, ls) else 0
        cb = capi.PROGRESS_FN(tramp)
        self._progress_keep = (cb, ids)
        self._check(self.lib.isdf_set_progress(self.h, C.cast(cb, C.c_void_p), C.cast(ids, C.c_void_p), 4))

    def optimize_lbfgs(self, x0, **params):
        """Returns (x, result dict).  params override lbfgs_parameter_t defaults (mem_size, max_iterations, ...)."""
        x = np.ascontiguousarray(x0, dtype=np.float64).copy()
        p = lbfgs_params(self.lib, **params)
        r = capi.IsdfLbfgsResult()
        self._check(self.lib.isdf_optimize_lbfgs(self.h, _p(x), x.size, C.byref(p), C.byref(r)))
        return x, {"f": r.f, "status": r.status, "iterations": r.iterations, "evaluations": r.evaluations, "wall_ms": r.wall_ms}

    def optimize_lbfgs_batch(self, N, heads, tails, rho, x0s, **params):
        """heads / tails: n_traj x 3 x 3 (columns pos, vel, acc); x0s: n_traj x n.  Returns (xs, [result dicts], wall_ms)."""
        x = np.ascontiguousarray(x0s, dtype=np.float64).copy()
        n_traj = x.shape[0]
        h = np.ascontiguousarray(np.asarray(heads, dtype=np.float64).transpose(0, 2, 1)).reshape(-1)     # column-major 3x3 each
        t = np.ascontiguousarray(np.asarray(tails, dtype=np.float64).transpose(0, 2, 1)).reshape(-1)
        p = lbfgs_params(self.lib, **params)
        res = (capi.IsdfLbfgsResult * n_traj)()
        wall = C.c_double(0)
        self._check(self.lib.isdf_optimize_lbfgs_batch(self.h, n_traj, int(N), _p(h), _p(t), float(rho), _p(x), C.byref(p), res, C.byref(wall)))
        out = [{"f": r.f, "status": r.status, "iterations": r.iterations, "evaluations": r.evaluations, "rounds": r.reserved} for r in res]
        return x, out, wall.value

    # ---- instrumentation
    def mesh_info(self):
        """isdf_mesh_info as a dict (faces, nodes, depth, wg, closed, solid, lattice dims, measured |1 - 2w| range)"""
        a = (C.c_int * 16)()
        self._check(self.lib.isdf_mesh_info(self.h, a))
        return {"faces": a[0], "nodes": a[1], "depth": a[2], "wg": a[3], "closed": a[4], "solid": a[5], "lattice": (a[6], a[7], a[8]),
                "s_range": (a[9] * 1e-6, a[10] * 1e-6), "flat_slots": a[11],
                "defect_thickness": a[12] * 1e-9, "defect_s": a[13] * 1e-3}

    def multi_info(self):
        n = C.c_int(0); m = C.c_int(0)
        self._check(self.lib.isdf_multi_info(self.h, C.byref(n), C.byref(m)))
        return n.value, m.value

    def host_path(self):
        """How the last host-array step crossed PCIe (capi.HOST_PATH_*)."""
        return int(self.lib.isdf_host_path(self.h))

    def stats(self):
        s = capi.IsdfStats()
        self._check(self.lib.isdf_get_stats(self.h, C.byref(s)))
        return {"units": s.n_units, "culled": s.n_units_culled, "pairs": s.n_pairs,
                "grad_pairs": s.n_grad_pairs, "overflow": s.overflow}

    def profile_enable(self, every=1, secondary=False):
        """every = N > 0: instrument every N-th eval_device (dominant kernel; secondary=True: also the kernel after it);
        0/False: off."""
        every = int(every)
        self._check(self.lib.isdf_profile_enable(self.h, every | (0x10000 if (secondary and every > 0) else 0)))

    def profile_read(self):
        n = C.c_int(0)
        ms = C.c_double(0)
        self._check(self.lib.isdf_profile_read(self.h, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def profile_read_secondary(self):
        ms = C.c_double(0)
        self._check(self.lib.isdf_profile_read_secondary(self.h, C.byref(ms)))
        return ms.value
