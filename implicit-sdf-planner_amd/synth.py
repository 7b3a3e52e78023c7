"""Synthetic workloads for tests and bench.py (SURVEY.md §8(d) "Synthetic inputs").

Everything here is deterministic in its seeds and produces plain numpy arrays, so the GPU path and the CPU
oracle read identical bytes.  No reference code or oracle code is used: the ESDF comes from scipy's exact
Euclidean distance transform (which equals the reference's Felzenszwalb passes, src/map_manager/src/Gridmap3D.cpp:361-414,
because squared distances between voxel centres are exact integers — tests/test_oracle_selfcheck.py checks
that equality against the oracle's restatement), and MINCO coefficients come from a dense numpy solve of the
linear system of src/utils/include/utils/minco.hpp:433-513.
"""
import math
import numpy as np

from . import capi

# ---------------------------------------------------------------- shapes (reference constants, Shape.hpp)
_SHAPE_DEFAULTS = {
    "Torus": (capi.SHAPE_TORUS, [2.5, 0.3]),
    "Torus_big": (capi.SHAPE_TORUS, [3.5, 0.3]),
    "Cappedtorus": (capi.SHAPE_CAPPEDTORUS, [math.sin(40), math.cos(40), 3.5, 0.3]),
    "CappedCone": (capi.SHAPE_CAPPEDCONE, [2.0, 0.8, 0, 0, -1, 0, 0, 1]),
    "RoundedCone": (capi.SHAPE_ROUNDEDCONE, [1.5, 0.6, 4.5]),
    "WireframeBox": (capi.SHAPE_WIREFRAMEBOX, [1.8, 2.5, 3.5, 0.1]),
    "BendLinear": (capi.SHAPE_BENDLINEAR, [2.0, 0.25]),
    "BendLinear_big": (capi.SHAPE_BENDLINEAR, [3.2, 0.45]),
    "TwistBox": (capi.SHAPE_TWISTBOX, [2.0, 2.0, 2.0, 3.14159265358979323846 / 6]),
    "BendBox": (capi.SHAPE_BENDBOX, [2.0, 2.0, 2.0, 0.5]),
    "Table": (capi.SHAPE_TABLE, [0.0, 0.0, 0.0, 3.5, 1.75, 0.7, 2.8, 1.05, 0.0, 3.5, 1.75, 2.8]),
    "Trefoil": (capi.SHAPE_TREFOIL, [3.5, 0.2, 0.2, 0.05, 0.4]),
    "SmoothDifference": (capi.SHAPE_SMOOTHDIFFERENCE, [3.0, 3.0, 0.5, 1.0, 0.25]),
    "SmoothIntersection": (capi.SHAPE_SMOOTHINTERSECTION, [3.0, 3.0, 0.5, 1.0, 0.25]),
    "SmoothIntersection_big": (capi.SHAPE_SMOOTHINTERSECTION, [9.0, 9.0, 1.5, 3.0, 0.25]),
    "CSG": (capi.SHAPE_CSG, [3.0, 4.5, 1.5]),
    "Box": (capi.SHAPE_BOX, [3.0, 0.3, 0.3]),     # conf.box_x/y/z of config_CappedCone.yaml
    "Ball": (capi.SHAPE_BALL, [1.0]),
}
REGISTRY_NAMES = [n for n in _SHAPE_DEFAULTS if n not in ("Box", "Ball")]   # sw_manager.hpp:74-123 (16 entries)


def poly_rotation(roll_deg, pitch_deg, yaw_deg):
    """Rotate = yaw * pitch * roll (src/utils/src/Shape.cpp:38-43), angles in degrees with PI=3.14159265358979323846."""
    PI = 3.14159265358979323846
    r, p, y = (a * PI / 180.0 for a in (roll_deg, pitch_deg, yaw_deg))
    Rx = np.array([[1, 0, 0], [0, math.cos(r), -math.sin(r)], [0, math.sin(r), math.cos(r)]])
    Ry = np.array([[math.cos(p), 0, math.sin(p)], [0, 1, 0], [-math.sin(p), 0, math.cos(p)]])
    Rz = np.array([[math.cos(y), -math.sin(y), 0], [math.sin(y), math.cos(y), 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def make_shape(name, params=None, poly_params=(0, 0, 0, 0, 0, 0), grad_mode=capi.GRAD_DEFAULT, bound_radius=0.0,
               bbox=None):
    """isdf_shape for a registry name; poly_params = yaml [x, y, z, roll, pitch, yaw] (degrees)."""
    kind, defaults = _SHAPE_DEFAULTS[name]
    s = capi.IsdfShape()
    s.kind = kind
    s.grad_mode = grad_mode
    vals = list(defaults if params is None else params)
    for i, v in enumerate(vals):
        s.params[i] = float(v)
    for i in range(3):
        s.trans[i] = float(poly_params[i])
    R = poly_rotation(poly_params[3], poly_params[4], poly_params[5])
    for i in range(3):
        for j in range(3):
            s.rotate[3 * i + j] = float(R[i, j])
    s.bound_radius = float(bound_radius)
    if bbox is not None:        # (center xyz, half xyz) in the body frame
        for i in range(3):
            s.bbox_center[i] = float(bbox[0][i])
            s.bbox_half[i] = float(bbox[1][i])
    s.n_vertices = 0
    s.n_faces = 0
    return s


def make_mesh_shape(vertices, faces):
    """MESH kind.  The returned struct keeps references to the numpy arrays alive (attribute _keep)."""
    import ctypes as C
    V = np.ascontiguousarray(vertices, dtype=np.float64)
    F = np.ascontiguousarray(faces, dtype=np.int32)
    s = capi.IsdfShape()
    s.kind = capi.SHAPE_MESH
    s.grad_mode = capi.GRAD_DEFAULT
    for i in range(9):
        s.rotate[i] = 1.0 if i % 4 == 0 else 0.0
    s.mesh_vertices = V.ctypes.data_as(C.POINTER(C.c_double))
    s.mesh_faces = F.ctypes.data_as(C.POINTER(C.c_int32))
    s.n_vertices = V.shape[0]
    s.n_faces = F.shape[0]
    s._keep = (V, F)
    return s


def l_prism_mesh(a=1.6, b=1.0, t=0.4, h=0.5):
    """Closed, concave triangle mesh: an L-shaped prism (arms a and b long, t thick, height h), centred on its bounding
    box - the kind of robot demo6 loads (config_L.yaml, Lthick.obj: 20 triangles).  Returns (V 12x3, F 20x3), outward."""
    poly = np.array([[0, 0], [a, 0], [a, t], [t, t], [t, b], [0, b]], dtype=np.float64)     # counter-clockwise
    n = poly.shape[0]
    V = np.concatenate([np.c_[poly, np.full(n, -h / 2)], np.c_[poly, np.full(n, h / 2)]])
    V[:, 0] -= a / 2
    V[:, 1] -= b / 2
    F = []
    for i in range(n):                      # side walls
        j = (i + 1) % n
        F += [[i, j, n + j], [i, n + j, n + i]]
    tris = [[0, 1, 2], [0, 2, 3], [0, 3, 4], [0, 4, 5]]           # fan is valid for this L (vertex 0 sees every edge)
    for t3 in tris:
        F.append([n + t3[0], n + t3[1], n + t3[2]])                # top (+z)
        F.append([t3[0], t3[2], t3[1]])                            # bottom (-z)
    return V, np.array(F, dtype=np.int32)


def blob_mesh(nu, nv, seed=5, scale=1.0):
    """Closed triangle mesh with 2 * nu * (nv - 1) faces: a lobed, flattened UV sphere with jittered vertices - stands in
    for the reference's scanned robot meshes (2-12 k faces) at any size.  Returns (V, F), outward orientation."""
    rng = np.random.default_rng(seed)
    th = np.linspace(0, np.pi, nv + 1)[1:-1]
    ph = np.linspace(0, 2 * np.pi, nu, endpoint=False)
    P = [[0, 0, 1.0]] + [[np.sin(t) * np.cos(p) * (1 + 0.2 * np.cos(3 * p)), np.sin(t) * np.sin(p), np.cos(t) * 0.8]
                         for t in th for p in ph] + [[0, 0, -1.0]]
    P = (np.array(P) + rng.normal(0, 1e-3, (len(P), 3))) * scale
    F = []
    for j in range(nu):
        F.append([0, 1 + j, 1 + (j + 1) % nu])
    for i in range(nv - 2):
        for j in range(nu):
            a = 1 + i * nu + j; b = 1 + i * nu + (j + 1) % nu; c = a + nu; d = b + nu
            F += [[a, c, d], [a, d, b]]
    last = len(P) - 1
    base = 1 + (nv - 2) * nu
    for j in range(nu):
        F.append([last, base + (j + 1) % nu, base + j])
    return P, np.array(F, dtype=np.int32)


def bench_box_shape():
    """convex box, half-extents (1.6, 0.3, 0.3) m, the Box formula (Shape.hpp:2343-2351) used the
    way a plugin author would (Generalshape subclass + DEFINE_USEFUL_FUNCTION => central-difference gradient)."""
    half = (1.6, 0.3, 0.3)
    return make_shape("Box", params=half, grad_mode=capi.GRAD_CENTRAL,
                      bound_radius=math.sqrt(sum(h * h for h in half)), bbox=((0, 0, 0), half))


def bench_rounded_cone_shape():
    """0.4x-scaled RoundedCone (r1 0.6, r2 0.24, h 1.8)."""
    r1, r2, h = 0.6, 0.24, 1.8
    return make_shape("RoundedCone", params=(r1, r2, h), bound_radius=h + r2,
                      bbox=((0, 0, (h + r2 - r1) / 2), (r1, r1, (h + r2 + r1) / 2)))


# ---------------------------------------------------------------- config (demo yaml defaults)
def default_config(variant=capi.V3_ESDF_TILE, **kw):
    """config_CappedCone.yaml values unless overridden."""
    c = capi.IsdfConfig()
    c.device = 0
    c.variant = variant
    c.kernel_size = 13
    c.integral_intervs = 64
    c.enable_dyn = 1
    c.enable_pos = 1
    c.enable_cull = 0
    c.safety_hor = 0.866
    c.weight_p = 4000.0
    c.weight_v = c.weight_omg = c.weight_theta = 1000.0
    c.vmax = 10.0
    c.omgmax = 10.0
    c.thetamax = 100.0
    c.smoothing_eps = 1.0e-2
    c.occ_thresh = 0.0
    c.vehicle_mass = 0.61
    c.grav_acc = 9.8
    c.horiz_drag = 0.10
    c.vert_drag = 0.10
    c.paras_drag = 0.01
    c.speed_eps = 1.0e-4
    for k, v in kw.items():
        if not hasattr(c, k):
            raise AttributeError(k)
        setattr(c, k, v)
    return c


# ---------------------------------------------------------------- maps
def random_box_map(n=(256, 256, 256), res=0.2, occupancy=0.15, seed=12345, edge=(0.4, 2.0)):
    """Axis-aligned boxes with edges U[edge] m at uniform positions until ~`occupancy` of the voxels are occupied
    (the demo PCDs have 13.5-17 k of 91 k cells occupied).  Returns uint8 [nx,ny,nz], z fastest."""
    rng = np.random.default_rng(seed)
    nx, ny, nz = n
    occ = np.zeros(n, dtype=np.uint8)
    target = occupancy * occ.size
    count = 0
    ext = np.array(n) * res
    it = 0
    while count < target and it < 10_000_000:
        it += 1
        e = rng.uniform(edge[0], edge[1], 3)
        c = rng.uniform(0, 1, 3) * ext
        lo = np.clip(np.floor((c - e / 2) / res).astype(int), 0, np.array(n) - 1)
        hi = np.clip(np.floor((c + e / 2) / res).astype(int), 0, np.array(n) - 1)
        blk = occ[lo[0]:hi[0] + 1, lo[1]:hi[1] + 1, lo[2]:hi[2] + 1]
        count += int(blk.size - blk.sum())
        blk[...] = 1
    return occ


def esdf_from_occupancy(occ, res):
    """Unsigned ESDF in metres, float32: res * sqrt(squared index distance to the nearest occupied voxel)."""
    from scipy import ndimage
    if not occ.any():
        return np.full(occ.shape, np.float32(1e9), dtype=np.float32)
    d = ndimage.distance_transform_edt(occ == 0)
    return (res * d).astype(np.float32)


# ---------------------------------------------------------------- trajectories
def minco_coeffs(head, tail, in_pts, T):
    """MINCO s=3 coefficients, 6N x 3 (row r, col c), from the banded system of minco.hpp:433-513.
    head/tail: 3x3 with columns [pos vel acc]; in_pts: 3 x (N-1); T: N durations."""
    T = np.asarray(T, dtype=np.float64)
    N = T.size
    A = np.zeros((6 * N, 6 * N))
    b = np.zeros((6 * N, 3))
    A[0, 0] = 1.0
    A[1, 1] = 1.0
    A[2, 2] = 2.0
    b[0] = head[:, 0]
    b[1] = head[:, 1]
    b[2] = head[:, 2]
    for i in range(N - 1):
        t1 = T[i]; t2 = t1 * t1; t3 = t2 * t1; t4 = t2 * t2; t5 = t4 * t1
        r = 6 * i
        A[r + 3, r + 3] = 6.0; A[r + 3, r + 4] = 24.0 * t1; A[r + 3, r + 5] = 60.0 * t2; A[r + 3, r + 9] = -6.0
        A[r + 4, r + 4] = 24.0; A[r + 4, r + 5] = 120.0 * t1; A[r + 4, r + 10] = -24.0
        A[r + 5, r:r + 6] = [1.0, t1, t2, t3, t4, t5]
        A[r + 6, r:r + 6] = [1.0, t1, t2, t3, t4, t5]; A[r + 6, r + 6] = -1.0
        A[r + 7, r + 1:r + 6] = [1.0, 2 * t1, 3 * t2, 4 * t3, 5 * t4]; A[r + 7, r + 7] = -1.0
        A[r + 8, r + 2:r + 6] = [2.0, 6 * t1, 12 * t2, 20 * t3]; A[r + 8, r + 8] = -2.0
        b[r + 5] = in_pts[:, i]
    t1 = T[N - 1]; t2 = t1 * t1; t3 = t2 * t1; t4 = t2 * t2; t5 = t4 * t1
    r = 6 * N
    A[r - 3, r - 6:r] = [1.0, t1, t2, t3, t4, t5]
    A[r - 2, r - 5:r] = [1.0, 2 * t1, 3 * t2, 4 * t3, 5 * t4]
    A[r - 1, r - 4:r] = [2.0, 6 * t1, 12 * t2, 20 * t3]
    b[r - 3] = tail[:, 0]
    b[r - 2] = tail[:, 1]
    b[r - 1] = tail[:, 2]
    return np.linalg.solve(A, b)


def colmajor(coeffs_6Nx3):
    """Eigen::MatrixX3d memory image (column-major) of a [6N,3] numpy array."""
    return np.ascontiguousarray(np.asarray(coeffs_6Nx3, dtype=np.float64).T).reshape(-1)


def random_trajectory(extent, N, seed, piece_T=1.0, jitter=0.5, margin=3.0, occ=None, res=None):
    """Start/goal uniform in the interior (kept `margin` m off the walls, in free space when occ is given),
    straight-line waypoints + N(0, jitter) noise, uniform durations.  Returns (T[N], coeffs[6N,3])."""
    rng = np.random.default_rng(seed)
    ext = np.asarray(extent, dtype=np.float64)

    def free_point():
        for _ in range(1000):
            p = margin + rng.uniform(0, 1, 3) * (ext - 2 * margin)
            if occ is None:
                return p
            idx = np.floor(p / res).astype(int)
            if occ[idx[0], idx[1], idx[2]] == 0:
                return p
        return p

    p0, p1 = free_point(), free_point()
    lam = np.linspace(0, 1, N + 1)[1:-1]
    way = p0[:, None] + (p1 - p0)[:, None] * lam[None, :] + rng.normal(0, jitter, (3, N - 1))
    way = np.clip(way, margin * 0.5, (ext - margin * 0.5)[:, None])
    head = np.zeros((3, 3)); head[:, 0] = p0
    tail = np.zeros((3, 3)); tail[:, 0] = p1
    T = np.full(N, float(piece_T))
    return T, minco_coeffs(head, tail, way, T)


def constraint_points(occ, origin, res, waypoints, half, offset=(0, 0, 0)):
    """Obstacle-point set of the swept-volume sweep: union over waypoints of occupied voxel centres inside an AABB of
    half-size `half` (plan_manager.cpp:232-254 with getPointsInAABBOutOfLastOne, PCSmap_manager.h:182-216; the
    'out of last box' test only avoids duplicates, so a set union is the same set).  Sorted by voxel id (the
    reference's unordered_map iteration order is unspecified)."""
    n = np.array(occ.shape)
    origin = np.asarray(origin, dtype=np.float64)
    bmax = origin + n * res
    ids = set()
    for w in np.asarray(waypoints).reshape(-1, 3):
        c1 = np.clip(w - half + offset, origin, bmax)
        c2 = np.clip(w + half + offset, origin, bmax)
        lo = np.minimum(np.floor((c1 - origin) / res).astype(int), n - 1)
        hi = np.minimum(np.floor((c2 - origin) / res).astype(int), n - 1)
        sub = occ[lo[0]:hi[0] + 1, lo[1]:hi[1] + 1, lo[2]:hi[2] + 1]
        ii, jj, kk = np.nonzero(sub)
        for a, b_, c in zip(ii + lo[0], jj + lo[1], kk + lo[2]):
            ids.add((int(a) * n[1] + int(b_)) * n[2] + int(c))
    ids = np.array(sorted(ids), dtype=np.int64)
    if ids.size == 0:
        return np.zeros((0, 3))
    k = ids % n[2]
    j = (ids // n[2]) % n[1]
    i = ids // (n[1] * n[2])
    return (np.stack([i, j, k], axis=1) + 0.5) * res + origin
