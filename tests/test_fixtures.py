"""The library's readers for the reference's OWN input files (csrc/fixtures.hip: ASCII .pcd, Wavefront .obj through the
Generalshape constructor's transform, the flat yaml configs) against independent numpy parsing of the files under
/root/reference/src/plan_manager, and the committed demo fixtures (tests/golden/ref_demo_inputs.npz, made by
tests/golden/make_golden_demo.py from those files) - CPU only, no device needed."""
import math
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PM = "/root/reference/src/plan_manager"
GOLD = os.path.join(ROOT, "tests", "golden", "ref_demo_inputs.npz")
need_ref = pytest.mark.skipif(not os.path.isdir(PM), reason="/root/reference not present (GPU box): the golden fixtures cover it")


def _eigen_rotate(pp):
    """yaw * pitch * roll with Eigen's AngleAxis::toRotationMatrix() for unit axes, independently restated in numpy."""
    PI = 3.14159265358979323846

    def aa(angle, axis):
        a = np.zeros(3); a[axis] = 1.0
        s, c = math.sin(angle), math.cos(angle)
        sa = s * a; c1 = (1.0 - c) * a
        R = np.zeros((3, 3))
        t = c1[0] * a[1]; R[0, 1] = t - sa[2]; R[1, 0] = t + sa[2]
        t = c1[0] * a[2]; R[0, 2] = t + sa[1]; R[2, 0] = t - sa[1]
        t = c1[1] * a[2]; R[1, 2] = t - sa[0]; R[2, 1] = t + sa[0]
        R[0, 0] = c1[0] * a[0] + c; R[1, 1] = c1[1] * a[1] + c; R[2, 2] = c1[2] * a[2] + c
        return R
    return aa(pp[5] * PI / 180.0, 2) @ aa(pp[4] * PI / 180.0, 1) @ aa(pp[3] * PI / 180.0, 0)


@need_ref
def test_pcd_reader_vs_numpy_and_the_voxel_grid_of_demo1(pkg, orc, product_lib):
    fx = pkg.fixtures
    path = os.path.join(PM, "map_pcds", "CappedCone.pcd")
    xyz = fx.read_pcd(path)
    ref = np.loadtxt(path, skiprows=11, dtype=np.float64).astype(np.float32)
    assert xyz.shape == ref.shape == (80107, 3) and np.array_equal(xyz, ref)
    # rcvGlobalMapHandler at the demo's 1 m voxels: the 51 x 51 x 35 grid SURVEY.md 5 measured, 13.5-17 k occupied cells
    occ, bmin, bmax = orc.pointcloud_to_occupancy(xyz, 1.0, 1)
    assert occ.shape == (51, 51, 35) and 13000 < int(occ.sum()) < 17500
    g = np.load(GOLD)
    assert np.array_equal(g["CappedCone_xyz"], xyz)
    for name in ("Torus_big", "Bendlinear_big", "SmoothIntersection"):
        p = os.path.join(PM, "map_pcds", name + ".pcd")
        a = fx.read_pcd(p)
        hdr = [l for l in open(p).read(2000).splitlines() if l.startswith("POINTS")][0]
        assert a.shape[0] == int(hdr.split()[1])


@need_ref
@pytest.mark.parametrize("name,nv,nf", [("Lthick", 60, 20), ("mybox", 8, 12), ("RoundedCone", None, None)])
def test_obj_reader_and_body_transform(pkg, product_lib, name, nv, nf):
    fx = pkg.fixtures
    path = os.path.join(PM, "shapes", name + ".obj")
    V, F = fx.read_obj(path)
    vs = [l.split()[1:4] for l in open(path) if l.startswith("v ")]
    fs = [[int(t.split("/")[0]) - 1 for t in l.split()[1:]] for l in open(path) if l.startswith("f ")]
    Vr = np.array(vs, dtype=np.float64)
    Fr = np.array([[f[0], f[j], f[j + 1]] for f in fs for j in range(1, len(f) - 1)], dtype=np.int32)
    if nv is not None:
        assert V.shape == (nv, 3) and F.shape == (nf, 3)
    assert np.array_equal(V, Vr) and np.array_equal(F, Fr)
    assert F.min() == 0 and F.max() == V.shape[0] - 1
    # Generalshape's constructor: V <- (V.homogeneous() * Trans^T).hnormalized()
    pp = np.array([0.3, -0.2, 0.1, 120.0, -35.0, 50.0])
    W, t, R = fx.body_transform(V, pp)
    Rn = _eigen_rotate(pp)
    assert np.abs(R - Rn).max() <= 2e-16 and np.array_equal(t, pp[:3])
    assert np.abs(W - (V @ Rn.T + pp[:3])).max() <= 1e-14 * (1.0 + np.abs(V).max())
    assert np.abs(R @ R.T - np.eye(3)).max() <= 1e-15
    # the identity offset of the shipped mesh configs leaves the vertices untouched
    W0, _, R0 = fx.body_transform(V, np.zeros(6))
    assert np.array_equal(W0, V) and np.array_equal(R0, np.eye(3))


@need_ref
def test_yaml_configs(pkg, product_lib):
    fx, capi = pkg.fixtures, pkg.capi
    p = fx.load_yaml_config(os.path.join(PM, "config", "config_CappedCone.yaml"))      # demo1.launch:22
    s = p.sweep
    assert (s.kernel_size, s.integral_intervs, p.sta_threshold, p.threads_num) == (13, 64, 1, 30)
    assert (s.safety_hor, s.weight_p, s.weight_v, s.weight_omg, s.weight_theta) == (0.866, 4000.0, 1000.0, 1000.0, 1000.0)
    assert (s.vmax, s.omgmax, s.thetamax, s.smoothing_eps) == (10.0, 10.0, 100.0, 1.0e-2)
    assert (s.vehicle_mass, s.grav_acc, s.horiz_drag, s.vert_drag, s.paras_drag, s.speed_eps) == (0.61, 9.8, 0.10, 0.10, 0.01, 0.0001)
    assert (p.rho, p.inittime, p.momentum, p.occupancy_resolution) == (20.0, 2.5, 0.0, 1.0)
    assert list(p.poly_params) == [0.0, 0.0, 0.0, 120.0, 0.0, 0.0] and list(p.offset_aabb) == [0.0, 0.0, 0.0]
    assert p.inputdata == b"shapes/RoundedCone.obj" and p.pcdmapname == b"CappedCone"
    assert (p.frontend.kernel_size, p.frontend.kernel_max_roll, p.frontend.kernel_max_pitch, p.frontend.kernel_ang_res, p.frontend.front_end_safeh) == (13, 45.0, 45.0, 9.0, 0.0)
    assert list(p.map_bound) == [-25.0, 25.0, -25.0, 25.0, 0.0, 15.0] and list(p.box) == [3.0, 0.3, 0.3]
    assert s.variant == capi.V1_SWEPT
    # the robot the config names: the analytic RoundedCone class (F5) with the 120 degree roll of poly_params
    shape = fx.shape_from_config(p, PM)
    assert shape.kind == capi.SHAPE_ROUNDEDCONE and list(shape.params)[:3] == [1.5, 0.6, 4.5]
    assert np.abs(np.array(list(shape.rotate)).reshape(3, 3) - _eigen_rotate(list(p.poly_params))).max() <= 2e-16
    # demo6 (config_L.yaml): a stem outside the registry -> the mesh Generalshape over the transformed obj
    pL = fx.load_yaml_config(os.path.join(PM, "config", "config_L.yaml"))
    assert pL.inputdata == b"shapes/Lthick.obj" and pL.sweep.kernel_size == 17 and pL.sweep.safety_hor == 0.6
    sL = fx.shape_from_config(pL, PM)
    assert sL.kind == capi.SHAPE_MESH and (sL.n_vertices, sL.n_faces) == (60, 20)
    g = np.load(GOLD)
    assert np.array_equal(g["Lthick_V"], sL._keep[0][:60]) and np.array_equal(g["Lthick_F"], sL._keep[1][:20])


def test_golden_demo_fixtures_are_self_consistent(pkg, orc):
    """What travels to the GPU box: the point cloud of demo1's map, the transformed vertices of demo6's robot and the yaml
    values - readable without /root/reference."""
    g = np.load(GOLD)
    assert g["CappedCone_xyz"].dtype == np.float32 and g["CappedCone_xyz"].shape == (80107, 3)
    occ, bmin, bmax = orc.pointcloud_to_occupancy(g["CappedCone_xyz"], 1.0, 1)
    assert occ.shape == (51, 51, 35)
    assert g["Lthick_V"].shape == (60, 3) and g["Lthick_F"].shape == (20, 3) and g["mybox_V"].shape == (8, 3)
    assert float(g["CappedCone_yaml_safety_hor"]) == 0.866 and int(g["CappedCone_yaml_kernel_size"]) == 13


def _write_pcd(path, fields, counts, rows, data="ascii"):
    with open(path, "w") as f:
        f.write("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\n")
        f.write("FIELDS " + " ".join(fields) + "\n")
        f.write("SIZE " + " ".join("4" for _ in fields) + "\nTYPE " + " ".join("F" for _ in fields) + "\n")
        if counts is not None:
            f.write("COUNT " + " ".join(str(c) for c in counts) + "\n")
        f.write(f"WIDTH {len(rows)}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {len(rows)}\nDATA {data}\n")
        for r in rows:
            f.write(" ".join(repr(float(v)) for v in r) + "\n")


def test_pcd_reader_honours_count(pkg, product_lib, tmp_path):
    """ADVICE (round 2): a field with COUNT > 1 occupies that many tokens of a data line (pcl's ascii reader walks fields x
    counts) - the columns of x / y / z shift accordingly; malformed COUNT lines are refused, not mis-read."""
    fx, capi = pkg.fixtures, pkg.capi
    rng = np.random.default_rng(5)
    xyz = rng.uniform(-10, 10, (37, 3)).astype(np.float32)
    nrm = rng.uniform(-1, 1, (37, 3)).astype(np.float32)
    # plain layout, with and without a COUNT line of ones
    for counts in (None, (1, 1, 1)):
        p = str(tmp_path / f"plain_{counts is None}.pcd")
        _write_pcd(p, ("x", "y", "z"), counts, xyz)
        assert np.array_equal(fx.read_pcd(p), xyz)
    # a 3-wide field in FRONT of the coordinates and a 2-wide one between y and z
    p = str(tmp_path / "wide.pcd")
    rows = [np.concatenate([n, [a[0], a[1]], [7.0, 8.0], [a[2]]]) for a, n in zip(xyz, nrm)]
    _write_pcd(p, ("normal", "x", "y", "pad", "z"), (3, 1, 1, 2, 1), rows)
    assert np.array_equal(fx.read_pcd(p), xyz)
    # refused: COUNT of a coordinate != 1 (unsupported), COUNT line of the wrong length / a zero count (invalid), binary data
    lib = product_lib
    def rc(path):
        return lib.isdf_read_pcd(path.encode(), None, 0)
    p = str(tmp_path / "xwide.pcd"); _write_pcd(p, ("x", "y", "z"), (2, 1, 1), [np.r_[a[0], a] for a in xyz]); assert rc(p) == capi.ISDF_ERR_UNSUPPORTED
    p = str(tmp_path / "short.pcd"); _write_pcd(p, ("x", "y", "z"), (1, 1), xyz); assert rc(p) == capi.ISDF_ERR_INVALID_ARG
    p = str(tmp_path / "zero.pcd"); _write_pcd(p, ("x", "y", "z", "w"), (1, 1, 1, 0), xyz); assert rc(p) == capi.ISDF_ERR_INVALID_ARG
    p = str(tmp_path / "bin.pcd"); _write_pcd(p, ("x", "y", "z"), None, xyz, data="binary"); assert rc(p) == capi.ISDF_ERR_UNSUPPORTED
