#!/usr/bin/env python
"""Generates tests/golden/ref_demo_inputs.npz: the INPUTS of the reference's demo1 / demo6 (they cannot travel to the GPU box
with /root/reference) as the PRODUCT's own readers deliver them - the point cloud of src/plan_manager/map_pcds/CappedCone.pcd
(float32, compressed), the vertices / faces of shapes/Lthick.obj and shapes/mybox.obj after the Generalshape constructor's
transform with their configs' poly_params, and the yaml values of config_CappedCone.yaml / config_L.yaml / config_box.yaml the
tests use.  tests/test_fixtures.py checks the readers against independent parsing of the files (here) and this file against
the readers.  Run in the build container: python tests/golden/make_golden_demo.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

PM = "/root/reference/src/plan_manager"
pkg = graft.load_package()
fx = pkg.fixtures
out = {"CappedCone_xyz": fx.read_pcd(os.path.join(PM, "map_pcds", "CappedCone.pcd"))}
for cfg_name, tag in (("config_CappedCone.yaml", "CappedCone"), ("config_L.yaml", "L"), ("config_box.yaml", "box")):
    p = fx.load_yaml_config(os.path.join(PM, "config", cfg_name))
    s = p.sweep
    for k in ("kernel_size", "integral_intervs", "safety_hor", "weight_p", "weight_v", "weight_omg", "weight_theta", "vmax", "omgmax",
              "thetamax", "smoothing_eps", "vehicle_mass", "grav_acc", "horiz_drag", "vert_drag", "paras_drag", "speed_eps"):
        out[f"{tag}_yaml_{k}"] = np.array(getattr(s, k))
    for k in ("occupancy_resolution", "sta_threshold", "rho", "inittime", "traj_parlength"):
        out[f"{tag}_yaml_{k}"] = np.array(getattr(p, k))
    out[f"{tag}_yaml_poly_params"] = np.array(list(p.poly_params))
    out[f"{tag}_yaml_offset_aabb"] = np.array(list(p.offset_aabb))
    out[f"{tag}_yaml_inputdata"] = np.array(p.inputdata.decode())
    stem = os.path.splitext(os.path.basename(p.inputdata.decode()))[0]
    shape = fx.shape_from_config(p, PM)
    if shape.kind == pkg.capi.SHAPE_MESH:
        out[f"{stem}_V"] = shape._keep[0][:shape.n_vertices].copy()
        out[f"{stem}_F"] = shape._keep[1][:shape.n_faces].copy()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_demo_inputs.npz"), **out)
print({k: (v.shape, str(v.dtype)) for k, v in out.items() if v.ndim})
