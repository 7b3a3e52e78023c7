"""Headless reproduction of the reference's demo pipeline around the hot path (test helper, not a test):
PCD + yaml + robot in -> optimized trajectory out, without ROS.

  map       PCSmapManager::rcvGlobalMapHandler (PCSmap_manager.cpp:87-200)      -> Engine.set_pointcloud (device)
  robot     SweptVolumeManager::initShape (sw_manager.hpp:239-275)               -> registry class / mesh from the config
  path      PlannerManager::generatePath (plan_manager.cpp:181-198): the SE(3) A* front end - AstarPathSearch + getPath
            (front_end_Astar.hpp:238-376) = Engine.frontend_build + Engine.frontend_astar (attitude kernels, bit-packed map and
            the whole configuration space on the device, the search over that table)
  waypoints PlannerManager::generateTraj (plan_manager.cpp:202-254): every index_gap-th path cell, obstacle points gathered in
            boxes of half size bd / 3 around them (Engine.gather_points, device), inittime per piece
  mid end   OUT OF SCOPE (no collision term): the waypoints themselves are the initial guess, like the first overload of
            optimize_traj_lmbm (back_end_optimizer.cpp:3-30)
  back end  lmbm::lmbm_optimize on costFunctionLmbm (back_end_optimizer.cpp:125-132): the reference's Fortran LMBM
            (oracle/_ref/liblmbm.so, test infrastructure) drives the callback under test.
"""
import math

import numpy as np


def plan_config_from_golden(pkg, g, tag):
    """isdf_plan_config from the yaml values committed in tests/golden/ref_demo_inputs.npz (tag: CappedCone | L | box)."""
    capi = pkg.capi
    p = capi.IsdfPlanConfig()
    capi.load_library().isdf_plan_config_default(p)
    s = p.sweep
    for k in ("kernel_size", "integral_intervs"):
        setattr(s, k, int(g[f"{tag}_yaml_{k}"]))
    for k in ("safety_hor", "weight_p", "weight_v", "weight_omg", "weight_theta", "vmax", "omgmax", "thetamax", "smoothing_eps",
              "vehicle_mass", "grav_acc", "horiz_drag", "vert_drag", "paras_drag", "speed_eps"):
        setattr(s, k, float(g[f"{tag}_yaml_{k}"]))
    p.occupancy_resolution = float(g[f"{tag}_yaml_occupancy_resolution"]); p.sta_threshold = int(g[f"{tag}_yaml_sta_threshold"])
    p.rho = float(g[f"{tag}_yaml_rho"]); p.inittime = float(g[f"{tag}_yaml_inittime"]); p.traj_parlength = float(g[f"{tag}_yaml_traj_parlength"])
    for i in range(6):
        p.poly_params[i] = float(g[f"{tag}_yaml_poly_params"][i])
    for i in range(3):
        p.offset_aabb[i] = float(g[f"{tag}_yaml_offset_aabb"][i])
    p.inputdata = str(g[f"{tag}_yaml_inputdata"]).encode()
    p.frontend.kernel_size = s.kernel_size
    return p


def build_plan(pkg, eng, plan, xyz, start_cell, goal_cell):
    """Map, path, waypoints, obstacle points, initial variables.  eng: an Engine of the plan's V1 config with the robot set."""
    capi = pkg.capi
    res = plan.occupancy_resolution
    dims = eng.set_pointcloud(xyz, res, plan.sta_threshold)
    occ, origin, bmax = eng.get_grid(capi.GRID_OCCUPANCY)
    eng.frontend_build(plan.frontend)                                                   # initShapeKernels + generateMapKernel (plan_manager.cpp:400-408)
    start = (np.array(start_cell) + 0.5) * res + origin; goal = (np.array(goal_cell) + 0.5) * res + origin
    path_xyz, path_rp, _, astar = eng.frontend_astar(start, goal)
    assert astar.success, "the front end found no path"
    path = [p for p in path_xyz]                                                        # recent_path: cube centres, start -> goal
    index_gap = math.ceil(plan.traj_parlength / res)                                    # plan_manager.cpp:206-213
    tpl = plan.traj_parlength
    while index_gap >= len(path) - 1:
        tpl /= 1.5
        index_gap = math.ceil(tpl / res)
    Q = [path[i] for i in range(index_gap, len(path) - 1, index_gap)]                   # :235-245
    bd = plan.sweep.kernel_size * res
    M = eng.gather_points(np.array(Q), bd / 3.0, offset=np.array(list(plan.offset_aabb)))
    N = len(Q) + 1
    head = np.zeros((3, 3)); head[:, 0] = path[0]
    tail = np.zeros((3, 3)); tail[:, 0] = path[-1]
    eng.set_trajectory(N, head, tail, plan.rho)
    T0 = np.full(N, plan.inittime)
    x0 = eng.pack_variables(T0, np.array(Q))
    return {"occ": occ, "origin": origin, "path": np.array(path), "path_rp": path_rp, "start": start, "goal": goal, "astar": astar, "Q": np.array(Q), "N": N, "M": M, "head": head, "tail": tail, "x0": x0,
            "points": eng.get_points()}
