"""Headless reproduction of the reference's demo pipeline around the hot path (test helper, not a test):
PCD + yaml + robot in -> optimized trajectory out, without ROS.

  map       PCSmapManager::rcvGlobalMapHandler (PCSmap_manager.cpp:87-200)      -> Engine.set_pointcloud (device)
  robot     SweptVolumeManager::initShape (sw_manager.hpp:239-275)               -> registry class / mesh from the config
  path      the reference's SE(3) A* front end is OUT OF SCOPE: a plain 26-connected A* over the occupancy grid
            inflated by the robot's radius stands in for it - it only has to produce a collision-free voxel path
  waypoints PlannerManager::generateTraj (plan_manager.cpp:202-254): every index_gap-th path cell, obstacle points gathered in
            boxes of half size bd / 3 around them (Engine.gather_points, device), inittime per piece
  mid end   OUT OF SCOPE (no collision term): the waypoints themselves are the initial guess, like the first overload of
            optimize_traj_lmbm (back_end_optimizer.cpp:3-30)
  back end  lmbm::lmbm_optimize on costFunctionLmbm (back_end_optimizer.cpp:125-132): the reference's Fortran LMBM
            (oracle/_ref/liblmbm.so, test infrastructure) drives the callback under test.
"""
import heapq
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


def astar(occ, start, goal, inflate):
    """26-connected A* on the voxel grid; a cell is blocked if an occupied cell lies within `inflate` cells (Chebyshev)."""
    from scipy.ndimage import maximum_filter
    blocked = maximum_filter(occ.astype(np.uint8), size=2 * inflate + 1, mode="constant", cval=1) > 0
    start, goal = tuple(start), tuple(goal)
    assert not blocked[start] and not blocked[goal], "start / goal not free"
    nb = [(a, b, c) for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1) if (a, b, c) != (0, 0, 0)]
    h = lambda p: math.dist(p, goal)
    openq = [(h(start), 0.0, start)]
    came, gs = {}, {start: 0.0}
    while openq:
        _, gcur, cur = heapq.heappop(openq)
        if cur == goal:
            path = [cur]
            while cur in came:
                cur = came[cur]; path.append(cur)
            return path[::-1]
        if gcur > gs.get(cur, 1e300):
            continue
        for d in nb:
            q = (cur[0] + d[0], cur[1] + d[1], cur[2] + d[2])
            if min(q) < 0 or q[0] >= occ.shape[0] or q[1] >= occ.shape[1] or q[2] >= occ.shape[2] or blocked[q]:
                continue
            ng = gcur + math.sqrt(d[0] ** 2 + d[1] ** 2 + d[2] ** 2)
            if ng < gs.get(q, 1e300):
                gs[q] = ng; came[q] = cur
                heapq.heappush(openq, (ng + h(q), ng, q))
    raise RuntimeError("no path")


def build_plan(pkg, eng, plan, xyz, start_cell, goal_cell, inflate):
    """Map, path, waypoints, obstacle points, initial variables.  eng: an Engine of the plan's V1 config with the robot set."""
    capi = pkg.capi
    res = plan.occupancy_resolution
    dims = eng.set_pointcloud(xyz, res, plan.sta_threshold)
    occ, origin, bmax = eng.get_grid(capi.GRID_OCCUPANCY)
    path_cells = astar(occ, start_cell, goal_cell, inflate)
    path = [(np.array(c) + 0.5) * res + origin for c in path_cells]                  # getGridCubeCenter
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
    return {"occ": occ, "origin": origin, "path": np.array(path), "Q": np.array(Q), "N": N, "M": M, "head": head, "tail": tail, "x0": x0,
            "points": eng.get_points()}
