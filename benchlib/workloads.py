"""Synthetic workloads of SURVEY 8(d) shared by bench.py and the developer tools."""
import numpy as np


def build_workload(pkg, n_pieces, grid_n, res, seed_map=12345, seed_traj=777):
    synth = pkg.synth
    occ = synth.random_box_map((grid_n,) * 3, res=res, occupancy=0.15, seed=seed_map)
    esdf = synth.esdf_from_occupancy(occ, res)
    ext = np.array(occ.shape) * res
    T, Cf = synth.random_trajectory(ext, n_pieces, seed=seed_traj, piece_T=1.0, jitter=0.5, margin=4.0, occ=occ, res=res)
    return occ, esdf, T, synth.colmajor(Cf)


def ends_of(T, cm, N):
    """(head, tail, inner waypoints) of a trajectory given as (T, column-major coefficients): the boundary states isdf_set_trajectory
    takes (3 x 3, columns position / velocity / acceleration) and the (N - 1) x 3 waypoints isdf_pack_variables takes."""
    c6 = cm.reshape(3, 6 * N).T.reshape(N, 6, 3)
    head = np.stack([c6[0, 0], c6[0, 1], 2.0 * c6[0, 2]], axis=1)
    tl = T[-1]
    pw = lambda k: sum(np.prod([p - q for q in range(k)]) * c6[N - 1, p] * tl ** (p - k) for p in range(k, 6))      # noqa: E731
    return head, np.stack([pw(0), pw(1), pw(2)], axis=1), c6[1:, 0, :]
