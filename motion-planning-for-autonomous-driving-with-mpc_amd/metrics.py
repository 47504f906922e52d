"""Post-hoc trajectory metrics of the reference's planner (scope row f4) on the GPU.

Mirrors `MPCPlanner.plot_deviation_euclidean_dis` (MPC_Planner/mpc_planner.py:184-199, the array it saves as
deviation.txt) and `MPCPlanner.compute_rmsd` (mpc_planner.py:279-292, RMSD.txt), without the plotting; plus the
clearance of the 3 x 3 approximation circles that the NLP constrains (optimizer.py:395-411).  Arithmetic on the device
through `mpc_metrics_batch` (include/mpcgpu.h); there is no CPU path.
"""
import numpy as np

from .solver import BatchedMPCSolver


def deviation_euclidean_dis(solver: BatchedMPCSolver, x, origin_reference_path):
    """deviation.txt of mpc_planner.py:184-199 for one trajectory (L,5) or a batch (B,L,5)."""
    x = np.asarray(x, dtype=np.float64)
    out = solver.metrics(x, origin_path=origin_reference_path)["deviation"]
    return out[0] if x.ndim == 2 else out


def compute_rmsd(solver: BatchedMPCSolver, x, reference_path):
    """(rmsd_x, rmsd_y) of mpc_planner.py:279-292 (divisor L - 1) for one trajectory or a batch."""
    x = np.asarray(x, dtype=np.float64)
    out = solver.metrics(x, ref_path=reference_path)["rmsd"]
    return out[0] if x.ndim == 2 else out


def min_clearance(solver: BatchedMPCSolver, x, r_sum, all_pairs=False):
    """min over steps and circle pairs of (centre distance - r_sum).  Default: the three pairs (ego circle j, obstacle
    circle j) the reference constrains (optimizer.py:395-403); all_pairs=True: all nine."""
    x = np.asarray(x, dtype=np.float64)
    out = solver.metrics(x, r_sum=r_sum, all_pairs=all_pairs)["clearance"]
    return out[0] if x.ndim == 2 else out


def collision_verdict(solver: BatchedMPCSolver, x, obstacles=None, left_boundary=None, right_boundary=None, ego_length=4.3, ego_width=1.8):
    """what test/test_mpc_planner.py:37-47 prints -- does the ego vehicle (mpc_planner.py:99: a 4.3 x 1.8 m rectangle on the planned
    states) collide with an obstacle of the scenario or with the road boundary? -- for one trajectory (L,5) or a batch (B,L,5).
    Returns (collides, first_collision_step, leaves_road, first_off_road_step); scenario.obstacle_rectangles / road_corridor build
    the obstacle rows and boundary polylines from a scenario."""
    x = np.asarray(x, dtype=np.float64)
    r = solver.validity(x, obstacles=obstacles, left=left_boundary, right=right_boundary, ego_length=ego_length, ego_width=ego_width)
    fc, fo = r["first_collision"], r["first_off_road"]
    if x.ndim == 2:
        return bool(fc[0] >= 0), int(fc[0]), bool(fo[0] >= 0), int(fo[0])
    return fc >= 0, fc, fo >= 0, fo


__all__ = ["deviation_euclidean_dis", "compute_rmsd", "min_clearance", "collision_verdict"]
