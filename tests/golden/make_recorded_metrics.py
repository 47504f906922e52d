#!/usr/bin/env python3
"""recorded_metrics.npz: the post-hoc metric files of the reference's six recorded runs, as numbers -- per run `deviation.txt`
(mpc_planner.py:184-199: distance of every planned state to the nearest vertex of the ROUTE PLANNER's reference path) and `RMSD.txt`
(:279-292: against the resampled reference path).  Together with the recorded states (plant_step_kat.npz) they pin the scenario ->
reference-path pipeline (row f2): deviation.txt depends on every vertex of the origin path near the trajectory, RMSD.txt on every one of
the resampled reference points.  Reads /root/reference (build container only); the fixture is data."""
import os

import numpy as np

REF = "/root/reference/test"
OUT = os.path.dirname(os.path.abspath(__file__))
out = {}
for fw in ("casadi", "forcespro"):
    for name in ("ZAM_Over-1_1_lane_following", "ZAM_Over-1_1_collision_avoidance", "USA_Lanker-2_18_T-1_lane_following"):
        d = os.path.join(REF, f"2D_plots_{fw}_{name}")
        key = f"{fw}_{name}".replace("-", "_")
        out[key + "__deviation"] = np.loadtxt(os.path.join(d, "deviation.txt"))
        if os.path.exists(os.path.join(d, "RMSD.txt")):
            out[key + "__rmsd"] = np.loadtxt(os.path.join(d, "RMSD.txt"))
np.savez_compressed(os.path.join(OUT, "recorded_metrics.npz"), **out)
print({k: v.shape for k, v in out.items()})
