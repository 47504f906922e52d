"""KKT error of ONE collision-avoidance instance iteration by iteration on the two KKT solvers (the MFMA sweep of k_solve_wg, the lane sweep of
the streaming path): the instance is solved alone with iteration limits 1, 2, 3, ...  Usage (GPU box): python tools/ca_trace.py INSTANCE [last]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from helpers import CA_CFG, ca_batch, make_solver, set_cfg_bounds
i = int(sys.argv[1]); last = int(sys.argv[2]) if len(sys.argv) > 2 else 60
x0, p = ca_batch(CA_CFG, 1, start=i)
rows = []
for k in range(1, last + 1):
    out = []
    for hyb in ("1", "0"):
        s = make_solver(CA_CFG, max_iter=k); set_cfg_bounds(s, CA_CFG)
        s.set_option("rescue", "0"); s.set_option("hybrid", hyb)
        r = s.solve(x0, p)
        out.append((int(r.status[0]), int(r.iters[0]), float(r.kkt[0])))
        s.close()
    rows.append(out)
    print(f"{k:3d}  MFMA: status {out[0][0]:3d} it {out[0][1]:3d} E0 {out[0][2]:.3e}   lane: status {out[1][0]:3d} it {out[1][1]:3d} E0 {out[1][2]:.3e}", flush=True)
    if out[0][0] == 1 and out[1][0] == 1: break
