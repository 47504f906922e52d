import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from helpers import CA_CFG, ca_batch, make_solver, set_cfg_bounds
from oracle.binding import OracleSolver
B = 1024
x0, p = ca_batch(CA_CFG, B)
s = make_solver(CA_CFG); set_cfg_bounds(s, CA_CFG)
s.set_option("rescue", "0")
a = s.solve(x0, p)
s.set_option("hybrid", "0")
b = s.solve(x0, p)
ro = OracleSolver(CA_CFG).solve_batch(x0, p, nthreads=16)
oi = ro["iters"]; os_ = ro["status"]
for name, r in (("k_solve_wg (MFMA sweep)", a), ("pipeline (lane sweep)", b)):
    print(name, "mean", r.iters.mean(), "top", sorted(r.iters.tolist())[-8:], "not converged", int((r.status != 1).sum()))
print("oracle mean", oi.mean(), "top", sorted(oi.tolist())[-8:], "not converged", int((os_ != 1).sum()))
idx = np.argsort(-a.iters)[:8]
print("slowest of the MFMA path:", [(int(i), int(a.iters[i]), int(b.iters[i]), int(oi[i])) for i in idx], "(instance, wg, pipeline, oracle)")
