"""Configuration 3 (1024 collision-avoidance cold starts): the instances that need the second chance -- status and iterations of the first attempt,
iterations accumulated over the levels, final status.  Usage (GPU box): python tools/ca_rescued.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from helpers import CA_CFG, ca_batch, make_solver, set_cfg_bounds
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
x0, p = ca_batch(CA_CFG, B)
s = make_solver(CA_CFG); set_cfg_bounds(s, CA_CFG)
s.set_option("rescue", "0")
a = s.solve(x0, p)
s.set_option("rescue", "1")
b = s.solve(x0, p)
idx = np.nonzero(a.status != 1)[0]
print(f"first attempt: {len(idx)} of {B} not converged; iterations of the converged ones: mean {a.iters[a.status == 1].mean():.2f} max {a.iters[a.status == 1].max()}")
for i in idx:
    print(f"  instance {i:4d}: first attempt status {a.status[i]:3d} after {a.iters[i]:3d} iterations (kkt {a.kkt[i]:.1e}); with the second chance status {b.status[i]} after {b.iters[i]} accumulated")
print("whole batch with the second chance: max iterations", b.iters.max(), "rescued", s.last_rescued())
