"""Stress the pipeline's hand-offs: thousands of solves of the same batches, every result compared bit for bit with the
per-kernel path's (an intermittently stale read of another workgroup's rows would show up as a differing instance).
Usage (GPU box): python tools/pipe_stress.py [solves]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np  # noqa: E402
from helpers import CA_CFG, FAMILIES, ca_batch, make_solver, set_cfg_bounds  # noqa: E402
from oracle.nlp_numpy import synthetic_batch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
cases = []
for fam, B in (("zamlf_n30_nx6", 4096), ("zamlf_n30_nx6", 1000), ("zamlf_n30_nx5", 8192), ("usalf_n50_nx5", 2048), ("zamlf_n10_nx5", 3000), ("zamlf_n30_nx6", 16384)):
    cfg, kw = FAMILIES[fam]
    x0, p = synthetic_batch(cfg, B, **kw)
    cases.append((fam, make_solver(cfg), x0, p))
    cases[-1][1].set_option("hybrid", "0")        # pipeline against one launch per kernel: the hybrid solve rounds differently
x0, p = ca_batch(CA_CFG, 1024)
s = make_solver(CA_CFG)
s.set_option("hybrid", "0")
set_cfg_bounds(s, CA_CFG)
cases.append(("collision avoidance", s, x0, p))
bad = 0
t0 = time.time()
for fam, s, x0, p in cases:
    s.set_option("pipeline", "0")
    ref = s.solve(x0, p)
    s.set_option("pipeline", "1")
    reps = max(3, n // len(cases) if "collision" not in fam else n // 40)
    ran = 0
    for _ in range(reps):
        r = s.solve(x0, p)
        ran += s.get_pipeline_profile()["ran"]
        if not (np.array_equal(r.x, ref.x) and np.array_equal(r.iters, ref.iters) and np.array_equal(r.status, ref.status)):
            bad += 1
            d = np.nonzero((r.x != ref.x).any(axis=1))[0]
            print("MISMATCH", fam, "instances", d[:16], flush=True)
    print(f"{fam:22s} B={len(x0):5d}: {reps} solves, {ran} in the pipeline, mismatches so far {bad}", flush=True)
print("total mismatches", bad, "in", round(time.time() - t0, 1), "s")
sys.exit(1 if bad else 0)
