"""The straggler launch of the hybrid solve (k_solve_wg behind k_pipeline) on the headline batch: how many instances it takes over, in how many
workgroups, rounds of the slowest workgroup, time -- and the histogram of iteration counts that explains them.
Usage (GPU box): python tools/wg_profile.py [B] [option=value ...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from helpers import FAMILIES, make_solver
from oracle.nlp_numpy import synthetic_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg, kw = FAMILIES["zamlf_n30_nx6"]
x0, p = synthetic_batch(cfg, B, **kw)
s = make_solver(cfg)
for e in sys.argv[2:]:
    k, v = e.split("="); s.set_option(k, v)
s.set_profiling(True)
for _ in range(3): r = s.solve(x0, p)
pp, rp = s.get_pipeline_profile(), s.get_resident_profile()
print("pipeline:", {k: (round(float(v), 4) if isinstance(v, (float, np.floating)) else v) for k, v in pp.items()})
print("k_solve_wg:", {k: (round(float(v), 4) if isinstance(v, (float, np.floating)) else v) for k, v in rp.items()})
it = r.iters
h = np.bincount(it)
print("iterations histogram:", {i: int(n) for i, n in enumerate(h) if n})
rounds_p = int(pp.get("rounds", 0))
for thr in range(rounds_p - 1, rounds_p + 3):
    live = it > thr
    pairs = live.reshape(-1, 2)
    print(f"  instances needing more than {thr} iterations: {int(live.sum())}; adjacent pairs with one / two such: {int((pairs.sum(1) == 1).sum())} / {int((pairs.sum(1) == 2).sum())}")
# where the long ones sit
long_ = np.flatnonzero(it >= it.max() - 2)
print("instances within 2 of the maximum:", [(int(i), int(it[i])) for i in long_][:20])
