import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from helpers import FAMILIES, make_solver
from oracle.nlp_numpy import synthetic_batch
cfg, kw = FAMILIES["zamlf_n30_nx6"]
x0, p = synthetic_batch(cfg, 600, **kw)
def run(opts, n=6):
    s = make_solver(cfg)
    s.set_option("hybrid", "0")
    ref = s.solve(x0, p)
    for k, v in opts: s.set_option(k, v)
    outs = [s.solve(x0, p) for _ in range(n)]
    for i, a in enumerate(outs):
        d = a.x != ref.x
        rows = np.flatnonzero(d.any(axis=1))
        print(opts, "solve", i, "rows differing from the pipeline's:", len(rows), "elements:", int(d.sum()), "iters differ:", int((a.iters != ref.iters).sum()),
              "max |dx|:", float(np.abs(a.x - ref.x).max()), "first rows:", rows[:8].tolist(), "same as solve 0:", bool(np.array_equal(a.x, outs[0].x)))
    return ref
r1 = run([("big_wg", "1")])
r2 = run([("pipeline", "0")])
r3 = run([("groups", "2")])
print("pipeline refs identical across handles:", np.array_equal(r1.x, r2.x), np.array_equal(r1.x, r3.x))
