"""Per-phase shader-clock stamps of the single-launch pipeline (k_pipeline), printed to stderr by the library:
the sixth work item of every stage worker and the sixth pass of every Riccati worker (option timing = 2).
Usage (GPU box): python tools/pipe_timing.py [B] [fixed_iters]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from helpers import FAMILIES, make_solver  # noqa: E402
from oracle.nlp_numpy import synthetic_batch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
fixed = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfg, kw = FAMILIES["zamlf_n30_nx6"]
x0, p = synthetic_batch(cfg, B, **kw)
s = make_solver(cfg, fixed_iters=fixed) if fixed else make_solver(cfg)
s.solve(x0, p)                                  # warm-up (allocations, first-touch)
s.set_option("timing", "2")
s.solve(x0, p)
s.solve(x0, p)
