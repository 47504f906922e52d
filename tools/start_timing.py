"""Shader-clock stamps of k_start (option timing = 8): rows -> LDS, Z / REF stores, start-point safeguard, start iterate phase by phase.
Usage (GPU box): python tools/start_timing.py [B] [family]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from helpers import FAMILIES, make_solver, set_cfg_bounds
from oracle.nlp_numpy import synthetic_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
fam = sys.argv[2] if len(sys.argv) > 2 else "zamlf_n30_nx6"
cfg, kw = FAMILIES[fam]
x0, p = synthetic_batch(cfg, B, **kw)
s = make_solver(cfg)
set_cfg_bounds(s, cfg)
for _ in range(3):
    s.solve(x0, p)
s.set_option("timing", "8")
for _ in range(3):
    s.solve(x0, p)
