"""Shader-clock stamps of one round of k_solve_wg (option res_timing).  Usage (GPU box): python tools/res_timing.py [B] [hybrid_bx] [hybrid_live]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from helpers import FAMILIES, make_solver
from oracle.nlp_numpy import synthetic_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg, kw = FAMILIES["zamlf_n30_nx6"]
x0, p = synthetic_batch(cfg, B, **kw)
s = make_solver(cfg)
s.set_option("hybrid", "1"); s.set_option("hybrid_bx", sys.argv[2] if len(sys.argv) > 2 else "1"); s.set_option("hybrid_live", sys.argv[3] if len(sys.argv) > 3 else "64")
s.set_option("timing", "4")
r = s.solve(x0, p); r = s.solve(x0, p)
print(s.get_resident_profile())
