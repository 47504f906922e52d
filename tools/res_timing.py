import os, sys
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import numpy as np
from helpers import FAMILIES, make_solver
from oracle.nlp_numpy import synthetic_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg, kw = FAMILIES["zamlf_n30_nx6"]
x0, p = synthetic_batch(cfg, B, **kw)
s = make_solver(cfg)
s.set_option("res_timing", "1")
r = s.solve(x0, p); r = s.solve(x0, p)
print(s.get_resident_profile())
