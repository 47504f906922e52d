import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from helpers import FAMILIES, make_solver
from oracle.nlp_numpy import synthetic_batch
cfg, kw = FAMILIES["zamlf_n30_nx6"]
x0, p = synthetic_batch(cfg, 4096, **kw)
s = make_solver(cfg, fixed_iters=20)
s.solve(x0, p)
os.environ["MPCGPU_PIPE_TIMING"] = "1"
s.solve(x0, p); s.solve(x0, p)
