import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from helpers import FAMILIES, make_solver, set_cfg_bounds
from oracle.nlp_numpy import synthetic_batch
fam, B = sys.argv[1], int(sys.argv[2])
cfg, kw = FAMILIES[fam]
x0, p = synthetic_batch(cfg, B, **kw)
for opts in [{"pipe_help": "0"}, {"pipe_help": "1"}, {"pipe_help": "1", "hybrid": "0"}, {"pipe_help": "1", "pipe_l2inv": "1"}, {"pipe_help": "1", "mb_pipe": "0"}, {"pipe_help": "1", "rescue": "0"}]:
    s = make_solver(cfg); set_cfg_bounds(s, cfg)
    for k, v in opts.items(): s.set_option(k, v)
    for rep in range(2):
        r = s.solve(x0, p)
        print(opts, "rep", rep, "piped", s.get_pipeline_profile()["ran"], "disabled", s.get_option("pipe_disabled"), "conv", float(np.mean(r.status == 1)), "iters", r.iters.mean(), flush=True)
