"""k_pipeline<.., HELP> (option pipe_help) against the plain pipeline on the batches where it is on by default: did the launch run to its end (a lost
stage item shows as an abandoned launch), same bits?  Usage (GPU box): python tools/help_check.py [reps]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from helpers import FAMILIES, make_solver, set_cfg_bounds
from oracle.nlp_numpy import synthetic_batch
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
bad = 0
for fam, B in (("usalf_n50_nx5", 2500), ("usalf_n50_nx5", 4096), ("zamlf_n30_nx5", 8192), ("zamlf_n30_nx6", 8192), ("zamlf_n30_nx6", 4096)):
    cfg, kw = FAMILIES[fam]
    x0, p = synthetic_batch(cfg, B, **kw)
    ref = None
    for help_ in ("0", "1"):
        s = make_solver(cfg); set_cfg_bounds(s, cfg)
        s.set_option("pipe_help", help_)
        ran = 0
        for rep in range(reps):
            r = s.solve(x0, p)
            ran += bool(s.get_pipeline_profile()["ran"])
            if ref is None: ref = r
            same = np.array_equal(r.x, ref.x) and np.array_equal(r.iters, ref.iters)
            bad += (not same)
        print(f"{fam} B={B} pipe_help={help_}: {ran} of {reps} solves in the pipeline, abandoned {s.get_option('pipe_aborts')}, same bits {same}", flush=True)
        bad += ran != reps
print("problems:", bad)
sys.exit(1 if bad else 0)
