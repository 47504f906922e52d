"""Paired-store experiment with the workspace poisoned (option poison: NaN into rows [r0, r1) of every tile before each solve): which instances read
memory the solve has not written?  Usage (GPU box): python tools/pair_poison.py [r0 r1] [option=value ...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from helpers import FAMILIES, make_solver
from oracle.nlp_numpy import synthetic_batch
args = [a for a in sys.argv[1:] if "=" not in a]
opts = [a.split("=") for a in sys.argv[1:] if "=" in a]
r0, r1 = (int(args[0]), int(args[1])) if len(args) >= 2 else (0, 0)
cfg, kw = FAMILIES["zamlf_n30_nx6"]
x0, p = synthetic_batch(cfg, 600, **kw)
s = make_solver(cfg)
s.set_option("hybrid", "0")
for k, v in opts: s.set_option(k, v)
ref = s.solve(x0, p)
s.set_option("poison_r0", str(r0)); s.set_option("poison_r1", str(r1)); s.set_option("poison", "1")
for rep in range(2):
    a = s.solve(x0, p)
    bad = np.flatnonzero((a.status != 1) | ~np.isfinite(a.x).all(axis=1))
    diff = np.flatnonzero((a.x != ref.x).any(axis=1))
    print(f"rows [{r0}, {r1}) poisoned, {opts}: {len(bad)} instances not converged / not finite {bad[:12].tolist()}, statuses {sorted(set(a.status[bad].tolist()))}; "
          f"{len(diff)} rows differ from the unpoisoned solve {diff[:12].tolist()}, iteration counts differ on {int((a.iters != ref.iters).sum())}", flush=True)
