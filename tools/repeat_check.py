"""Run-to-run reproducibility of the default solve path: the headline batch (hybrid solve) and a collision-avoidance batch solved N times in ONE process
and in fresh handles -- every result must be the bits of the first.  Usage (GPU box): python tools/repeat_check.py [N]"""
import hashlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from helpers import CA_CFG, FAMILIES, ca_batch, make_solver, set_cfg_bounds
from oracle.nlp_numpy import synthetic_batch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for fam, B in (("zamlf_n30_nx6", 4096), ("ca", 1024), ("usalf_n50_nx5", 4096)):
    if fam == "ca":
        cfg = CA_CFG; x0, p = ca_batch(cfg, B)
    else:
        cfg, kw = FAMILIES[fam]; x0, p = synthetic_batch(cfg, B, **kw)
    hashes, means = set(), set()
    s = None
    for i in range(n):
        if i % 10 == 0:                       # a fresh handle (fresh workspace) every ten solves
            s = make_solver(cfg)
            if fam == "ca": set_cfg_bounds(s, cfg)
        r = s.solve(x0, p)
        hashes.add(hashlib.sha256(r.x.tobytes() + r.status.tobytes() + r.iters.tobytes()).hexdigest()[:16])
        means.add(round(float(r.iters.mean()), 6))
    print(f"{fam} B={B}: {n} solves, {len(hashes)} distinct result(s) {sorted(hashes)[:3]}, mean iterations {sorted(means)}", flush=True)
