"""An instance's result must not depend on its neighbours: with the hybrid solve off (option hybrid = 0: the pipeline runs every tile to the end; the
hybrid solve's hand-over makes the last bits depend on the batch's composition, by design) random windows of a pool against the pool solved at
once -- the same bits, statuses and iteration counts, whichever wavefront, tile, XCD or Riccati instantiation (symmetrised or plain) an instance
lands in.  Usage (GPU box): python tools/fuzz_windows.py [cases=80] [seed=1]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from helpers import CA_CFG, FAMILIES, ca_batch, make_solver, set_cfg_bounds
from oracle.nlp_numpy import synthetic_batch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
POOL = 6000
pools = {fam: (cfg, synthetic_batch(cfg, POOL, **kw)) for fam, (cfg, kw) in FAMILIES.items()}
pools["ca"] = (CA_CFG, ca_batch(CA_CFG, 2500))
ref, solv = {}, {}
for fam, (cfg, (X0, P)) in pools.items():
    s = make_solver(cfg)
    if fam == "ca":
        set_cfg_bounds(s, cfg)
    s.set_option("hybrid", "0")
    s.set_option("rescue", "0")                  # (the first attempt alone: the second chance runs on other kernels)
    ref[fam] = s.solve(X0, P)
    solv[fam] = s
    print(f"pool {fam}: {X0.shape[0]} instances, converged {float((ref[fam].status == 1).mean()):.4f}, pipeline {s.get_pipeline_profile()['ran']}", flush=True)
bad = 0
names = list(pools)
for c in range(n):
    fam = names[c % len(names)]
    cfg, (X0, P) = pools[fam]
    cap = X0.shape[0]
    B = int([1, 8, 63, 64, 65, 512, 1024, 1025, 2048][rng.integers(9)]) if rng.random() < 0.4 else int(rng.integers(1, cap + 1))
    B = min(B, cap)
    o = int(rng.integers(0, cap - B + 1))
    s = solv[fam]
    if rng.random() < 0.3:
        s.set_option("pipeline", "0")
    r = s.solve(np.ascontiguousarray(X0[o:o + B]), np.ascontiguousarray(P[o:o + B]))
    pl = s.get_pipeline_profile()["ran"]
    s.set_option("pipeline", "1")
    R = ref[fam]
    ok = np.array_equal(r.x, R.x[o:o + B]) and np.array_equal(r.status, R.status[o:o + B]) and np.array_equal(r.iters, R.iters[o:o + B])
    bad += not ok
    print(f"{'ok ' if ok else 'BAD'} {fam:14s} B={B:5d} off={o:5d} pipeline={int(pl)} |dx|={float(np.abs(r.x - R.x[o:o + B]).max()):.2e} iters differ on {int((r.iters != R.iters[o:o + B]).sum())}", flush=True)
print(f"problems: {bad} of {n}")
