"""Fingerprints of the solver's output (rows, statuses, iteration counts) for several problem families and batch sizes: run with two builds of
the library (tools/ab_bench.sh copies lib*.so variants in place) and compare the lines -- a refactoring that only moves data must not change a bit.
Usage (GPU box): python tools/bits_dump.py [tag]"""
import hashlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from helpers import CA_CFG, FAMILIES, ca_batch, make_solver, set_cfg_bounds
from oracle.nlp_numpy import synthetic_batch

tag = sys.argv[1] if len(sys.argv) > 1 else ""
cases = [("zamlf_n30_nx6", 4096, {"hybrid": "0"}), ("zamlf_n30_nx6", 4096, {"pipeline": "0"}), ("zamlf_n30_nx6", 4096, {}), ("zamlf_n30_nx6", 256, {}), ("zamlf_n30_nx6", 777, {"hybrid_bx": "2"}), ("zamlf_n30_nx5", 1000, {}), ("usalf_n50_nx5", 2048, {}),
         ("zamlf_n10_nx5", 512, {}), ("ca", 1024, {}), ("ca", 300, {"hybrid_bx": "2"}), ("ca", 256, {"rescue": "0"}), ("zamlf_n30_nx6", 512, {"bound_mask": "0"})]
for fam, B, opts in cases:
    if fam == "ca":
        cfg = CA_CFG
        x0, p = ca_batch(cfg, B)
    else:
        cfg, kw = FAMILIES[fam]
        x0, p = synthetic_batch(cfg, B, **kw)
    s = make_solver(cfg)
    if fam == "ca":
        set_cfg_bounds(s, cfg)
    for k, v in opts.items():
        s.set_option(k, v)
    r = s.solve(x0, p)
    h = hashlib.sha256(r.x.tobytes() + r.status.tobytes() + r.iters.tobytes()).hexdigest()[:16]
    print(f"{tag:6s} {fam:14s} B={B:5d} {str(opts):24s} conv {np.mean(r.status == 1):.4f} iters {r.iters.mean():.4f}/{r.iters.max()} sha {h}", flush=True)
