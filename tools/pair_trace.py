"""Paired-store experiment: per-iteration scalars (mu, theta, phi, alpha, alpha_dual, delta_w, E0, trials) of a 600-instance batch on a FRESH handle,
dumped to an .npz -- run once with the default library and once with a -DMPC_EXP_PAIR build, then compare (tools/pair_trace.py cmp a.npz b.npz)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
if sys.argv[1] == "cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    ta, tb = a["trace"], b["trace"]
    n = min(len(ta), len(tb))
    names = ["mu", "theta", "phi", "alpha", "alpha_dual", "delta_w", "E0", "trials"]
    bad = np.flatnonzero(a["iters"] != b["iters"])
    print("instances with another iteration count:", len(bad), bad[:24].tolist())
    for i in bad[:6]:
        print(f"instance {i}: iterations {int(a['iters'][i])} / {int(b['iters'][i])}")
        for it in range(n):
            d = [q for q in range(8) if ta[it, q, i] != tb[it, q, i]]
            if d:
                print(f"   first difference at trace row {it}: " + ", ".join(f"{names[q]} {ta[it, q, i]:.6g} / {tb[it, q, i]:.6g}" for q in d))
                lo = max(0, it - 2)
                for r in range(lo, min(n, it + 2)):
                    print(f"      row {r}: mu {ta[r,0,i]:.4g}/{tb[r,0,i]:.4g} alpha {ta[r,3,i]:.6g}/{tb[r,3,i]:.6g} alpha_dual {ta[r,4,i]:.6g}/{tb[r,4,i]:.6g} E0 {ta[r,6,i]:.4g}/{tb[r,6,i]:.4g}")
                break
    sys.exit(0)
from helpers import FAMILIES, make_solver
from oracle.nlp_numpy import synthetic_batch
cfg, kw = FAMILIES["zamlf_n30_nx6"]
x0, p = synthetic_batch(cfg, 600, **kw)
s = make_solver(cfg)
r, tr = s.solve_trace(x0, p)
np.savez(sys.argv[1], trace=tr, iters=r.iters, x=r.x)
print(sys.argv[1], "mean iterations", float(r.iters.mean()), "trace rows", len(tr))
