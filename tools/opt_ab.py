"""A/B of the values of ONE run-time option inside one process (same box, same handle, alternating blocks of solves): medians of the batch
time and of the two launches of the iteration loop, and whether the results are the same bits.
Usage (GPU box): python tools/opt_ab.py OPTION V0 V1 [B] [family] [extra=value ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
from helpers import CA_CFG, FAMILIES, ca_batch, make_solver, set_cfg_bounds
from oracle.nlp_numpy import synthetic_batch

opt, vals = sys.argv[1], sys.argv[2:4]
B = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
fam = sys.argv[5] if len(sys.argv) > 5 else "zamlf_n30_nx6"
if fam == "ca":
    cfg = CA_CFG
    x0, p = ca_batch(cfg, B)
else:
    cfg, kw = FAMILIES[fam]
    x0, p = synthetic_batch(cfg, B, **kw)
s = make_solver(cfg)
set_cfg_bounds(s, cfg)
for e in sys.argv[6:]:
    k, v = e.split("=")
    s.set_option(k, v)
d = [torch.from_numpy(a).cuda() for a in (x0, p)]
out = torch.empty_like(d[0]); st = torch.empty(B, dtype=torch.int32, device="cuda"); it = torch.empty_like(st)
kk = torch.empty(B, dtype=torch.float64, device="cuda")


def step():
    s.solve_device(B, d[0].data_ptr(), d[1].data_ptr(), out.data_ptr(), st.data_ptr(), it.data_ptr(), kk.data_ptr())


ts = {v: [] for v in vals}; pm = {v: [] for v in vals}; wm = {v: [] for v in vals}; res = {}
for rep in range(6):
    for v in vals:
        s.set_option(opt, v)
        s.set_profiling(False)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        for _ in range(15):
            t0 = time.perf_counter(); step(); ts[v].append(time.perf_counter() - t0)
        s.set_profiling(True)
        for _ in range(4):
            step(); pm[v].append(s.get_pipeline_profile()["ms"]); wm[v].append(s.get_resident_profile()["ms"])
        res[v] = (out.cpu().numpy().copy(), it.cpu().numpy().copy(), st.cpu().numpy().copy())
for v in vals:
    print(f"{opt}={v:4s} {fam} B={B}: median {np.median(ts[v])*1e3:.4f} ms/batch (min {np.min(ts[v])*1e3:.4f})  pipeline {np.median(pm[v]):.4f} ms  wg {np.median(wm[v]):.4f} ms"
          f"  conv {np.mean(res[v][2] == 1):.4f} iters {res[v][1].mean():.3f}/{res[v][1].max()}", flush=True)
a, b = res[vals[0]], res[vals[1]]
print("results bit-identical:", bool(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])),
      " max |dx| =", float(np.max(np.abs(a[0] - b[0]))), " iteration counts differ on", int(np.sum(a[1] != b[1])))
