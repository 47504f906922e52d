"""One timing sample of the headline solve with the library that is in place (A/B comparisons on one box: tools/ab_bench.sh copies
variants of libmpcgpu.so over the built one and calls this in fresh processes, alternating).
Usage (GPU box): [MPC_AB_OPTS="option=value ..."] python tools/ab_time.py [B] [family] [tag]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
from helpers import CA_CFG, FAMILIES, ca_batch, make_solver, set_cfg_bounds
from oracle.nlp_numpy import synthetic_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
fam = sys.argv[2] if len(sys.argv) > 2 else "zamlf_n30_nx6"
tag = sys.argv[3] if len(sys.argv) > 3 else ""
if fam == "ca":
    cfg = CA_CFG
    x0, p = ca_batch(cfg, B)
else:
    cfg, kw = FAMILIES[fam]
    x0, p = synthetic_batch(cfg, B, **kw)
s = make_solver(cfg)
if fam == "ca":
    set_cfg_bounds(s, cfg)
for e in os.environ.get("MPC_AB_OPTS", "").split():          # option=value pairs, e.g. MPC_AB_OPTS="hybrid_live=40"
    k_, v_ = e.split("=")
    s.set_option(k_, v_)
d = [torch.from_numpy(a).cuda() for a in (x0, p)]
out = torch.empty_like(d[0]); st = torch.empty(B, dtype=torch.int32, device="cuda"); it = torch.empty_like(st); kk = torch.empty(B, dtype=torch.float64, device="cuda")
def step():
    s.solve_device(B, d[0].data_ptr(), d[1].data_ptr(), out.data_ptr(), st.data_ptr(), it.data_ptr(), kk.data_ptr())
for _ in range(5): step()
torch.cuda.synchronize()
ts = []
for _ in range(40):
    t0 = time.perf_counter(); step(); ts.append(time.perf_counter() - t0)
s.set_profiling(True)
pm, wm = [], []
for _ in range(10):
    step(); pm.append(s.get_pipeline_profile()["ms"]); wm.append(s.get_resident_profile()["ms"])
print(f"{tag:8s} {fam} B={B}: median {np.median(ts)*1e3:.4f} ms/batch (min {np.min(ts)*1e3:.4f})  pipeline {np.median(pm):.4f} ms  wg {np.median(wm):.4f} ms  conv {float((st==1).float().mean()):.4f} iters {float(it.float().mean()):.3f}/{int(it.max())}", flush=True)
