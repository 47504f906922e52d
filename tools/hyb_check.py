"""Hybrid solve (pipeline while a tile has many instances iterating, then k_solve_wg: one wavefront per instance with the MFMA Riccati)
against the pure pipeline: agreement and time per batch for a sweep of hand-over thresholds.
Usage (GPU box): python tools/hyb_check.py [B] [family|ca] [live thresholds ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
from helpers import CA_CFG, FAMILIES, ca_batch, make_solver, set_cfg_bounds
from oracle.nlp_numpy import synthetic_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
fam = sys.argv[2] if len(sys.argv) > 2 else "zamlf_n30_nx6"
lives = [int(a) for a in sys.argv[3:]] or [-1, 8, 16, 24, 32, 64]
if fam == "ca":
    cfg = CA_CFG
    x0, p = ca_batch(cfg, B)
else:
    cfg, kw = FAMILIES[fam]
    x0, p = synthetic_batch(cfg, B, **kw)
s = make_solver(cfg)
if fam == "ca":
    set_cfg_bounds(s, cfg)
    s.set_option("rescue", "0")
d = [torch.from_numpy(a).cuda() for a in (x0, p)]
out = torch.empty_like(d[0]); st = torch.empty(B, dtype=torch.int32, device="cuda"); it = torch.empty_like(st); kk = torch.empty(B, dtype=torch.float64, device="cuda")
def step():
    s.solve_device(B, d[0].data_ptr(), d[1].data_ptr(), out.data_ptr(), st.data_ptr(), it.data_ptr(), kk.data_ptr())
def run(tag):
    r = s.solve(x0, p)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
    s.set_profiling(True); step(); rp = s.get_resident_profile(); pp = s.get_pipeline_profile(); s.set_profiling(False)
    print(f"{fam} B={B} {tag:28s} {ms:7.3f} ms/batch = {B / ms * 1e3 / 1e6:6.3f} M steps/s  conv={np.mean(r.status == 1):.4f} iters mean {r.iters.mean():.2f} max {r.iters.max()}"
          f"  launch ms: pipeline {pp['ms']:.3f} ({pp['rounds']} rounds) + wg {rp['ms']:.3f} (slowest {rp['rounds']} rounds, {rp['workgroup_rounds']} wg-rounds, {rp['sweeps']} sweeps)", flush=True)
    return r
s.set_option("hybrid", "0")
base = run("pipeline")
for bx in (1, 2):
    for live in lives:
        s.set_option("hybrid", "1"); s.set_option("hybrid_bx", str(bx)); s.set_option("hybrid_live", str(live))
        r = run(f"hybrid bx={bx} live<={live}")
        both = (base.status == 1) & (r.status == 1)
        print(f"      vs pipeline: iters equal {np.mean(base.iters == r.iters):.4f} (|d| max {np.abs(base.iters - r.iters).max()})  status equal {np.mean(base.status == r.status):.4f}"
              f"  max |dx| (both converged) {np.abs(base.x[both] - r.x[both]).max():.2e}  kkt max {r.kkt[r.status == 1].max():.2e}", flush=True)
s.set_option("hybrid", "0")
