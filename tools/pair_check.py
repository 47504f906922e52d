"""Two threads per (instance, stage) in the stage phases (option pair: stage_pair of csrc/mpcgpu.hip, ROLE_A / ROLE_B of
csrc/mpc_stage_math.h) against one thread per stage: agreement with each other and with the oracle, time per batch, launch durations.
Usage (GPU box): python tools/pair_check.py [B] [family|ca] [n oracle rows]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
from helpers import CA_CFG, FAMILIES, ca_batch, make_solver, set_cfg_bounds
from oracle.binding import OracleSolver
from oracle.nlp_numpy import synthetic_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
fam = sys.argv[2] if len(sys.argv) > 2 else "zamlf_n30_nx6"
n_or = int(sys.argv[3]) if len(sys.argv) > 3 else 256
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
    print(f"{fam} B={B} {tag:18s} {ms:7.3f} ms/batch = {B / ms * 1e3 / 1e6:6.3f} M steps/s  conv={np.mean(r.status == 1):.4f} iters mean {r.iters.mean():.2f} max {r.iters.max()}"
          f"  launch ms: pipeline {pp['ms']:.3f} ({pp['rounds']} rounds) + wg {rp['ms']:.3f} (slowest {rp['rounds']} rounds, {rp['workgroup_rounds']} wg-rounds)", flush=True)
    return r


ro = OracleSolver(cfg).solve_batch(x0[:n_or], p[:n_or], nthreads=16)
res = {}
for pair in ("0", "1"):
    s.set_option("pair", pair)
    r = res[pair] = run(f"pair={pair}")
    both = (r.status[:n_or] == 1) & (ro["status"] == 1)
    print(f"      vs oracle ({n_or} rows): status equal {np.mean(r.status[:n_or] == ro['status']):.4f}  iters equal {np.mean(r.iters[:n_or] == ro['iters']):.4f}"
          f"  max |dx| (both converged) {np.abs(r.x[:n_or][both] - ro['x'][both]).max():.2e}  kkt max {r.kkt[r.status == 1].max():.2e}", flush=True)
a, b = res["0"], res["1"]
both = (a.status == 1) & (b.status == 1)
print(f"pair vs single: status equal {np.mean(a.status == b.status):.4f}  iters equal {np.mean(a.iters == b.iters):.4f}  max |dx| {np.abs(a.x[both] - b.x[both]).max():.2e}")
# repeatability of the pair path
s.set_option("pair", "1")
r2 = s.solve(x0, p)
print("pair path run twice: bit-identical", bool(np.array_equal(r2.x, b.x) and np.array_equal(r2.iters, b.iters)))
