"""The resident solve (k_resident) against the streaming paths and the oracle: agreement and time per batch.
Usage (GPU box): python tools/res_check.py [B] [family] [--ca]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
from helpers import CA_CFG, FAMILIES, ca_batch, make_solver, set_cfg_bounds
from oracle.nlp_numpy import synthetic_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
fam = sys.argv[2] if len(sys.argv) > 2 else "zamlf_n30_nx6"
if fam == "ca":
    cfg = CA_CFG
    x0, p = ca_batch(cfg, B)
else:
    cfg, kw = FAMILIES[fam]
    x0, p = synthetic_batch(cfg, B, **kw)
for fixed in (0, 20):
    s = make_solver(cfg, fixed_iters=fixed) if fixed else make_solver(cfg)
    if fam == "ca":
        set_cfg_bounds(s, cfg)
        s.set_option("rescue", "0")
    res = {}
    for mode in ("0", "2"):
        s.set_option("resident", mode)
        r = s.solve(x0, p)
        rp = s.get_resident_profile()
        d = [torch.from_numpy(a).cuda() for a in (x0, p)]
        out = torch.empty_like(d[0]); st = torch.empty(B, dtype=torch.int32, device="cuda"); it = torch.empty_like(st); kk = torch.empty(B, dtype=torch.float64, device="cuda")
        def step():
            s.solve_device(B, d[0].data_ptr(), d[1].data_ptr(), out.data_ptr(), st.data_ptr(), it.data_ptr(), kk.data_ptr())
        for _ in range(3): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): step()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
        s.set_profiling(True); step(); rp2 = s.get_resident_profile(); pp2 = s.get_pipeline_profile(); s.set_profiling(False)
        res[mode] = r
        print(f"{fam} B={B} fixed={fixed} resident={mode} ran={rp['ran']} {ms:.3f} ms/batch = {B / ms * 1e3 / 1e6:.3f} M steps/s  conv={np.mean(r.status == 1):.4f} "
              f"iters mean {r.iters.mean():.2f} max {r.iters.max()}  launch_ms res={rp2['ms']:.3f} pipe={pp2['ms']:.3f} rounds={rp['rounds']} wg_rounds={rp['workgroup_rounds']} sweeps={rp['sweeps']}", flush=True)
    a, b = res["0"], res["2"]
    both = (a.status == 1) & (b.status == 1)
    print("   streaming vs resident: iters equal", float(np.mean(a.iters == b.iters)), " status equal", float(np.mean(a.status == b.status)),
          " max |dx| (both converged) =", float(np.abs(a.x[both] - b.x[both]).max()), " kkt max", float(b.kkt[b.status == 1].max()), flush=True)
    if fam == "ca":
        break
