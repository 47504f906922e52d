"""Repetition test of the hybrid solve (k_pipeline hands tiles to k_solve_wg): the same batch solved again and again must give the same
bits -- a race in the hand-over (a tile retired while its rows are still in flight, a straggler workgroup reading rows the pipeline has
not written back) would show as a run-to-run difference.  Usage (GPU box): python tools/hyb_stress.py [repetitions] [B ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
from helpers import CA_CFG, FAMILIES, ca_batch, make_solver, set_cfg_bounds
from oracle.nlp_numpy import synthetic_batch

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 500
sizes = [int(a) for a in sys.argv[2:]] or [4096, 3000, 8192, 1024]
for fam in ("zamlf_n30_nx6", "usalf_n50_nx5", "ca"):
    for B in sizes:
        if fam == "ca":
            if B > 2048: continue
            cfg = CA_CFG; x0, p = ca_batch(cfg, B)
        else:
            cfg, kw = FAMILIES[fam]; x0, p = synthetic_batch(cfg, B, **kw)
        s = make_solver(cfg)
        if fam == "ca":
            set_cfg_bounds(s, cfg); s.set_option("rescue", "0")
        d = [torch.from_numpy(a).cuda() for a in (x0, p)]
        out = torch.empty_like(d[0]); st = torch.empty(B, dtype=torch.int32, device="cuda"); it = torch.empty_like(st); kk = torch.empty(B, dtype=torch.float64, device="cuda")
        ref = None; bad = 0; t0 = time.perf_counter()
        for r in range(reps):
            s.solve_device(B, d[0].data_ptr(), d[1].data_ptr(), out.data_ptr(), st.data_ptr(), it.data_ptr(), kk.data_ptr())
            cur = (out.clone(), st.clone(), it.clone())
            if ref is None: ref = cur
            elif not (torch.equal(cur[0], ref[0]) and torch.equal(cur[1], ref[1]) and torch.equal(cur[2], ref[2])): bad += 1
        dt = time.perf_counter() - t0
        pp, rp = s.get_pipeline_profile(), s.get_resident_profile()
        print(f"{fam:16s} B={B:5d}: {reps} solves, {bad} differ from the first; converged {float((ref[1] == 1).float().mean()):.4f}; "
              f"pipeline ran {pp['ran']}, k_solve_wg ran {rp['ran']}; {dt / reps * 1e3:.3f} ms per solve incl. the comparison", flush=True)
        assert bad == 0
print("OK")
