"""Split of an XCD's CUs between Riccati workers and stage workers of k_pipeline (option pipe_ric, per XCD): time per batch.
Usage (GPU box): python tools/ric_split.py [B] [pipe_ric values ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
from helpers import FAMILIES, make_solver
from oracle.nlp_numpy import synthetic_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
vals = [int(a) for a in sys.argv[2:]] or [0, 4, 5, 6, 7, 8, 10, 12]
cfg, kw = FAMILIES["zamlf_n30_nx6"]
x0, p = synthetic_batch(cfg, B, **kw)
s = make_solver(cfg)
d = [torch.from_numpy(a).cuda() for a in (x0, p)]
out = torch.empty_like(d[0]); st = torch.empty(B, dtype=torch.int32, device="cuda"); it = torch.empty_like(st); kk = torch.empty(B, dtype=torch.float64, device="cuda")
def step():
    s.solve_device(B, d[0].data_ptr(), d[1].data_ptr(), out.data_ptr(), st.data_ptr(), it.data_ptr(), kk.data_ptr())
for hyb in ("1", "0"):
    s.set_option("hybrid", hyb)
    for v in vals:
        s.set_option("pipe_ric", str(v))
        for _ in range(3): step()
        ms = []
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): step()
            torch.cuda.synchronize(); ms.append((time.perf_counter() - t0) / 20 * 1e3)
        s.set_profiling(True); step(); pp = s.get_pipeline_profile(); rp = s.get_resident_profile(); s.set_profiling(False)
        print(f"hybrid={hyb} pipe_ric={v:2d}: {min(ms):.3f} ms/batch (median {sorted(ms)[1]:.3f})  pipeline {pp['ms']:.3f} ms ({pp['rounds']} rounds, {pp.get('riccati_workers')} + {pp.get('stage_workers')} workers)  wg {rp['ms']:.3f} ms  conv {float((st == 1).float().mean()):.4f}", flush=True)
