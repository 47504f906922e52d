"""ms per batch over the horizon N (ZAM_Over-1_1 lane-following weights, nx = 5, default options, device-resident buffers): which path serves a horizon and
whether a horizon is slower than a longer one.  Usage (GPU box): python tools/horizon_sweep.py [B=4096] [horizons ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from helpers import make_solver
from oracle.nlp_numpy import NLPConfig, WEIGHTS_ZAM_LF, synthetic_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
Ns = [int(a) for a in sys.argv[2:]] or [3, 5, 7, 8, 10, 11, 12, 15, 16, 20, 22, 23, 24, 28, 30, 31, 32, 36, 40, 47, 48, 50, 56, 63, 64, 80, 100, 127]
prev = None
for N in Ns:
    cfg = NLPConfig(N=N, nx=5, **WEIGHTS_ZAM_LF)
    x0, p = synthetic_batch(cfg, B)
    s = make_solver(cfg)
    d = [torch.from_numpy(a).cuda() for a in (x0, p)]
    out = torch.empty_like(d[0]); st = torch.empty(B, dtype=torch.int32, device="cuda"); it = torch.empty_like(st)
    def step(): s.solve_device(B, d[0].data_ptr(), d[1].data_ptr(), out.data_ptr(), st.data_ptr(), it.data_ptr())
    for _ in range(3): step()
    ts = []
    for _ in range(15):
        t0 = time.perf_counter(); step(); ts.append(time.perf_counter() - t0)
    ms = float(np.median(ts)) * 1e3
    note = "  <-- slower than a longer horizon would suggest" if prev and ms < prev[1] * 0.97 else ""
    print(f"N={N:3d} B={B}: {ms:7.3f} ms  {ms / (N + 1) * 1e3:6.1f} us per stage  pipeline={int(s.get_pipeline_profile()['ran'])} stragglers' kernel={int(s.get_resident_profile()['ran'])} iterations {float(it.float().mean()):.2f}/{int(it.max())} converged {float((st == 1).float().mean()):.4f}{note}", flush=True)
    prev = (N, ms)
    del s
