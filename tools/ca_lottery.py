"""How much of configuration 3's time is the family's sensitivity to the last bits: the 1024 collision-avoidance cold starts solved with the
reference path perturbed by a relative 1e-13 (far below any tolerance), a few times -- slowest instance and batch time of each.
Usage (GPU box): python tools/ca_lottery.py [draws]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
from helpers import CA_CFG, ca_batch, make_solver, set_cfg_bounds

B = 1024
x0, p = ca_batch(CA_CFG, B)
s = make_solver(CA_CFG)
set_cfg_bounds(s, CA_CFG)
rng = np.random.default_rng(1)
out = torch.empty(B, x0.shape[1], dtype=torch.float64, device="cuda"); st = torch.empty(B, dtype=torch.int32, device="cuda"); it = torch.empty_like(st)
kk = torch.empty(B, dtype=torch.float64, device="cuda")
for d in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    pq = p * (1.0 + (0.0 if d == 0 else 1e-13) * rng.standard_normal(p.shape))          # (the reference path: the cold start itself is zeros and copies)
    dx = torch.from_numpy(x0).cuda()
    dp = torch.from_numpy(pq).cuda()
    def step():
        s.solve_device(B, dx.data_ptr(), dp.data_ptr(), out.data_ptr(), st.data_ptr(), it.data_ptr(), kk.data_ptr())
        torch.cuda.synchronize()
    step(); step()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); step(); ts.append(time.perf_counter() - t0)
    itc = it.cpu().numpy()
    print(f"draw {d}: {np.median(ts)*1e3:.3f} ms  converged {float((st == 1).float().mean()):.4f}  iterations mean {itc.mean():.2f}  top five {sorted(itc.tolist())[-5:]}  rescued {s.last_rescued()}", flush=True)
