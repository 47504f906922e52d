"""Wall time of every one of n consecutive headline solves (device-resident buffers, what bench.py's timed loop does): mean, median, the outliers.
Usage (GPU box): python tools/step_times.py [n] [B]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from helpers import FAMILIES, make_solver
from oracle.nlp_numpy import synthetic_batch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
cfg, kw = FAMILIES["zamlf_n30_nx6"]
x0, p = synthetic_batch(cfg, B, **kw)
s = make_solver(cfg)
d = [torch.from_numpy(a).cuda() for a in (x0, p)]
out = torch.empty_like(d[0]); st = torch.empty(B, dtype=torch.int32, device="cuda"); it = torch.empty_like(st); kk = torch.empty(B, dtype=torch.float64, device="cuda")
def step(): s.solve_device(B, d[0].data_ptr(), d[1].data_ptr(), out.data_ptr(), st.data_ptr(), it.data_ptr(), kk.data_ptr())
for _ in range(3): step()
torch.cuda.synchronize()
ts = []
t0 = time.perf_counter()
for _ in range(n):
    a = time.perf_counter(); step(); ts.append((time.perf_counter() - a) * 1e3)
tot = (time.perf_counter() - t0) / n * 1e3
ts = np.array(ts)
print(f"B={B}: {n} steps: mean {ts.mean():.4f} (loop {tot:.4f}) median {np.median(ts):.4f} min {ts.min():.4f} max {ts.max():.4f} ms")
print("first 25:", " ".join(f"{v:.3f}" for v in ts[:25]))
print("above 1.03 x median:", [(i, round(float(v), 3)) for i, v in enumerate(ts) if v > 1.03 * np.median(ts)])
