"""One timing sample of the FORCES-mode SQP step (row f3) with the library that is in place: bench.py's `other_paths.forces_sqp_step` on its own.
Usage (GPU box): python tools/forces_time.py [B] [tag]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
tag = sys.argv[2] if len(sys.argv) > 2 else ""
sf, fstep, d_fl, d_it, *_ = bench.forces_setup(B)
for _ in range(5):
    fstep()
torch.cuda.synchronize()
ts = []
for _ in range(12):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fstep()
    e1.record(); e1.synchronize()
    ts.append(e0.elapsed_time(e1) / 5)
fl, it = d_fl.cpu().numpy(), d_it.cpu().numpy()
out = fstep.keep[3].cpu().numpy()
print(f"{tag:6s} forces SQP step B={B}: median {np.median(ts):.4f} ms/batch = {B / np.median(ts) / 1e3:.3f} M solves/s  solved {np.mean(fl == 1):.4f}  qp iterations {it.mean():.2f}/{it.max()}"
      f"  checksum {float(np.abs(out).sum()):.10e}", flush=True)
