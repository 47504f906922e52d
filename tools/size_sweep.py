"""ms per batch and instances per second over the batch size (device-resident buffers, default options): where the path changes (stragglers' kernel alone,
one or two instances per wavefront; pipeline + stragglers; chunks) and whether a size is slower than a larger one.
Usage (GPU box): python tools/size_sweep.py [family=zamlf_n30_nx6] [sizes ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np, torch
from helpers import FAMILIES, make_solver
from oracle.nlp_numpy import synthetic_batch
fam = sys.argv[1] if len(sys.argv) > 1 else "zamlf_n30_nx6"
sizes = [int(a) for a in sys.argv[2:]] or [64, 256, 512, 1024, 1025, 1536, 2048, 2049, 2304, 2560, 3072, 3500, 3584, 3800, 4096, 4097, 5120, 6144, 8192, 8193, 12288]
cfg, kw = FAMILIES[fam]
X0, P = synthetic_batch(cfg, max(sizes), **kw)
s = make_solver(cfg)
prev = None
for B in sizes:
    d = [torch.from_numpy(np.ascontiguousarray(a[:B])).cuda() for a in (X0, P)]
    out = torch.empty_like(d[0]); st = torch.empty(B, dtype=torch.int32, device="cuda"); it = torch.empty_like(st)
    def step(): s.solve_device(B, d[0].data_ptr(), d[1].data_ptr(), out.data_ptr(), st.data_ptr(), it.data_ptr())
    for _ in range(3): step()
    ts = []
    for _ in range(25):
        t0 = time.perf_counter(); step(); ts.append(time.perf_counter() - t0)
    ms = float(np.median(ts)) * 1e3
    note = "  <-- slower than the smaller size before it by more than the instances it adds" if prev and ms > prev[1] * 1.02 * max(1.0, B / prev[0]) else ""
    print(f"{fam} B={B:6d}: {ms:7.3f} ms  {B / ms / 1e3:6.2f} M instances/s  pipeline={int(s.get_pipeline_profile()['ran'])} stragglers' kernel={int(s.get_resident_profile()['ran'])} converged {float((st == 1).float().mean()):.4f}{note}", flush=True)
    prev = (B, ms)
