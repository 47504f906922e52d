"""Per-batch times of the oracle (C port, OpenMP over instances) on the host cores at several thread counts: what bench.py's cpu_baseline
leg rests on.  Usage: python tools/cpu_baseline_probe.py [nbatches]"""
import os, sys, time
HOST = len(os.sched_getaffinity(0))
os.environ.setdefault("OMP_PROC_BIND", "close"); os.environ.setdefault("OMP_PLACES", "cores")
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from oracle.binding import OracleSolver
from oracle.nlp_numpy import NLPConfig, synthetic_batch
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 12
cfg = NLPConfig(N=30, nx=6)
x0, p = synthetic_batch(cfg, 4096)
osol = OracleSolver(cfg)
print("host cpus", HOST, "OMP_PROC_BIND", os.environ.get("OMP_PROC_BIND"), "OMP_PLACES", os.environ.get("OMP_PLACES"), "with torch" if "--torch" in sys.argv else "")
if "--torch" in sys.argv:
    import torch; torch.zeros(1).cuda()
for cores in [c for c in (16, 32, 64, 128, 256) if c <= HOST]:
    ts = []
    out = osol.solve_batch(x0, p, nthreads=cores)
    for i in range(nb):
        t0 = time.perf_counter(); osol.solve_batch(x0, p, nthreads=cores, out=out); ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e3
    print(f"{cores:4d} threads: batches (ms) {np.round(ts, 1).tolist()}  median of the last {nb - 2}: {np.median(ts[2:]):.1f} ms = {4096 / np.median(ts[2:]) * 1e3:.0f} steps/s")
