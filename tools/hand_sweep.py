"""Hand-over threshold of the hybrid solve (instances left in a tile when it leaves the pipeline) over batches the bench does not time:
other instances of the generator (start offsets), other batch sizes.  Usage (GPU box): python tools/hand_sweep.py [all|n30|n50]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
from helpers import FAMILIES, make_solver
from oracle.nlp_numpy import synthetic_batch

def run(fam, B, start, lives):
    cfg, kw = FAMILIES[fam]
    x0, p = synthetic_batch(cfg, B, start=start, **kw)
    s = make_solver(cfg)
    d = [torch.from_numpy(a).cuda() for a in (x0, p)]
    out = torch.empty_like(d[0]); st = torch.empty(B, dtype=torch.int32, device="cuda"); it = torch.empty_like(st)
    res = []
    for hl in lives:
        s.set_option("hybrid_live", str(hl))
        for _ in range(3):
            s.solve_device(B, d[0].data_ptr(), d[1].data_ptr(), out.data_ptr(), st.data_ptr(), it.data_ptr())
        ts = []
        for _ in range(15):
            t0 = time.perf_counter(); s.solve_device(B, d[0].data_ptr(), d[1].data_ptr(), out.data_ptr(), st.data_ptr(), it.data_ptr()); ts.append(time.perf_counter() - t0)
        res.append(f"{hl}: {np.median(ts) * 1e3:.3f}")
    print(f"{fam} B={B} start={start} iters {float(it.float().mean()):.2f}/{int(it.max())}  ms per batch by hybrid_live  " + "  ".join(res), flush=True)

which = sys.argv[1] if len(sys.argv) > 1 else "all"          # all | n30 | n50
if which in ("all", "n30"):
    for start in (0, 10000, 50000, 123456):
        run("zamlf_n30_nx6", 4096, start, (-1, 32, 40, 44, 48, 52))
    run("zamlf_n30_nx6", 8192, 0, (-1, 16, 20, 24, 28))
    run("zamlf_n30_nx6", 3000, 0, (-1, 40, 50, 60))
    run("zamlf_n30_nx5", 4096, 0, (-1, 32, 40, 48, 52))
if which in ("all", "n50"):
    for start in (0, 10000, 50000, 123456):
        run("usalf_n50_nx5", 4096, start, (-1, 20, 28, 36, 40, 44, 48))
    run("usalf_n50_nx5", 3000, 0, (-1, 28, 36, 44, 52))
    run("usalf_n50_nx5", 8192, 0, (-1, 10, 14, 20, 26))
