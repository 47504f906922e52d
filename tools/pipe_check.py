"""Compare the single-launch pipeline (k_pipeline) with one launch per kernel: bit-identity and time per batch.
Usage (GPU box): python tools/pipe_check.py [B] [family]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
from helpers import FAMILIES, make_solver
from oracle.nlp_numpy import synthetic_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
fam = sys.argv[2] if len(sys.argv) > 2 else "zamlf_n30_nx6"
cfg, kw = FAMILIES[fam]
x0, p = synthetic_batch(cfg, B, **kw)
for fixed in (0, 20):
    s = make_solver(cfg, fixed_iters=fixed) if fixed else make_solver(cfg)
    s.set_option("hybrid", "0")        # pipeline against one launch per kernel: the hybrid solve rounds differently
    res = {}
    for mode in ("0", "1"):
        s.set_option("pipeline", mode)
        r = s.solve(x0, p)
        pp = s.get_pipeline_profile()
        d = [torch.from_numpy(a).cuda() for a in (x0, p)]
        out = torch.empty_like(d[0]); st = torch.empty(B, dtype=torch.int32, device="cuda"); it = torch.empty_like(st); kk = torch.empty(B, dtype=torch.float64, device="cuda")
        def step():
            s.solve_device(B, d[0].data_ptr(), d[1].data_ptr(), out.data_ptr(), st.data_ptr(), it.data_ptr(), kk.data_ptr())
        for _ in range(3): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): step()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
        res[mode] = r
        print(f"fixed={fixed} pipeline={mode} ran={pp['ran']} {ms:.3f} ms/batch = {B / ms * 1e3 / 1e6:.3f} M steps/s  conv={np.mean(r.status == 1):.4f} "
              f"iters mean {r.iters.mean():.2f} max {r.iters.max()}  prof={ {k: (round(v, 3) if isinstance(v, float) else v) for k, v in pp.items()} }", flush=True)
    a, b = res["0"], res["1"]
    print("   bit-identical:", np.array_equal(a.x, b.x), np.array_equal(a.iters, b.iters), np.array_equal(a.status, b.status),
          " max |dx| =", float(np.abs(a.x - b.x).max()), flush=True)
