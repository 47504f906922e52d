#!/usr/bin/env python3
"""Throughput of the BASELINE.json parity configurations 2-4 (device-resident buffers), for the table in DESIGN.md.
usage (GPU box): python tools/configs_bench.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from helpers import CA_CFG, FAMILIES, ca_batch, make_solver, set_cfg_bounds  # noqa: E402
from oracle.nlp_numpy import synthetic_batch  # noqa: E402


def run(name, cfg, x0, p, reps=10):
    B = x0.shape[0]
    s = make_solver(cfg)
    set_cfg_bounds(s, cfg)
    dx0, dp = torch.tensor(x0, device="cuda"), torch.tensor(p, device="cuda")
    out = torch.empty_like(dx0)
    st = torch.empty(B, dtype=torch.int32, device="cuda")
    it = torch.empty(B, dtype=torch.int32, device="cuda")
    kk = torch.empty(B, dtype=torch.float64, device="cuda")

    def once():
        s.solve_device(B, dx0.data_ptr(), dp.data_ptr(), out.data_ptr(), st.data_ptr(), it.data_ptr(), kk.data_ptr())
        torch.cuda.synchronize()
    once()
    once()
    t = time.time()
    for _ in range(reps):
        once()
    dt = (time.time() - t) / reps
    ok = (st == 1)
    print("%-58s B=%5d  %7.2f ms/batch  %9.0f steps/s  converged %6.2f %%  iters mean %5.2f max %3d" %
          (name, B, dt * 1e3, B / dt, 100 * ok.float().mean().item(), it[ok].float().mean().item(), it.max().item()))


if __name__ == "__main__":
    cfg, kw = FAMILIES["zamlf_n30_nx6"]
    run("config 2: N=30 nx=6 lane following, synthetic refs", cfg, *synthetic_batch(cfg, 256, **kw))
    run("config 3: ZAM_Over-1_1 collision avoidance, N=30 (cold starts)", CA_CFG, *ca_batch(CA_CFG, 1024))
    cfg, kw = FAMILIES["usalf_n50_nx5"]
    run("config 4: USA_Lanker weights, N=50 long horizon", cfg, *synthetic_batch(cfg, 4096, **kw))
    cfg, kw = FAMILIES["zamlf_n30_nx6"]
    run("metric config: N=30 nx=6, batch 4096", cfg, *synthetic_batch(cfg, 4096, **kw))
    # config 5, one shard of the mixed scenario sweep on one GPU (bench.py --workload mixed is the N-GPU form)
    import subprocess
    print("config 5 (one 4096-row shard of the mixed sweep, three handles back to back):", flush=True)
    subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "mixed", "--steps", "10", "--warmup", "2"], check=False)
