#!/usr/bin/env python3
"""ad-hoc GPU bring-up script (not a pytest): parity vs oracle + first timings."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpc_amd
from oracle.nlp_numpy import NLPConfig, WEIGHTS_ZAM_LF, WEIGHTS_USA_LF, synthetic_batch
from oracle.binding import OracleSolver

def run(cfg, B, name, **kw):
    x0, p = synthetic_batch(cfg, B, **kw)
    s = mpc_amd.BatchedMPCSolver(cfg.N, cfg.nx, Q=cfg.Qdiag, R=cfg.R, obstacle_centers=cfg.obstacle_centers)
    t = time.time(); r = s.solve(x0, p); t1 = time.time() - t
    t = time.time(); r = s.solve(x0, p); t2 = time.time() - t
    ro = OracleSolver(cfg).solve_batch(x0, p, nthreads=8)
    print(f"[{name}] B={B} gpu status {np.unique(r.status, return_counts=True)} iters mean {r.iters.mean():.2f} max {r.iters.max()} "
          f"| oracle iters mean {ro['iters'].mean():.2f} | iters equal {(r.iters == ro['iters']).mean():.3f} "
          f"| max|dx| {np.abs(r.x - ro['x']).max():.3e} | kkt max {r.kkt.max():.2e} | host solve {t1*1e3:.1f} ms / {t2*1e3:.1f} ms", flush=True)
    return s

if __name__ == "__main__":
    run(NLPConfig(N=10, nx=5, **WEIGHTS_ZAM_LF), 64, "LF N10 nx5")
    run(NLPConfig(N=30, nx=5, **WEIGHTS_ZAM_LF), 256, "LF N30 nx5")
    run(NLPConfig(N=30, nx=6, **WEIGHTS_ZAM_LF), 256, "LF N30 nx6")
    run(NLPConfig(N=50, nx=5, **WEIGHTS_USA_LF), 256, "USA N50 nx5", v_range=(5.0, 9.0))
    # timing at batch 4096 with device-resident buffers
    import torch
    cfg = NLPConfig(N=30, nx=6, **WEIGHTS_ZAM_LF)
    B = 4096
    x0, p = synthetic_batch(cfg, B)
    for fixed in (0, 20):
        s = mpc_amd.BatchedMPCSolver(cfg.N, cfg.nx, Q=cfg.Qdiag, R=cfg.R, obstacle_centers=cfg.obstacle_centers, fixed_iters=fixed)
        dx0 = torch.from_numpy(x0).cuda(); dp = torch.from_numpy(p).cuda()
        dout = torch.empty_like(dx0); dst = torch.empty(B, dtype=torch.int32, device="cuda"); dit = torch.empty_like(dst); dk = torch.empty(B, dtype=torch.float64, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        for _ in range(3):
            s.solve_device(B, dx0.data_ptr(), dp.data_ptr(), dout.data_ptr(), dst.data_ptr(), dit.data_ptr(), dk.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        t = time.time(); K = 10
        for _ in range(K):
            s.solve_device(B, dx0.data_ptr(), dp.data_ptr(), dout.data_ptr(), dst.data_ptr(), dit.data_ptr(), dk.data_ptr(), stream=stream)
        torch.cuda.synchronize(); dt = (time.time() - t) / K
        st = dst.cpu().numpy(); it = dit.cpu().numpy()
        print(f"[bench fixed={fixed}] B={B}: {dt*1e3:.3f} ms/batch -> {B/dt:,.0f} MPC steps/s; status {np.unique(st, return_counts=True)} iters mean {it.mean():.2f} max {it.max()}", flush=True)
        s.set_profiling(True)
        s.solve_device(B, dx0.data_ptr(), dp.data_ptr(), dout.data_ptr(), dst.data_ptr(), dit.data_ptr(), dk.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        print("   profile:", s.get_profile(), flush=True)
