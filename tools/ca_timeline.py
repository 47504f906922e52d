"""Where the time of BASELINE configuration 3 goes (ZAM_Over-1_1 collision avoidance, B = 1024, cold starts through the obstacle):
the main solve with the second chance off, the statuses / iteration counts it leaves, and the whole call with the second chance on.
Usage (GPU box): python tools/ca_timeline.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
from helpers import CA_CFG, ca_batch, make_solver, set_cfg_bounds

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cfg = CA_CFG
x0, p = ca_batch(cfg, B)
s = make_solver(cfg)
set_cfg_bounds(s, cfg)
d = [torch.from_numpy(a).cuda() for a in (x0, p)]
out = torch.empty_like(d[0]); st = torch.empty(B, dtype=torch.int32, device="cuda"); it = torch.empty_like(st)
kk = torch.empty(B, dtype=torch.float64, device="cuda")


def step():
    s.solve_device(B, d[0].data_ptr(), d[1].data_ptr(), out.data_ptr(), st.data_ptr(), it.data_ptr(), kk.data_ptr())
    torch.cuda.synchronize()


def timed(tag, reps=8):
    for _ in range(2):
        step()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); step(); ts.append(time.perf_counter() - t0)
    stc, itc = st.cpu().numpy(), it.cpu().numpy()
    bad = np.flatnonzero(stc != 1)
    print(f"{tag}: median {np.median(ts)*1e3:.3f} ms (min {np.min(ts)*1e3:.3f})  converged {np.mean(stc == 1):.4f}  iters mean {itc.mean():.2f} max {itc.max()}"
          f"  rescued {s.last_rescued()}", flush=True)
    return stc, itc, bad


for opt in sys.argv[2:]:
    k, v = opt.split("=")
    s.set_option(k, v)
s.set_option("rescue", "0")
stc, itc, bad = timed("second chance off")
print("  not converged:", len(bad), " statuses", sorted(set(stc[bad].tolist())), " their iteration counts", sorted(itc[bad].tolist()))
ok = itc[stc == 1]
print("  converged ones: iteration histogram (10-bins)", np.histogram(ok, bins=range(0, 111, 10))[0].tolist())
s.set_option("rescue", "1")
stc2, itc2, bad2 = timed("second chance on ")
print("  accumulated iterations of the rescued:", sorted(itc2[bad].tolist()))
