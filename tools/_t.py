import sys
sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo")
import numpy as np, time, torch
import workloads as wl
fam = wl.FAMILIES["zamlf_n30_nx6"]
B=4096
x0, p = wl.batch(fam, B)
for fixed in (0, 20):
    s = wl.make_solver(fam, fixed_iters=fixed) if fixed else wl.make_solver(fam)
    d = [torch.from_numpy(a).cuda() for a in (x0, p)]
    out = torch.empty_like(d[0]); st = torch.empty(B, dtype=torch.int32, device="cuda"); it = torch.empty_like(st); kk = torch.empty(B, dtype=torch.float64, device="cuda")
    def step(): s.solve_device(B, d[0].data_ptr(), d[1].data_ptr(), out.data_ptr(), st.data_ptr(), it.data_ptr(), kk.data_ptr())
    for _ in range(3): step()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(30): step()
    torch.cuda.synchronize(); ms=(time.perf_counter()-t0)/30*1e3
    s.set_profiling(True); step(); pp=s.get_pipeline_profile(); s.set_profiling(False)
    print("fixed=%d: %.3f ms/batch = %.3f M steps/s, launch %.3f ms, rounds %d, conv %.4f" % (fixed, ms, B/ms/1e3, pp["ms"], pp["rounds"], (st==1).float().mean().item()))
s = wl.make_solver(fam, fixed_iters=20); s.solve(x0,p); s.set_option("pipe_timing","1"); s.solve(x0,p)
