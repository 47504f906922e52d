import sys, os
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
os.environ["MPCGPU_STAGE_TIMING"]="1"
from helpers import *
cfg,kw=FAMILIES["zamlf_n30_nx6"]
for B in (4096, 256):
    x0,p=synthetic_batch(cfg,B,**kw)
    s=make_solver(cfg); r=s.solve(x0,p); r=s.solve(x0,p)
    print("B",B,"iters",r.iters.mean())
