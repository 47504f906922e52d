"""FORCES-mode SQP step (mpc_forces_solve_batch): an instance's result must not depend on the batch it is solved in -- random windows of a pool of
3000 instances (with and without the obstacle, both Hessian modes) against the pool solved at once: the same bits, flags and iteration counts.
Usage (GPU box): python tools/fuzz_forces.py [cases=60] [seed=1]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from helpers import pkg
from oracle import forces_model_numpy as FM
import test_forces_qp as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
POOL = 3000
w = FM.WEIGHTS_MODEL_C
s = pkg.BatchedMPCSolver(T.N, 5, Q=w["Q"], R=w["R"], P=w["P"])
zbar, params, xinit = T.family(POOL, seed=11)
ref = {m: s.forces_solve(zbar, xinit, params, T.LB, T.UB, T.HL, T.HU, hessian_mode=m) for m in (0, 1)}
print("pool:", {m: f"converged {float((ref[m][1] == 1).mean()):.3f}, iterations {float(ref[m][2].mean()):.2f}" for m in ref})
bad = 0
t0 = time.time()
for c in range(n):
    m = int(rng.integers(2))
    B = int([1, 2, 15, 16, 17, 31, 33, 255, 257, 1024, 2047][rng.integers(11)]) if rng.random() < 0.5 else int(rng.integers(1, POOL + 1))
    o = int(rng.integers(0, POOL - B + 1))
    x, flag, it, res = s.forces_solve(zbar[o:o + B], xinit[o:o + B], params[o:o + B], T.LB, T.UB, T.HL, T.HU, hessian_mode=m)
    ok = np.array_equal(x, ref[m][0][o:o + B]) and np.array_equal(flag, ref[m][1][o:o + B]) and np.array_equal(it, ref[m][2][o:o + B])
    bad += not ok
    print(f"{'ok ' if ok else 'BAD'} hessian_mode={m} B={B:5d} off={o:5d} |dx|={float(np.abs(x - ref[m][0][o:o + B]).max()):.2e}", flush=True)
print(f"problems: {bad} of {n} in {time.time() - t0:.0f} s")
