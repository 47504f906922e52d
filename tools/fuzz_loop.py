"""The closed loop on the device (mpc_closed_loop_batch: every solve enqueued without host synchronisation) for random numbers of egos, steps,
horizons and state counts against the same loop with the host between the steps (option loop_async = 0): the same bits; without noise every step converged and the loop was not replayed.
Usage (GPU box): python tools/fuzz_loop.py [cases=40] [seed=1]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from helpers import pkg
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def loop_inputs(B, L, v, psi, seed):
    k = np.arange(L)
    path = np.stack([k * v * 0.1 * np.cos(psi), k * v * 0.1 * np.sin(psi)], axis=1)
    r = np.random.default_rng(seed)
    init = np.tile([0.0, 0.0, 0.0, v, psi], (B, 1))
    init[:, 1] += r.uniform(-0.5, 0.5, B)
    init[:, 3] *= r.uniform(0.9, 1.1, B)
    return init, np.tile(path, (B, 1, 1)), np.full((B, L), psi), np.full(B, v)


solvers = {}
bad = 0
t0 = time.time()
for c in range(n):
    N, nx = [(10, 5), (30, 5), (30, 6), (50, 5)][rng.integers(4)]
    B = int([1, 7, 64, 65, 255, 1024, 1025, 4096][rng.integers(8)]) if rng.random() < 0.4 else int(rng.integers(1, 3000))
    L = N + int(rng.integers(0, 9))                     # (the interface wants at least a horizon of steps, as the reference's iter_length)
    Lp = L + int(rng.integers(0, N))
    init, path, orient, vdes = loop_inputs(B, Lp, float(rng.uniform(8, 20)), float(rng.uniform(-0.3, 0.3)), int(rng.integers(1 << 30)))
    if (N, nx) not in solvers:
        s = pkg.BatchedMPCSolver(N, nx); s.set_bounds()
        solvers[N, nx] = s
    s = solvers[N, nx]
    mode, sigma = (int(rng.integers(0, 3)), 0.05) if rng.random() < 0.3 else (0, 0.0)
    s.set_option("loop_async", "1")
    ta, ca, sa = s.closed_loop(init, path, orient, vdes, L, noise_mode=mode, sigma=sigma, seed=c)
    rep = s.last_loop_replayed()
    s.set_option("loop_async", "0")
    ts, cs, ss = s.closed_loop(init, path, orient, vdes, L, noise_mode=mode, sigma=sigma, seed=c)
    same = np.array_equal(ta, ts) and np.array_equal(ca, cs) and np.array_equal(sa, ss)
    conv = float((sa == 1).mean())
    # (with noise on the measured state a step can start outside its bounds -- the steering angle past its limit -- and stop unconverged: reported in
    #  step_status, the loop is then replayed with the host in between; the bar under noise is that both forms agree)
    ok = same and (conv == 1.0 or mode != 0) and np.isfinite(ta).all() and (rep == (conv < 1.0))
    bad += not ok
    print(f"{'ok ' if ok else 'BAD'} N={N:2d} nx={nx} B={B:5d} steps={L:2d} path={Lp:3d} noise={mode} async loop replayed={rep} same bits={same} converged steps={conv:.4f}", flush=True)
print(f"problems: {bad} of {n} in {time.time() - t0:.0f} s")
