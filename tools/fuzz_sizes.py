"""Random batch sizes and row windows through the default path (start kernel, pipeline, stragglers -- whichever the size selects) against the path
with one launch per kernel: same statuses, rows to 1e-7 (collision avoidance: 1e-4, see below), every instance converged (or, with an iteration limit of 6 / 9, stopped at the same count on both paths); the default path twice -> the same bits.  Usage (GPU box): python tools/fuzz_sizes.py [cases=120] [seed=1]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from helpers import CA_CFG, FAMILIES, ca_batch, make_solver, set_cfg_bounds
from oracle.nlp_numpy import synthetic_batch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
POOL = 9000
pools = {}
for fam, (cfg, kw) in FAMILIES.items():
    pools[fam] = (cfg, synthetic_batch(cfg, 17000 if fam in ("zamlf_n30_nx6", "zamlf_n10_nx5") else POOL, **kw))      # (two families beyond one persistent launch's 16384 instances)
pools["ca"] = (CA_CFG, ca_batch(CA_CFG, 3000))
solvers = {}
LIMITS = (100, 6, 9)               # iteration limits: the default, and two at which part of a lane-following batch stops unconverged (status 0)
for fam, (cfg, _) in pools.items():
    for mi in LIMITS if fam != "ca" else (100,):
        a, b = make_solver(cfg, max_iter=mi), make_solver(cfg, max_iter=mi)
        if fam == "ca":
            set_cfg_bounds(a, cfg); set_cfg_bounds(b, cfg)
        b.set_option("pipeline", "0"); b.set_option("hybrid", "0")
        solvers[fam, mi] = (a, b)
bad = 0
t0 = time.time()
names = list(pools)
sizes = [1, 2, 7, 8, 9, 63, 64, 65, 127, 129, 255, 257, 511, 1023, 1024, 1025, 2047, 2049, 2305, 4095, 4097, 8191, 8193, 16383, 16385]
for c in range(n):
    fam = names[c % len(names)]
    cfg, (X0, P) = pools[fam]
    cap = X0.shape[0]
    B = int(sizes[rng.integers(len(sizes))]) if rng.random() < 0.4 else int(rng.integers(1, cap + 1))
    B = min(B, cap)
    o = int(rng.integers(0, cap - B + 1))
    x0, p = np.ascontiguousarray(X0[o:o + B]), np.ascontiguousarray(P[o:o + B])
    mi = 100 if (fam == "ca" or rng.random() < 0.7) else int(LIMITS[1 + rng.integers(2)])
    a, b = solvers[fam, mi]
    # options of the default path drawn per case (they pick other kernels, never other results beyond the last bits): instances per wavefront of the
    # stragglers' kernel, helping Riccati workers, the second chance inside / behind the launch, run-time bound look-up; collision avoidance: the
    # obstacle handed over per instance (variant 0 of the kernels), moved by up to 0.2 m per instance
    opts = {}
    if rng.random() < 0.35:
        k = ("hybrid_bx", "pipe_help", "rescue_wg", "bound_mask")[rng.integers(4)]
        opts[k] = {"hybrid_bx": ("1", "2"), "pipe_help": ("0", "1"), "rescue_wg": ("0", "2"), "bound_mask": ("0", "1")}[k][rng.integers(2)]
    obst = None
    if fam == "ca" and rng.random() < 0.4:
        obst = np.tile(cfg.obstacle_centers.ravel(), (B, 1)) + np.repeat(rng.uniform(-0.2, 0.2, (B, 1, 2)), 3, axis=1).reshape(B, 6)
    for k, v in opts.items():
        a.set_option(k, v)
    r1 = a.solve(x0, p, obst=obst); mode = (a.get_pipeline_profile()["ran"], a.get_resident_profile()["ran"]); resc = a.last_rescued()
    r2 = a.solve(x0, p, obst=obst)
    for k in opts:
        a.set_option(k, {"hybrid_bx": "0", "pipe_help": "-1", "rescue_wg": "1", "bound_mask": "1"}[k])
    rr = b.solve(x0, p, obst=obst)
    same = np.array_equal(r1.x, r2.x) and np.array_equal(r1.status, r2.status) and np.array_equal(r1.iters, r2.iters)
    dx = float(np.abs(r1.x - rr.x).max())
    st_ok = np.array_equal(r1.status, rr.status)
    conv = float((r1.status == 1).mean())
    dit = int(np.abs(r1.iters.astype(int) - rr.iters.astype(int)).max())
    # (collision avoidance: the two paths' Riccati sweeps round differently -- fp64 matrix pipe in k_solve_wg, scalar FMAs elsewhere --, and at a KKT error
    #  of 1e-8 the 1e10 ... 1e13 weights of active circle rows pin the rows to 1e-6 ... 1e-5 only, with different iteration counts on the way; the bar
    #  here is north_star's 1e-4 on the trajectories, the test suite certifies the KKT conditions of such rows against the numpy NLP)
    rowd = np.abs(r1.x - rr.x).max(axis=1)
    n6 = int((rowd > 1e-6).sum())
    # (... and nonconvex: an instance whose cold start runs straight at the obstacle may pass it on the other side on the other path -- both converged
    #  KKT points, metres apart; with the obstacle moved per instance a few such instances turn up: at most 0.5 % of a batch may differ that way)
    flips = int((rowd > 1e-2).sum())
    tol_ok = dx < 1e-7 or (fam == "ca" and float(rowd[rowd <= 1e-2].max(initial=0.0)) < 1e-4 and flips <= max(1, B // 200))
    it_ok = mi == 100 or np.array_equal(r1.iters, rr.iters)          # (a batch cut off by its iteration limit: the same counts on both paths)
    ok = same and st_ok and (conv == 1.0 or mi < 100) and tol_ok and it_ok
    if not ok:
        bad += 1
    print(f"{'ok ' if ok else 'BAD'} {fam:14s} max_iter={mi:3d} B={B:5d} off={o:5d} {'obst ' if obst is not None else ''}{' '.join(f'{k}={v}' for k, v in opts.items())} pipeline={int(mode[0])} wg={int(mode[1])} rescued={resc:3d} repeat-bits={same} status-eq={st_ok} conv={conv:.4f} |dx|={dx:.2e} ({n6} rows above 1e-6, {flips} on another local optimum) |dit|={dit}", flush=True)
print(f"problems: {bad} of {n} in {time.time() - t0:.0f} s")
