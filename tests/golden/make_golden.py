#!/usr/bin/env python3
"""
Generates the committed golden fixtures under tests/golden/.  Run in the BUILD container only
(`python tests/golden/make_golden.py`); the GPU box never runs this and never sees /root/reference.

Three fixture files:

  nlp_optima.npz        NLP optima of the reference formulation computed by scipy's SLSQP
                        on the numpy restatement (oracle/nlp_numpy.py) -- an independent solver, NOT our IPM --
                        cross-checked by scipy's trust-constr.  CasADi/IPOPT cannot be imported here
                        ("parity unpinned"), so these are the pinned optima.
  plant_step_kat.npz    rows (x_k, u_k, x_{k+1}) copied from the reference's recorded closed-loop runs
                        test/2D_plots_*/planned states.txt + control inputs.txt.  They pin the plant step
                        bit-exactly: forward Euler for casadi_* runs (optimizer.py:649-650), one RK4 step for
                        forcespro_* runs (optimizer.py:97-98,356).
  forces_model_kat.npz  random (z, p) -> (f, grad f, c, jac c, h, jac h) evaluated by the reference's own
                        CasADi-generated C (test/FORCESNLPsolver/FORCESNLPsolver_model.c via
                        FORCESNLPsolver_casadi2forces, compiled by oracle/Makefile into oracle/_ref).

Fixtures are data (inputs + expected outputs); no reference source text is stored.
"""
import os
import sys
import time

import numpy as np
from scipy.optimize import Bounds, NonlinearConstraint, minimize

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.nlp_numpy import (BicycleNLP, NLPConfig, OBSTACLE_ZAM, WEIGHTS_USA_LF, WEIGHTS_ZAM_CA,  # noqa: E402
                              WEIGHTS_ZAM_LF, synthetic_instance)

OUT = os.path.dirname(os.path.abspath(__file__))
REF_TEST = "/root/reference/test"


# ----------------------------------------------------------------------------------------------------------
def scipy_slsqp(nlp, x0, p):
    lbg, ubg, lbx, ubx = nlp.bounds()
    eq = np.where(lbg == ubg)[0]
    iq = np.where(lbg != ubg)[0]
    lo = iq[np.isfinite(lbg[iq])]
    hi = iq[np.isfinite(ubg[iq])]
    # the 9 obstacle rows are 3 distinct rows repeated (optimizer.py:395-403): pass each once to SLSQP,
    # duplicated rows make its QP subproblem rank deficient without changing the feasible set
    first_obst = 1 + nlp.nx * (nlp.N + 1)
    lo = np.array([i for i in lo if i < first_obst or (i - first_obst) % 3 == 0], dtype=int)
    # |y| >= 0 is implied
    lo = lo[lo != 0]
    cons = [dict(type="eq", fun=lambda w: nlp.g(w, p)[eq] - lbg[eq], jac=lambda w: nlp.jac(w, p)[eq]),
            dict(type="ineq", fun=lambda w: nlp.g(w, p)[lo] - lbg[lo], jac=lambda w: nlp.jac(w, p)[lo]),
            dict(type="ineq", fun=lambda w: ubg[hi] - nlp.g(w, p)[hi], jac=lambda w: -nlp.jac(w, p)[hi])]
    bnds = [(None if not np.isfinite(l) else l, None if not np.isfinite(u) else u) for l, u in zip(lbx, ubx)]
    r = minimize(lambda w: nlp.f(w, p), x0, jac=lambda w: nlp.grad(w, p), bounds=bnds, constraints=cons,
                 method="SLSQP", options=dict(maxiter=400, ftol=1e-13))
    return r


def scipy_trust_constr(nlp, x0, p):
    lbg, ubg, lbx, ubx = nlp.bounds()
    lbg = lbg.copy()
    lbg[0] = -np.inf
    first_obst = 1 + nlp.nx * (nlp.N + 1)
    keep = np.array([i for i in range(nlp.n_g) if i < first_obst or (i - first_obst) % 3 == 0])

    def hess(w, v):
        lam = np.zeros(nlp.n_g)
        lam[keep] = v
        return nlp.hess_lag(w, p, 0.0, lam)
    con = NonlinearConstraint(lambda w: nlp.g(w, p)[keep], lbg[keep], ubg[keep], jac=lambda w: nlp.jac(w, p)[keep], hess=hess)
    r = minimize(lambda w: nlp.f(w, p), x0, jac=lambda w: nlp.grad(w, p),
                 hess=lambda w: nlp.hess_lag(w, p, 1.0, np.zeros(nlp.n_g)), bounds=Bounds(lbx, ubx),
                 constraints=[con], method="trust-constr",
                 options=dict(maxiter=3000, gtol=1e-10, xtol=1e-12, barrier_tol=1e-10, initial_barrier_parameter=0.1))
    return r


def first_step_instance(cfg, x_init):
    """optimizer.py:577-583: at MPC step 0 every reference column is the initial state."""
    x_init = np.asarray(x_init, float)
    if cfg.nx == 6:
        x_init = np.append(x_init, 0.0)
    p = np.concatenate([np.zeros(2 * cfg.N), np.tile(x_init, cfg.N + 1)])
    return p.copy(), p


def ca_instance(cfg, b):
    """ZAM_Over-1_1 collision avoidance: straight reference through the obstacle, perturbed ego start."""
    rng = np.random.default_rng(20240929 + b)
    psi = 0.03495
    x_init = np.array([29.9948 + rng.uniform(-2, 2), -1.1501 + rng.uniform(-0.4, 0.4), 0.0, 20.0 * rng.uniform(0.9, 1.0), psi])
    v_des = 20.0
    Xr = np.zeros((cfg.N + 1, cfg.nx))
    Xr[0, :5] = x_init
    for k in range(1, cfg.N + 1):
        Xr[k, :5] = [29.9948 + k * v_des * cfg.dt * np.cos(psi), -1.1501 + k * v_des * cfg.dt * np.sin(psi), 0.0, v_des, psi]
    p = np.concatenate([np.zeros(2 * cfg.N), Xr.ravel()])
    x0 = np.concatenate([np.zeros(2 * cfg.N), np.tile(Xr[0], cfg.N + 1)])
    return x0, p


# ---- the optima grid: {ZAM-LF, ZAM-CA, USA-LF weights} x N in {10, 30, 50} x 8 instances, the nx = 6
# headline family, and the first-step (tiled reference) cases.  Every optimum comes from COLD starts of scipy solvers on the numpy
# restatement; nothing is seeded by the oracle or the kernels.  The nonconvex collision-avoidance family is solved from several
# cold starts (the caller's straight-through guess, swerves to either side, perturbed guesses) and EVERY distinct local optimum
# found is kept (`w_alt`): the tests accept whichever basin the solver under test lands in and report it.
GRID_COUNT = 8
FIRST_STATES = [[29.9948, -1.1501, 0.0, 20.0, 0.03495],       # ZAM_Over-1_1.xml:3260-3282
                [0.0, 0.0, 0.0, 6.8062, -0.4268],              # USA_Lanker-2_18_T-1.xml:113282-113317
                [10.0, 3.0, 0.0, 12.0, 1.2]]
MAX_ALT = 4


def grid_families():
    fam = {}
    for N in (10, 30, 50):
        fam[f"zamlf_n{N}_nx5"] = (NLPConfig(N=N, nx=5, **WEIGHTS_ZAM_LF), "syn", GRID_COUNT)
        fam[f"usalf_n{N}_nx5"] = (NLPConfig(N=N, nx=5, **WEIGHTS_USA_LF), "syn_usa", GRID_COUNT)
        fam[f"zamca_n{N}_nx5"] = (NLPConfig(N=N, nx=5, obstacle=OBSTACLE_ZAM, **WEIGHTS_ZAM_CA), "ca", GRID_COUNT)
        fam[f"first_n{N}_nx5"] = (NLPConfig(N=N, nx=5, **WEIGHTS_ZAM_LF), "first", 3)
    fam["zamlf_n30_nx6"] = (NLPConfig(N=30, nx=6, **WEIGHTS_ZAM_LF), "syn", GRID_COUNT)
    return fam


def ca_cold_starts(cfg, x0, p, b):
    """cold starts for the nonconvex family: none of them comes from a solver"""
    N, nx = cfg.N, cfg.nx
    starts = [("x0", x0)]
    Xr = p[2 * N:].reshape(N + 1, nx)
    ox, oy = cfg.obstacle[0], cfg.obstacle[1]
    for side, label in ((+1.0, "swerve_left"), (-1.0, "swerve_right")):
        X = Xr.copy()
        X[0] = Xr[0]
        bump = side * 4.5 * np.exp(-0.5 * ((X[1:, 0] - ox) / 8.0) ** 2)
        X[1:, 1] = oy + bump * (np.abs(X[1:, 0] - ox) < 30.0) + Xr[1:, 1] * (np.abs(X[1:, 0] - ox) >= 30.0)
        starts.append((label, np.concatenate([np.zeros(2 * N), X.ravel()])))
    rng = np.random.default_rng(1000 + b)
    for q in range(2):
        X = np.tile(Xr[0], (N + 1, 1))
        X[1:, 1] += rng.normal(0.0, 1.5)
        starts.append((f"shifted{q}", np.concatenate([np.zeros(2 * N), X.ravel()])))
    return starts


def generic_cold_starts(cfg, nlp, x0, p):
    """further cold starts for the (locally unique) lane-following families, used when SLSQP does not get through from the
    caller's guess: a zero-input rollout from the initial state, and the reference itself as the state guess"""
    N, nx = cfg.N, cfg.nx
    U, X = nlp.split(x0.copy())
    X = X.copy()
    X[0] = p[2 * N:2 * N + nx]
    for k in range(N):
        X[k + 1] = nlp.plant_step(X[k], np.zeros(2))
    Xr = p[2 * N:].reshape(N + 1, nx)
    return [("rollout", np.concatenate([np.zeros(2 * N), X.ravel()])), ("reference", np.concatenate([np.zeros(2 * N), Xr.ravel()]))]


def feasible(nlp, w, p, tol=1e-7):
    lbg, ubg, lbx, ubx = nlp.bounds()
    g = nlp.g(w, p)
    return bool(np.all(g >= lbg - tol) and np.all(g <= ubg + tol) and np.all(w >= lbx - tol) and np.all(w <= ubx + tol))


def solve_job(job):
    name, b = job
    cfg, kind, _ = grid_families()[name]
    nlp = BicycleNLP(cfg)
    if kind == "syn":
        x0, p = synthetic_instance(cfg, b)
    elif kind == "syn_usa":
        x0, p = synthetic_instance(cfg, b, v_range=(5.0, 9.0))
    elif kind == "ca":
        x0, p = ca_instance(cfg, b)
    else:
        x0, p = first_step_instance(cfg, FIRST_STATES[b])
    t = time.time()
    cache = os.path.join(os.environ.get("GOLDEN_CACHE", "/tmp/golden_cache"), f"{name}_{b}.npz")
    if os.path.exists(cache):
        c = np.load(cache, allow_pickle=True)
        return name, b, c["x0"], c["p"], [tuple(q) for q in c["checked"]]
    starts = ca_cold_starts(cfg, x0, p, b) if kind == "ca" else [("x0", x0)] + generic_cold_starts(cfg, nlp, x0, p)
    found = []                                           # (f, w, label)
    for label, s0 in starts:
        r = scipy_slsqp(nlp, s0, p)
        # SLSQP with ftol 1e-13 often ends with status 8 ("positive directional derivative") AT the optimum: accept a feasible end
        # point whose restart does not move (the independent trust-constr run below then checks that it is a KKT point)
        if r.status not in (0, 8) or not feasible(nlp, r.x, p):
            continue
        r2 = scipy_slsqp(nlp, r.x, p)
        if r2.status not in (0, 8) or not feasible(nlp, r2.x, p) or np.abs(r2.x - r.x).max() > 1e-5:
            continue
        r = r2
        if any(np.abs(r.x - w).max() < 1e-4 for _, w, _ in found):
            continue
        found.append((float(r.fun), r.x, label))
        if kind != "ca":
            break                                        # locally unique: the first cold start that gets through is the optimum
    assert found, (name, b, "no cold start converged")
    found.sort(key=lambda q: q[0])
    # second, independent solver (interior-point trust-region) on every optimum kept: started from a perturbation of the SLSQP
    # point it acts as an independent KKT check of that point; from the cold start when that works (recorded in `start`)
    checked = []
    for f, w, label in found[:MAX_ALT]:
        tcstart = "x0"
        rt = scipy_trust_constr(nlp, x0, p) if kind != "ca" else None
        dtc = float(np.abs(rt.x - w).max()) if rt is not None else np.inf
        if dtc > 1e-5:
            pert = np.random.default_rng(b).normal(0.0, 1e-3, w.size)
            rt = scipy_trust_constr(nlp, w + pert, p)
            dtc = float(np.abs(rt.x - w).max())
            tcstart = "slsqp+1e-3"
        checked.append((f, w, label + "|" + tcstart, dtc))
    os.makedirs(os.path.dirname(cache), exist_ok=True)
    np.savez(cache, x0=x0, p=p, checked=np.array(checked, dtype=object))
    print(f"{name}[{b}] {len(found)} optimum/optima: " + ", ".join(f"f={c[0]:.6f} ({c[2]}, |dw|={c[3]:.1e})" for c in checked) + f"  {time.time() - t:.0f}s", flush=True)
    return name, b, x0, p, checked


def make_optima():
    import multiprocessing as mp
    fam = grid_families()
    jobs = [(name, b) for name, (_, _, count) in fam.items() for b in range(count)]
    jobs.sort(key=lambda j: -fam[j[0]][0].N)             # long horizons first
    with mp.Pool(int(os.environ.get("GOLDEN_PROCS", "6"))) as pool:
        results = pool.map(solve_job, jobs, chunksize=1)
    out = {}
    for name, (cfg, kind, count) in fam.items():
        rs = sorted([r for r in results if r[0] == name], key=lambda r: r[1])
        nw = cfg.n_w
        out[f"{name}__x0"] = np.array([r[2] for r in rs])
        out[f"{name}__p"] = np.array([r[3] for r in rs])
        out[f"{name}__w"] = np.array([r[4][0][1] for r in rs])                      # the lowest optimum found
        out[f"{name}__f"] = np.array([r[4][0][0] for r in rs])
        out[f"{name}__dtc"] = np.array([r[4][0][3] for r in rs])
        out[f"{name}__start"] = np.array([r[4][0][2] for r in rs])
        alt = np.full((count, MAX_ALT, nw), np.nan)
        altf = np.full((count, MAX_ALT), np.nan)
        altd = np.full((count, MAX_ALT), np.nan)
        for i, r in enumerate(rs):
            for k, c in enumerate(r[4]):
                alt[i, k], altf[i, k], altd[i, k] = c[1], c[0], c[3]
        out[f"{name}__w_alt"] = alt                                                  # every distinct local optimum found (row 0 = w)
        out[f"{name}__f_alt"] = altf
        out[f"{name}__dtc_alt"] = altd
        out[f"{name}__cfg"] = np.array([cfg.N, cfg.nx, cfg.dt, *cfg.Q, *cfg.R, *cfg.obstacle])
    np.savez_compressed(os.path.join(OUT, "nlp_optima.npz"), **out)


def make_plant_kat():
    out = {}
    for d in sorted(os.listdir(REF_TEST)):
        if not d.startswith("2D_plots_"):
            continue
        xs = np.loadtxt(os.path.join(REF_TEST, d, "planned states.txt"))
        us = np.loadtxt(os.path.join(REF_TEST, d, "control inputs.txt"))
        key = d[len("2D_plots_"):].replace("-", "_")
        out[f"{key}__x"] = xs
        out[f"{key}__u"] = us
        print(d, xs.shape, us.shape)
    np.savez_compressed(os.path.join(OUT, "plant_step_kat.npz"), **out)


def make_forces_kat():
    from oracle.binding import ForcesModelRef
    ref = ForcesModelRef()
    rng = np.random.default_rng(7)
    n = 64
    Z = np.zeros((n, 7)); Pm = np.zeros((n, 10)); ST = np.zeros(n, dtype=np.int32)
    F = np.zeros(n); GF = np.zeros((n, 7)); Cc = np.zeros((n, 5)); JC = np.zeros((n, 5, 7)); H = np.zeros((n, 10)); JH = np.zeros((n, 10, 7))
    for i in range(n):
        z = np.array([rng.uniform(-0.4, 0.4), rng.uniform(-5, 5), rng.uniform(0, 100), rng.uniform(-5, 5),
                      rng.uniform(-0.5, 0.5), rng.uniform(0.5, 30), rng.uniform(-3, 3)])
        p = np.array([z[2] + rng.uniform(-2, 2), z[3] + rng.uniform(-2, 2), rng.uniform(5, 25), z[6] + rng.uniform(-0.2, 0.2),
                      *(np.array([z[2], z[3]] * 3) + rng.uniform(4, 30, 6) * rng.choice([-1, 1], 6))])
        st = 9 if i % 8 == 7 else int(rng.integers(0, 9))
        r = ref.eval(z, p, st)
        Z[i], Pm[i], ST[i] = z, p, st
        F[i], GF[i], Cc[i], JC[i], H[i], JH[i] = r["f"], r["grad_f"], r["c"], r["jac_c"], r["h"], r["jac_h"]
    np.savez_compressed(os.path.join(OUT, "forces_model_kat.npz"), z=Z, p=Pm, stage=ST, f=F, grad_f=GF, c=Cc, jac_c=JC, h=H, jac_h=JH)
    print("forces model KAT:", n, "vectors")


if __name__ == "__main__":
    what = sys.argv[1:] or ["plant", "forces", "optima"]
    if "plant" in what:
        make_plant_kat()
    if "forces" in what:
        make_forces_kat()
    if "optima" in what:
        make_optima()
