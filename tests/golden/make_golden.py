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


def make_optima():
    fam = {
        "zamlf_n10_nx5": (NLPConfig(N=10, nx=5, **WEIGHTS_ZAM_LF), "syn", 8),
        "zamlf_n30_nx5": (NLPConfig(N=30, nx=5, **WEIGHTS_ZAM_LF), "syn", 6),
        "zamlf_n30_nx6": (NLPConfig(N=30, nx=6, **WEIGHTS_ZAM_LF), "syn", 6),
        "usalf_n50_nx5": (NLPConfig(N=50, nx=5, **WEIGHTS_USA_LF), "syn_usa", 3),
        "zamca_n30_nx5": (NLPConfig(N=30, nx=5, obstacle=OBSTACLE_ZAM, **WEIGHTS_ZAM_CA), "ca", 4),
        "first_n10_nx5": (NLPConfig(N=10, nx=5, **WEIGHTS_ZAM_LF), "first", 3),
        "first_n30_nx5": (NLPConfig(N=30, nx=5, **WEIGHTS_ZAM_LF), "first", 1),
    }
    first_states = [[29.9948, -1.1501, 0.0, 20.0, 0.03495],       # ZAM_Over-1_1.xml:3260-3282
                    [0.0, 0.0, 0.0, 6.8062, -0.4268],              # USA_Lanker-2_18_T-1.xml:113282-113317
                    [10.0, 3.0, 0.0, 12.0, 1.2]]
    out = {}
    for name, (cfg, kind, count) in fam.items():
        nlp = BicycleNLP(cfg)
        X0, P, W, F, DTC, START = [], [], [], [], [], []
        for b in range(count):
            if kind == "syn":
                x0, p = synthetic_instance(cfg, b)
            elif kind == "syn_usa":
                x0, p = synthetic_instance(cfg, b, v_range=(5.0, 9.0))
            elif kind == "ca":
                x0, p = ca_instance(cfg, b)
            else:
                x0, p = first_step_instance(cfg, first_states[b])
            t = time.time()
            start = "x0"
            r = scipy_slsqp(nlp, x0, p)
            if kind == "ca":
                # nonconvex (pass left / right): make sure SLSQP and the oracle sit in the same basin by
                # polishing from the oracle's answer when the cold-start optima differ; recorded in `start`
                from oracle.binding import OracleSolver
                ro = OracleSolver(cfg).solve(x0, p)
                if not r.success or np.abs(r.x - ro["x"]).max() > 1e-4:
                    r2 = scipy_slsqp(nlp, ro["x"], p)
                    print(f"   [ca] cold SLSQP f={r.fun:.6f} (ok={r.success}) vs oracle f={ro['f']:.6f}; polished f={r2.fun:.6f}")
                    r, start = r2, "oracle"
            t1 = time.time() - t
            # second, independent solver (interior-point trust-region).  From the cold start it sometimes stops at
            # its iteration limit far from any optimum; then it is restarted from a perturbation of the SLSQP
            # point and acts as an independent KKT check of that point (recorded in `tcstart`).
            tcstart = "x0"
            rt = scipy_trust_constr(nlp, r.x if kind == "ca" else x0, p)
            dtc = float(np.abs(rt.x - r.x).max())
            if dtc > 1e-5:
                pert = np.random.default_rng(b).normal(0.0, 1e-3, r.x.size)
                rt = scipy_trust_constr(nlp, r.x + pert, p)
                dtc = float(np.abs(rt.x - r.x).max())
                tcstart = "slsqp+1e-3"
            print(f"{name}[{b}] slsqp ok={r.success} nit={r.nit} f={r.fun:.9f} ({t1:.1f}s)  trust-constr f={rt.fun:.9f} |dw|={dtc:.2e}",
                  flush=True)
            X0.append(x0); P.append(p); W.append(r.x); F.append(r.fun); DTC.append(dtc); START.append(start + '|' + tcstart)
        out[f"{name}__x0"] = np.array(X0)
        out[f"{name}__p"] = np.array(P)
        out[f"{name}__w"] = np.array(W)
        out[f"{name}__f"] = np.array(F)
        out[f"{name}__dtc"] = np.array(DTC)
        out[f"{name}__start"] = np.array(START)
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
