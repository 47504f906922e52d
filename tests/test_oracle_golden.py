"""CPU: the oracle (oracle/) pinned against the committed golden vectors (tests/golden/*.npz)."""
import os

import numpy as np
import pytest

from helpers import FAMILIES, cfg_from_golden
from oracle.binding import ForcesModelRef, OracleSolver, REF_PATH
from oracle.ipm_numpy import DenseIPM
from oracle.nlp_numpy import BicycleNLP, NLPConfig, synthetic_instance

# SURVEY.md section 8(c)'s grid: {ZAM-LF, ZAM-CA, USA-LF weights} x N in {10, 30, 50} x 8 instances, the nx = 6 headline family and
# the first-step (tiled reference) cases -- every optimum from cold starts of scipy solvers (tests/golden/make_golden.py)
OPT_FAMILIES = [f"{w}_n{n}_nx5" for w in ("zamlf", "usalf", "zamca", "first") for n in (10, 30, 50)] + ["zamlf_n30_nx6"]


def nearest_basin(x, W_alt, F_alt):
    """index of and distance to the closest recorded local optimum (the nonconvex family has two: pass left / pass right)"""
    d = [np.abs(x - w).max() if np.isfinite(f) else np.inf for w, f in zip(W_alt, F_alt)]
    return int(np.argmin(d)), float(np.min(d))


@pytest.fixture(scope="module")
def optima(golden_dir):
    return np.load(os.path.join(golden_dir, "nlp_optima.npz"))


@pytest.mark.parametrize("fam", OPT_FAMILIES)
def test_oracle_matches_scipy_optima(optima, fam):
    """C oracle optimum == scipy SLSQP optimum (independent solver) to 1e-6; SLSQP itself agrees with scipy's
    trust-constr to the recorded `dtc`."""
    cfg = cfg_from_golden(optima[f"{fam}__cfg"])
    osol = OracleSolver(cfg)
    X0, P, WA, FA, DA = (optima[f"{fam}__{k}"] for k in ("x0", "p", "w_alt", "f_alt", "dtc_alt"))
    assert np.nanmax(DA) < 5e-5, "the two scipy solvers must agree with each other"
    assert not any("oracle" in str(q) for q in optima[f"{fam}__start"]), "no golden may be seeded by the solver under test"
    assert len(X0) >= (3 if fam.startswith("first") else 8)
    basins = []
    for x0, p, wa, fa, da in zip(X0, P, WA, FA, DA):
        r = osol.solve(x0, p)
        assert r["status"] == 1
        assert r["kkt"] <= 1e-8
        k, dist = nearest_basin(r["x"], wa, fa)
        basins.append(k)
        assert abs(r["f"] - fa[k]) <= 1e-7 * max(1.0, abs(fa[k]))
        # scipy's own accuracy: the two scipy solvers differ from each other by `dtc` on this optimum
        assert dist <= max(2e-6, 2 * da[k]), (fam, k, dist)
    print(fam, "basins the oracle lands in (0 = lowest optimum found):", basins)


def test_first_step_brakes_at_friction_cap(optima):
    """tiled reference => a_0* = -sqrt(11.5) (visible in all recorded CasADi runs)."""
    cfg = cfg_from_golden(optima["first_n10_nx5__cfg"])
    r = OracleSolver(cfg).solve(optima["first_n10_nx5__x0"][0], optima["first_n10_nx5__p"][0])
    assert abs(r["x"][1] + np.sqrt(11.5)) < 1e-6


def test_dense_literal_ipm_equals_riccati_oracle():
    """literal 9-row / dense-KKT numpy IPM and the weight-3 / Riccati C oracle walk the same iterates"""
    cfg = NLPConfig(N=10, nx=5)
    nlp = BicycleNLP(cfg)
    lbg, ubg, lbx, ubx = nlp.bounds()
    lbg[0] = -np.inf      # vacuous lower bound of the |.| row, as in the oracle
    for b in range(3):
        x0, p = synthetic_instance(cfg, b)
        # start both from a dynamically consistent guess (zero-input rollout), which the oracle's start-point
        # safeguard leaves untouched, so that the two implementations really walk the same path
        U, X = nlp.split(x0.copy())
        for k in range(cfg.N):
            X[k + 1] = nlp.plant_step(X[k], U[k])
        x0 = np.concatenate([U.ravel(), X.ravel()])
        rd = DenseIPM(nlp).solve(x0, p, lbg=lbg)
        rc = OracleSolver(cfg, literal_friction_row=True).solve(x0, p)
        assert rd["status"] == 1 and rc["status"] == 1 and rd["iters"] == rc["iters"]
        assert np.abs(rd["x"] - rc["x"]).max() < 1e-11


@pytest.mark.parametrize("fam", list(FAMILIES))
def test_nlp_functions_numpy_vs_c(fam):
    cfg, kw = FAMILIES[fam]
    nlp, osol = BicycleNLP(cfg), OracleSolver(cfg)
    rng = np.random.default_rng(1)
    x0, p = synthetic_instance(cfg, 5, **kw)
    w = x0 + rng.normal(0, 0.2, cfg.n_w)
    assert abs(nlp.f(w, p) - osol.objective(w, p)) < 1e-9 * max(1, abs(nlp.f(w, p)))
    assert np.abs(nlp.g(w, p) - osol.constraints(w, p)).max() < 1e-12


def test_plant_step_kat_bit_exact(golden_dir):
    """every recorded row satisfies x[k+1] == step(x[k], u[k]) with ZERO error: forward Euler for the casadi runs
    (optimizer.py:649-650), one RK4 step for the forcespro runs (optimizer.py:97-98,356)."""
    kat = np.load(os.path.join(golden_dir, "plant_step_kat.npz"))
    osol = OracleSolver(NLPConfig(N=10, nx=5))
    runs = sorted({k.rsplit("__", 1)[0] for k in kat.files})
    assert len(runs) == 6
    nrows = 0
    for run in runs:
        xs, us = kat[f"{run}__x"], kat[f"{run}__u"]
        integ = "euler" if run.startswith("casadi") else "rk4"
        for k in range(len(xs) - 1):
            xn = osol.plant_step(xs[k], us[k], integ)
            assert np.array_equal(xn, xs[k + 1]), (run, k, np.abs(xn - xs[k + 1]).max())
            nrows += 1
    assert nrows == 2 * (29 + 29 + 69)


def test_forces_model_kat_dynamics(golden_dir):
    """RK4 stage dynamics of the reference's CasADi-generated C (FORCESNLPsolver_model.c casadi_f2) == oracle RK4"""
    kat = np.load(os.path.join(golden_dir, "forces_model_kat.npz"))
    osol = OracleSolver(NLPConfig(N=10, nx=5))
    n = 0
    for z, st, c in zip(kat["z"], kat["stage"], kat["c"]):
        if st == 9:
            continue            # terminal stage has no dynamics (FORCESNLPsolver_interface.c:145-191)
        xn = osol.plant_step(z[2:7], z[0:2], "rk4")
        assert np.abs(xn - c).max() < 1e-13
        n += 1
    assert n > 40


@pytest.mark.skipif(not os.path.exists(REF_PATH), reason="oracle/_ref not built (needs /root/reference)")
def test_forces_model_ref_reproduces_fixture(golden_dir):
    kat = np.load(os.path.join(golden_dir, "forces_model_kat.npz"))
    ref = ForcesModelRef()
    for i in range(0, 64, 7):
        r = ref.eval(kat["z"][i], kat["p"][i], int(kat["stage"][i]))
        assert np.array_equal(r["h"], kat["h"][i]) and np.array_equal(r["jac_c"], kat["jac_c"][i])
