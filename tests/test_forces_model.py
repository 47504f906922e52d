"""Row a11: the FORCES-mode stage functions (FORCESNLPsolver_model.c through FORCESNLPsolver_interface.c:41-198).
CPU: the numpy restatement against the golden vectors generated from the reference's own compiled C (oracle/_ref).
GPU: mpc_forces_stage_eval against the same golden vectors and against the restatement on a large random batch."""
import os

import numpy as np
import pytest

from helpers import pkg
from oracle import forces_model_numpy as F

NAMES = ("f", "grad_f", "c", "jac_c", "h", "jac_h")


def rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(1.0, np.abs(np.asarray(b)).max())


def test_restatement_matches_reference_generated_code(golden_dir):
    k = np.load(os.path.join(golden_dir, "forces_model_kat.npz"))
    for i in range(len(k["stage"])):
        term = int(k["stage"][i]) == 9                       # last stage of the N = 10 solver: objectiveN, no dynamics
        r = F.stage_functions(k["z"][i], k["p"][i], terminal=term)
        for name in NAMES:
            if term and name in ("c", "jac_c"):
                continue
            assert rel(r[name], k[name][i]) < 1e-14, (i, name)


def test_restatement_jacobians_by_finite_differences():
    rng = np.random.default_rng(1)
    z = np.array([0.1, 0.5, 1.0, 2.0, 0.3, 5.0, 0.4]) + 0.1 * rng.normal(size=7)
    p = np.array([1.5, 2.5, 6.0, 0.35, 10, 1, 11, 1.2, 9, 0.8])
    r = F.stage_functions(z, p)
    eps = 1e-6
    for name, jac in (("c", "jac_c"), ("h", "jac_h"), ("f", "grad_f")):
        J = np.zeros(np.shape(r[jac]))
        for j in range(7):
            dz = np.zeros(7)
            dz[j] = eps
            d = (np.asarray(F.stage_functions(z + dz, p)[name]) - np.asarray(F.stage_functions(z - dz, p)[name])) / (2 * eps)
            J[..., j] = d
        assert np.abs(J - r[jac]).max() < 1e-6 * max(1.0, np.abs(r[jac]).max())


@pytest.mark.gpu
def test_gpu_stage_functions_match_golden_and_restatement(golden_dir):
    w = F.WEIGHTS_MODEL_C
    s = pkg.BatchedMPCSolver(10, 5, Q=w["Q"], R=w["R"], P=w["P"])
    k = np.load(os.path.join(golden_dir, "forces_model_kat.npz"))
    term = k["stage"] == 9
    for sel, t in ((~term, False), (term, True)):
        g = s.forces_stage_eval(k["z"][sel], k["p"][sel], terminal=t)
        for name in NAMES:
            if t and name in ("c", "jac_c"):
                assert g[name] is None
                continue
            assert rel(g[name], k[name][sel]) < 1e-13, name
    rng = np.random.default_rng(2)
    B = 5000
    z = rng.normal(size=(B, 7)) * [0.3, 3, 30, 5, 0.4, 8, 1.0] + [0, 0, 40, 0, 0, 12, 0]
    p = rng.normal(size=(B, 10)) * 20 + 30
    g = s.forces_stage_eval(z, p)
    for b in range(0, B, 250):
        r = F.stage_functions(z[b], p[b])
        for name in NAMES:
            assert rel(g[name][b], r[name]) < 1e-12, (b, name)
