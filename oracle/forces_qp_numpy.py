"""
ORACLE (test infrastructure, not product code) -- the FORCES-mode solve of SURVEY.md section 8 row f3 restated in numpy.

`ForcesproOptimizer.solver()` (MPC_Planner/optimizer.py:196-245) asks FORCESPRO for an SQP solver with ONE quadratic
programme per call (`sqp_nlp.maxqps = 1`), a BFGS Hessian initialised to 2.5 I (`nlp.bfgs_init`, with a single QP it is never
updated) and Hessian regularisation 5e-6; the generated solver itself is a closed, licence-locked binary.  One call is
therefore one step of sequential quadratic programming from the caller's guess zbar = problem["x0"]:

    min_dz  sum_k  grad f_k(zbar_k)' dz_k + 1/2 dz_k' H_k dz_k     (H_k diagonal, see hessian_diag)
    s.t.    x_1 = xinit                                   (xinitidx = z[2:7], optimizer.py:222)
            x_{k+1} = c(zbar_k) + C_k dz_k,  k = 1..N-1    (RK4 dynamics, E = [0 I], optimizer.py:91-98, 219)
            lb <= zbar_k + dz_k <= ub                      (optimizer.py:100-110; the states of stage 1 are fixed by xinit)
            hl <= h(zbar_k) + J_k dz_k <= hu               (optimizer.py:121-149; the lower bound 0 of the friction row is
                                                            vacuous for a sum of squares and carries no constraint)
    z+ = zbar + dz

solved by a primal-dual interior-point method (Mehrotra predictor-corrector).  Here the Newton systems are assembled as
ONE dense KKT matrix and handed to numpy; the GPU path eliminates stage by stage (Riccati) -- two independent linear
algebra routes to the same iterates.  PARITY with FORCESPRO: unpinned (closed binary); the QP solutions are pinned
against scipy (tests/test_forces_qp.py).
"""
import numpy as np

from . import forces_model_numpy as FM

REG_HESSIAN = 5e-6


def hessian_diag(weights, N, mode=0):
    """(N,7) diagonal QP Hessian; mode 0: exact Hessian of the least-squares cost (Gauss-Newton), mode 1: the literal
    `bfgs_init = 2.5 I` of optimizer.py:234 (see csrc/mpc_forces_qp.h: forces_hessian_diag)."""
    H = np.zeros((N, 7))
    for k in range(N):
        if mode == 1:
            H[k] = 2.5
        elif k < N - 1:
            H[k] = 2.0 * np.concatenate((weights["R"], weights["Q"]))
        else:
            H[k] = 2.0 * np.concatenate(([0.0, 0.0], weights["P"]))
    return H + REG_HESSIAN

IPM_MAX_IT = 60
IPM_TOL = 1e-4          # residuals (dual, primal, equality); the accuracy an SQP step needs, see csrc/mpc_forces_qp.h
IPM_TOL_MU = 1e-6       # complementarity gap


def build_qp(zbar, params, xinit, lb, ub, hl, hu, weights=FM.WEIGHTS_MODEL_C, dt=0.1):
    """zbar (N,7), params (N,10), xinit (5) -> list of per-stage dicts: g (7), C (5,7), c (5), G (m,7), d (m)  [G dz <= d]"""
    N = zbar.shape[0]
    st = []
    for k in range(N):
        r = FM.stage_functions(zbar[k], params[k], terminal=(k == N - 1), weights=weights, dt=dt)
        rows, rhs = [], []
        for i in range(7):
            if k == 0 and i >= 2:
                continue                                    # fixed by xinit
            e = np.zeros(7)
            e[i] = 1.0
            if np.isfinite(lb[i]):
                rows.append(-e)
                rhs.append(zbar[k, i] - lb[i])
            if np.isfinite(ub[i]):
                rows.append(e)
                rhs.append(ub[i] - zbar[k, i])
        for j in range(10):
            if np.isfinite(hu[j]):
                rows.append(r["jac_h"][j])
                rhs.append(hu[j] - r["h"][j])
            if j > 0 and np.isfinite(hl[j]):
                rows.append(-r["jac_h"][j])
                rhs.append(r["h"][j] - hl[j])
        c, C = (None, None)
        if k < N - 1:
            rr = FM.stage_functions(zbar[k], params[k], terminal=False, weights=weights, dt=dt)
            c, C = rr["c"], rr["jac_c"]
        st.append(dict(g=r["grad_f"], C=C, c=c, G=np.array(rows), d=np.array(rhs)))
    return st


def solve_qp(st, zbar, xinit, Hd, max_it=IPM_MAX_IT, tol=IPM_TOL, tol_mu=IPM_TOL_MU):
    """Mehrotra predictor-corrector on the stage-structured QP; returns (dz (N,7), iterations, converged, kkt)."""
    N = len(st)
    nz = 7 * N
    m = [s["G"].shape[0] for s in st]
    M = sum(m)
    ne = 5 * N                                                # x_1 = xinit, N-1 dynamics rows
    # equality rows:  A dz = b
    A = np.zeros((ne, nz))
    b = np.zeros(ne)
    A[0:5, 2:7] = np.eye(5)
    b[0:5] = xinit - zbar[0, 2:7]
    for k in range(N - 1):
        r0 = 5 * (k + 1)
        A[r0:r0 + 5, 7 * (k + 1) + 2:7 * (k + 1) + 7] = np.eye(5)
        A[r0:r0 + 5, 7 * k:7 * k + 7] = -st[k]["C"]
        b[r0:r0 + 5] = st[k]["c"] - zbar[k + 1, 2:7]
    G = np.zeros((M, nz))
    d = np.zeros(M)
    g = np.zeros(nz)
    o = 0
    for k in range(N):
        G[o:o + m[k], 7 * k:7 * k + 7] = st[k]["G"]
        d[o:o + m[k]] = st[k]["d"]
        g[7 * k:7 * k + 7] = st[k]["g"]
        o += m[k]
    hvec = np.asarray(Hd, dtype=np.float64).ravel()
    x = np.zeros(nz)
    s = np.maximum(d, 1.0)
    lam = 1.0 / s                                             # centred start: s * lam = 1 on every row
    pi = np.zeros(ne)
    it, conv, kkt = 0, False, np.inf
    for it in range(max_it + 1):
        rd = hvec * x + g + G.T @ lam + A.T @ pi
        rp = G @ x + s - d
        re = A @ x - b
        mu = float(s @ lam) / M
        gscale = max(1.0, np.abs(g).max())                    # dual residual relative to the largest cost gradient
        rmax = max(np.abs(rp).max(), np.abs(re).max())
        rdmax = np.abs(rd).max()
        kkt = max(rmax, rdmax / gscale, mu)
        if not np.isfinite(kkt):
            break
        if rmax <= tol and rdmax <= tol * gscale and mu <= tol_mu:
            conv = True
            break
        if mu > 1e6 or it == max_it:                          # diverging multipliers: inconsistent linearised constraints
            break
        D = lam / s
        K = np.zeros((nz + ne, nz + ne))
        K[:nz, :nz] = np.diag(hvec) + G.T @ (D[:, None] * G)
        K[:nz, nz:] = A.T
        K[nz:, :nz] = A

        # affine step (rc = s*lam)
        def solve(rc):
            rhs1 = -(rd + G.T @ (D * rp - rc / s))
            sol = np.linalg.solve(K, np.concatenate((rhs1, -re)))
            dx, dpi = sol[:nz], sol[nz:]
            ds = -rp - G @ dx
            dlam = -(rc + lam * ds) / s
            return dx, ds, dlam, dpi

        def step_max(v, dv):
            neg = dv < 0
            return np.inf if not neg.any() else float(np.min(-v[neg] / dv[neg]))

        def step_len(v, dv):
            return min(1.0, step_max(v, dv))
        dxa, dsa, dla, dpa = solve(s * lam)
        a_aff = min(step_len(s, dsa), step_len(lam, dla))
        mu_aff = float((s + a_aff * dsa) @ (lam + a_aff * dla)) / M
        sigma = (mu_aff / mu) ** 3
        dx, ds, dl, dp = solve(s * lam + dsa * dla - sigma * mu)
        a_p = min(1.0, 0.995 * step_max(s, ds))
        a_d = min(1.0, 0.995 * step_max(lam, dl))
        x = x + a_p * dx
        s = s + a_p * ds
        lam = lam + a_d * dl
        pi = pi + a_d * dp
    return x.reshape(N, 7), it, conv, kkt


def sqp_step(zbar, params, xinit, lb, ub, hl, hu, weights=FM.WEIGHTS_MODEL_C, dt=0.1, mode=0):
    st = build_qp(zbar, params, xinit, lb, ub, hl, hu, weights, dt)
    dz, it, conv, kkt = solve_qp(st, zbar, xinit, hessian_diag(weights, zbar.shape[0], mode))
    return zbar + dz, it, conv, kkt
