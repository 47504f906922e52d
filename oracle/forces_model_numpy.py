"""
ORACLE (test infrastructure, not product code) -- numpy restatement of the FORCES-mode stage functions
(SURVEY.md section 8 row a11): what `FORCESNLPsolver_casadi2forces` (test/FORCESNLPsolver/FORCESNLPsolver_interface.c:41-198)
evaluates per stage from the CasADi-generated `casadi_f0..f9` (FORCESNLPsolver_model.c:75-1756), i.e. the model that
`ForcesproOptimizer.solver()` declares (MPC_Planner/optimizer.py:91-245):

  z = [deltaDot, aLong, x, y, delta, v, psi]   (optimizer.py:94)
  p = [x_ref, y_ref, v_des, psi_ref, obstacle centre / front / rear circle (x, y)]   (optimizer.py:124-127)
  f      stage cost (optimizer.py:158-176), terminal cost at the last stage (:178-194)
  c      one RK4 step (h = 0.1) of the kinematic single-track ODE from x = z[2:7] with u = z[0:2] (:91-98)
  h      [aLong^2 + (v * psi_dot)^2 ; 9 squared circle distances], psi_dot = v tan(delta) / 2.578 (:121-149)
  and their gradients / Jacobians with respect to z.

PINNED: against tests/golden/forces_model_kat.npz = 64 random (z, p, stage) through the reference's own generated C compiled
by oracle/Makefile (oracle/_ref), all six outputs.  The weights baked into that generated code are WEIGHTS_MODEL_C below
(read off the compiled model at unit vectors; they differ from the yaml files in test/config_files).
"""
import numpy as np

WHEELBASE_ODE = 2.5789128        # p.a + p.b, FORCESNLPsolver_model.c:334
WHEELBASE_FRICTION = 2.578       # configuration.wheelbase (yaml), optimizer.py:129
EGO_OFFSET = 0.75                # (disc_distance / 2) / 2, configuration.py:82-92, value at FORCESNLPsolver_model.c:877
WEIGHTS_MODEL_C = dict(Q=(2.0, 2.0, 50.0, 0.1, 5.0), R=(2.0, 0.2), P=(4.0, 4.0, 100.0, 0.2, 10.0))


def ode(x, u, l=WHEELBASE_ODE):
    return np.array([x[3] * np.cos(x[4]), x[3] * np.sin(x[4]), u[0], u[1], x[3] / l * np.tan(x[2])])


def ode_jac(x, l=WHEELBASE_ODE):
    """d ode / d x (5,5); d ode / d u is [0 0; 0 0; 1 0; 0 1; 0 0]"""
    F = np.zeros((5, 5))
    F[0, 3], F[0, 4] = np.cos(x[4]), -x[3] * np.sin(x[4])
    F[1, 3], F[1, 4] = np.sin(x[4]), x[3] * np.cos(x[4])
    F[4, 2] = x[3] / l / np.cos(x[2]) ** 2
    F[4, 3] = np.tan(x[2]) / l
    return F


def rk4_with_jacobian(z, dt=0.1, l=WHEELBASE_ODE):
    u, x = z[0:2], z[2:7]
    G = np.zeros((5, 2))
    G[2, 0] = G[3, 1] = 1.0
    T = np.hstack([np.zeros((5, 2)), np.eye(5)])                  # d x / d z
    U = np.hstack([np.eye(2), np.zeros((2, 5))])                  # d u / d z
    ks, dks = [], []
    xs, dxs = x, T
    for a in (0.0, 0.5, 0.5, 1.0):
        if ks:
            xs = x + a * dt * ks[-1]
            dxs = T + a * dt * dks[-1]
        ks.append(ode(xs, u, l))
        dks.append(ode_jac(xs, l) @ dxs + G @ U)
    c = x + dt / 6.0 * (ks[0] + 2 * ks[1] + 2 * ks[2] + ks[3])
    J = T + dt / 6.0 * (dks[0] + 2 * dks[1] + 2 * dks[2] + dks[3])
    return c, J


def stage_functions(z, p, terminal=False, weights=WEIGHTS_MODEL_C, dt=0.1):
    z = np.asarray(z, dtype=np.float64)
    p = np.asarray(p, dtype=np.float64)
    Q, R, P = weights["Q"], weights["R"], weights["P"]
    w = P if terminal else Q
    r = np.array([z[2] - p[0], z[3] - p[1], z[4], z[5] - p[2], z[6] - p[3]])
    f = float(np.dot(w, r * r))
    gf = np.zeros(7)
    gf[2:7] = 2.0 * np.asarray(w) * r
    if not terminal:
        f += R[0] * z[0] ** 2 + R[1] * z[1] ** 2
        gf[0], gf[1] = 2 * R[0] * z[0], 2 * R[1] * z[1]
    c, jc = (None, None) if terminal else rk4_with_jacobian(z, dt)
    h = np.zeros(10)
    jh = np.zeros((10, 7))
    td = np.tan(z[4])
    q = z[5] * z[5] * td / WHEELBASE_FRICTION                    # v * psi_dot
    h[0] = z[1] ** 2 + q ** 2
    jh[0, 1] = 2 * z[1]
    jh[0, 4] = 2 * q * z[5] * z[5] / np.cos(z[4]) ** 2 / WHEELBASE_FRICTION
    jh[0, 5] = 2 * q * 2 * z[5] * td / WHEELBASE_FRICTION
    cs, sn = np.cos(z[6]), np.sin(z[6])
    for e, sg in enumerate((0.0, 1.0, -1.0)):                     # centre, front, rear circle (configuration.py:69-93)
        ex, ey = z[2] + sg * EGO_OFFSET * cs, z[3] + sg * EGO_OFFSET * sn
        for j in range(3):
            ox, oy = p[4 + 2 * j], p[5 + 2 * j]
            row = 1 + 3 * e + j
            h[row] = (ex - ox) ** 2 + (ey - oy) ** 2
            jh[row, 2] = 2 * (ex - ox)
            jh[row, 3] = 2 * (ey - oy)
            jh[row, 6] = 2 * (ex - ox) * (-sg * EGO_OFFSET * sn) + 2 * (ey - oy) * (sg * EGO_OFFSET * cs)
    return dict(f=f, grad_f=gf, c=c, jac_c=jc, h=h, jac_h=jh)
