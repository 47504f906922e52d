"""
ORACLE (test infrastructure, not product code) -- numpy restatement of the CasADi NLP.

This file restates, in plain numpy with hand-derived derivatives, the multiple-shooting NLP that the
reference builds symbolically in `MPC_Planner/optimizer.py`:

  * decision / parameter vector layout ........ optimizer.py:550,552   (w = [vec(U); vec(X)], column-major)
  * cost ...................................... optimizer.py:493-511   (terminal P-term is dead code, :509-510)
  * constraint vector g ....................... optimizer.py:373-411   (friction row, x0 pin, Euler defects,
                                                                        9 obstacle rows per stage = 3 pairs x3)
  * lbg/ubg/lbx/ubx ........................... optimizer.py:413-491
  * kinematic single-track ODE ................ configuration.py:353-368 (l = a+b = 2.5789128)
  * 3-circle vehicle approximation ............ configuration.py:40-93

Only `tests/`, `bench.py`'s cpu_baseline leg, `__graft_entry__.smoke()` and the fixture generator under
`tests/golden/` may import this module.  Nothing under the product package imports it.

PARITY UNPINNED: CasADi/IPOPT is not installable in the build container (no wheel, no network) and the
reference's only test asserts nothing, so the NLP *optimum* is cross-checked against two independent scipy
solvers (see tests/golden/make_golden.py) instead of against IPOPT itself.

nx = 6 appends a decoupled progress state s (s' = v, zero weight, unbounded): SURVEY.md section 0.5.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

WHEELBASE_L = 2.5789128          # parameters_vehicle2: a + b  (value visible at FORCESNLPsolver_model.c:334)
FRICTION_DIV = 2.578             # literal in optimizer.py:378
EGO_LENGTH = 4.508               # parameters_vehicle2.l
EGO_WIDTH = 1.610                # parameters_vehicle2.w


def approximating_circle_radius(length: float, width: float):
    """configuration.py:40-66 -- returns (radius, distance between first and last circle centre)."""
    assert length >= 0 and width >= 0
    if np.isclose(length, 0.0) and np.isclose(width, 0.0):
        return 0.0, 0.0
    square_length = length / 3
    diagonal_square = np.sqrt((square_length / 2) ** 2 + (width / 2) ** 2)
    if diagonal_square > round(diagonal_square, 1):
        approx_radius = round(diagonal_square, 1) + 0.1
    else:
        approx_radius = round(diagonal_square, 1)
    return float(approx_radius), round(square_length * 2, 1)


def circle_centers(x, y, length, width, orientation):
    """configuration.py:69-93 -- centre, front, rear circle centres as a (3,2) array."""
    _, disc_distance = approximating_circle_radius(length, width)
    half = (disc_distance / 2) / 2
    c, s = math.cos(orientation), math.sin(orientation)
    return np.array([[x, y], [x + half * c, y + half * s], [x - half * c, y - half * s]], dtype=np.float64)


@dataclass
class NLPConfig:
    """Everything `Optimizer.__init__` (optimizer.py:34-68) pulls out of the planning configuration."""
    N: int = 10
    nx: int = 5
    dt: float = 0.1
    # weights (yaml `weights_setting`), Q = diag(x, y, steering angle, velocity, heading), R = diag(deltav, a)
    Q: tuple = (2.3, 2.3, 500.0, 0.1, 10.0)
    R: tuple = (2.0, 0.2)
    delta_min: float = -1.066
    delta_max: float = 1.066
    deltav_min: float = -0.4
    deltav_max: float = 0.4
    v_min: float = 0.0
    v_max: float = 50.8
    a_max: float = 11.5
    # static obstacle (position_x, position_y, length, width, orientation); dummy for lane following:
    # configuration.py:471-483 puts it at (-100, 0) with zero extent.
    obstacle: tuple = (-100.0, 0.0, 0.0, 0.0, 0.0)
    ego_length: float = EGO_LENGTH
    ego_width: float = EGO_WIDTH
    wheelbase: float = WHEELBASE_L
    friction_div: float = FRICTION_DIV

    nu: int = field(default=2, init=False)

    @property
    def n_w(self):
        return self.nu * self.N + self.nx * (self.N + 1)

    @property
    def n_g(self):
        return 1 + self.nx * (self.N + 1) + 9 * (self.N + 1)

    @property
    def Qdiag(self):
        q = np.zeros(self.nx)
        q[:5] = self.Q
        return q

    @property
    def Rdiag(self):
        return np.asarray(self.R, dtype=np.float64)

    @property
    def obstacle_centers(self):
        ox, oy, ol, ow, oth = self.obstacle
        return circle_centers(ox, oy, ol, ow, oth)

    @property
    def r_sum(self):
        r_ego, _ = approximating_circle_radius(self.ego_length, self.ego_width)
        r_obs, _ = approximating_circle_radius(self.obstacle[2], self.obstacle[3])
        return r_ego + r_obs

    @property
    def ego_offset(self):
        _, dd = approximating_circle_radius(self.ego_length, self.ego_width)
        return (dd / 2) / 2


# --------------------------------------------------------------------------------------------------------
# named weight sets of the committed configs
# --------------------------------------------------------------------------------------------------------
WEIGHTS_ZAM_LF = dict(Q=(2.3, 2.3, 500.0, 0.1, 10.0), R=(2.0, 0.2))        # config_LF_ZAM_Over-1_1.yaml:20-26
WEIGHTS_ZAM_CA = dict(Q=(2.3, 2.3, 500.0, 0.1, 160.0), R=(0.8, 0.8))       # config_CA_ZAM_Over-1_1.yaml:39-45
WEIGHTS_USA_LF = dict(Q=(200.0, 200.0, 150.0, 150.0, 1.0), R=(100.0, 10.0))  # config_LF_USA_Lanker-2_18_T-1.yaml:20-26
OBSTACLE_ZAM = (59.948, 0.08323, 6.0, 3.5, 0.07759)                         # ZAM_Over-1_1.xml:3235-3258


class BicycleNLP:
    """f, grad f, g, jac g, Hessian of the Lagrangian for one instance; dense outputs (small problems)."""

    def __init__(self, cfg: NLPConfig):
        self.cfg = cfg
        self.N, self.nx, self.nu = cfg.N, cfg.nx, cfg.nu
        self.n_w, self.n_g = cfg.n_w, cfg.n_g
        self.obst = cfg.obstacle_centers
        self.rho = cfg.ego_offset

    # ---- layout ----------------------------------------------------------------------------------------
    def iu(self, k):            # index of u_k[0] in w
        return self.nu * k

    def ix(self, k):            # index of x_k[0] in w
        return self.nu * self.N + self.nx * k

    def split(self, w):
        N, nx, nu = self.N, self.nx, self.nu
        w = np.asarray(w, dtype=np.float64).ravel()
        U = w[: nu * N].reshape(N, nu)
        X = w[nu * N:].reshape(N + 1, nx)
        return U, X

    def row_friction(self):
        return 0

    def row_x0(self):
        return 1

    def row_defect(self, k):    # rows of x_{k+1} - x_k - dt f(x_k,u_k), k = 0..N-1
        return 1 + self.nx * (k + 1)

    def row_obst(self, k):      # 9 rows for stage k
        return 1 + self.nx * (self.N + 1) + 9 * k

    # ---- model -----------------------------------------------------------------------------------------
    def ode(self, x, u):
        """configuration.py:353-368."""
        l = self.cfg.wheelbase
        out = np.zeros(self.nx)
        out[0] = x[3] * math.cos(x[4])
        out[1] = x[3] * math.sin(x[4])
        out[2] = u[0]
        out[3] = u[1]
        out[4] = x[3] / l * math.tan(x[2])
        if self.nx == 6:
            out[5] = x[3]
        return out

    def ode_jac(self, x, u):
        l = self.cfg.wheelbase
        nx = self.nx
        Fx = np.zeros((nx, nx))
        c, s = math.cos(x[4]), math.sin(x[4])
        Fx[0, 3] = c
        Fx[0, 4] = -x[3] * s
        Fx[1, 3] = s
        Fx[1, 4] = x[3] * c
        cd = math.cos(x[2])
        Fx[4, 2] = x[3] / (l * cd * cd)
        Fx[4, 3] = math.tan(x[2]) / l
        if nx == 6:
            Fx[5, 3] = 1.0
        Fu = np.zeros((nx, 2))
        Fu[2, 0] = 1.0
        Fu[3, 1] = 1.0
        return Fx, Fu

    def ode_hess_contract(self, x, lam):
        """sum_r lam[r] * Hessian_x f_r  (f is linear in u and has no x-u cross terms)."""
        l = self.cfg.wheelbase
        nx = self.nx
        H = np.zeros((nx, nx))
        c, s = math.cos(x[4]), math.sin(x[4])
        v = x[3]
        H[3, 4] += lam[0] * (-s) + lam[1] * c
        H[4, 4] += lam[0] * (-v * c) + lam[1] * (-v * s)
        cd = math.cos(x[2])
        td = math.tan(x[2])
        H[2, 3] += lam[4] / (l * cd * cd)
        H[2, 2] += lam[4] * v * 2.0 * td / (l * cd * cd)
        H[4, 3] = H[3, 4]
        H[3, 2] = H[2, 3]
        return H

    def plant_step(self, x, u):
        """forward Euler, optimizer.py:649-650."""
        return np.asarray(x, dtype=np.float64) + self.cfg.dt * self.ode(x, u)

    def obstacle_rows(self, x, want_hess=False):
        """three distinct distances (pairs (0,0),(1,1),(2,2): optimizer.py:395-403), Jacobian wrt
        (sx, sy, psi) and optionally the three 3x3 Hessians."""
        rho = self.rho
        c, s = math.cos(x[4]), math.sin(x[4])
        d = np.zeros(3)
        J = np.zeros((3, 3))
        Hs = np.zeros((3, 3, 3))
        for j, sg in enumerate((0.0, 1.0, -1.0)):
            cx = x[0] + sg * rho * c - self.obst[j, 0]
            cy = x[1] + sg * rho * s - self.obst[j, 1]
            dist = math.sqrt(cx * cx + cy * cy)
            ex, ey = cx / dist, cy / dist
            tx, ty = -sg * rho * s, sg * rho * c           # d centre / d psi
            d[j] = dist
            J[j] = (ex, ey, ex * tx + ey * ty)
            if want_hess:
                M = np.array([[1 - ex * ex, -ex * ey], [-ex * ey, 1 - ey * ey]]) / dist
                T = np.array([[1.0, 0.0, tx], [0.0, 1.0, ty]])
                H = T.T @ M @ T
                nxx, nyy = -sg * rho * c, -sg * rho * s    # d2 centre / d psi2
                H[2, 2] += ex * nxx + ey * nyy
                Hs[j] = H
        return d, J, Hs

    def friction(self, u0, x0, want_hess=False):
        """optimizer.py:378: sqrt((a^2 + v*(tan(delta)*v/2.578))^2) == |y|; derivative of |.| is sign(.) with
        sign(0) = 0 (CasADi simplifies sqrt(sq(y)) to fabs(y))."""
        kap = self.cfg.friction_div
        a, dl, v = u0[1], x0[2], x0[3]
        td = math.tan(dl)
        cd2 = math.cos(dl) ** 2
        y = a * a + v * (td * v / kap)
        sg = float(np.sign(y))
        val = abs(y)
        # gradient entries wrt (a, delta, v)
        g = sg * np.array([2 * a, v * v / (kap * cd2), 2 * v * td / kap])
        H = None
        if want_hess:
            H = sg * np.array([[2.0, 0.0, 0.0],
                               [0.0, 2 * v * v * td / (kap * cd2), 2 * v / (kap * cd2)],
                               [0.0, 2 * v / (kap * cd2), 2 * td / kap]])
        return val, g, H

    # ---- NLP functions ---------------------------------------------------------------------------------
    def f(self, w, p):
        U, X = self.split(w)
        _, Xr = self.split(p)
        Q, R = self.cfg.Qdiag, self.cfg.Rdiag
        obj = 0.0
        for i in range(self.N):
            e = X[i] - Xr[i + 1]
            obj += float(np.dot(Q * e, e) + np.dot(R * U[i], U[i]))
        return obj

    def grad(self, w, p):
        U, X = self.split(w)
        _, Xr = self.split(p)
        Q, R = self.cfg.Qdiag, self.cfg.Rdiag
        g = np.zeros(self.n_w)
        for i in range(self.N):
            g[self.iu(i): self.iu(i) + 2] = 2 * R * U[i]
            g[self.ix(i): self.ix(i) + self.nx] = 2 * Q * (X[i] - Xr[i + 1])
        return g

    def g(self, w, p):
        U, X = self.split(w)
        _, Xr = self.split(p)
        nx, N, dt = self.nx, self.N, self.cfg.dt
        out = np.zeros(self.n_g)
        out[0] = self.friction(U[0], X[0])[0]
        out[1:1 + nx] = X[0] - Xr[0]
        for i in range(N):
            out[self.row_defect(i): self.row_defect(i) + nx] = X[i + 1] - (self.ode(X[i], U[i]) * dt + X[i])
        for i in range(N + 1):
            d, _, _ = self.obstacle_rows(X[i])
            out[self.row_obst(i): self.row_obst(i) + 9] = np.repeat(d, 3)
        return out

    def jac(self, w, p):
        U, X = self.split(w)
        nx, N, dt = self.nx, self.N, self.cfg.dt
        J = np.zeros((self.n_g, self.n_w))
        _, gf, _ = self.friction(U[0], X[0])
        J[0, self.iu(0) + 1] = gf[0]
        J[0, self.ix(0) + 2] = gf[1]
        J[0, self.ix(0) + 3] = gf[2]
        J[1:1 + nx, self.ix(0): self.ix(0) + nx] = np.eye(nx)
        for i in range(N):
            Fx, Fu = self.ode_jac(X[i], U[i])
            r = self.row_defect(i)
            J[r:r + nx, self.ix(i + 1): self.ix(i + 1) + nx] = np.eye(nx)
            J[r:r + nx, self.ix(i): self.ix(i) + nx] = -(np.eye(nx) + dt * Fx)
            J[r:r + nx, self.iu(i): self.iu(i) + 2] = -dt * Fu
        for i in range(N + 1):
            _, Jo, _ = self.obstacle_rows(X[i])
            r = self.row_obst(i)
            for j in range(3):
                for rep in range(3):
                    J[r + 3 * j + rep, self.ix(i) + 0] = Jo[j, 0]
                    J[r + 3 * j + rep, self.ix(i) + 1] = Jo[j, 1]
                    J[r + 3 * j + rep, self.ix(i) + 4] = Jo[j, 2]
        return J

    def hess_lag(self, w, p, sigma, lam):
        """sigma * Hess f + sum_i lam_i Hess g_i   (dense, symmetric)."""
        U, X = self.split(w)
        nx, N, dt = self.nx, self.N, self.cfg.dt
        Q, R = self.cfg.Qdiag, self.cfg.Rdiag
        H = np.zeros((self.n_w, self.n_w))
        for i in range(N):
            iu, ix = self.iu(i), self.ix(i)
            H[iu:iu + 2, iu:iu + 2] += sigma * 2 * np.diag(R)
            H[ix:ix + nx, ix:ix + nx] += sigma * 2 * np.diag(Q)
            lam_d = lam[self.row_defect(i): self.row_defect(i) + nx]
            H[ix:ix + nx, ix:ix + nx] += -dt * self.ode_hess_contract(X[i], lam_d)
        _, _, Hf = self.friction(U[0], X[0], want_hess=True)
        idx = [self.iu(0) + 1, self.ix(0) + 2, self.ix(0) + 3]
        for a_ in range(3):
            for b_ in range(3):
                H[idx[a_], idx[b_]] += lam[0] * Hf[a_, b_]
        for i in range(N + 1):
            _, _, Hs = self.obstacle_rows(X[i], want_hess=True)
            r = self.row_obst(i)
            ix = self.ix(i)
            cols = [ix + 0, ix + 1, ix + 4]
            for j in range(3):
                lj = lam[r + 3 * j] + lam[r + 3 * j + 1] + lam[r + 3 * j + 2]
                for a_ in range(3):
                    for b_ in range(3):
                        H[cols[a_], cols[b_]] += lj * Hs[j, a_, b_]
        return H

    # ---- bounds (optimizer.py:413-491) -------------------------------------------------------------------
    def bounds(self):
        c = self.cfg
        N, nx = self.N, self.nx
        lbg = [0.0]
        ubg = [c.a_max]
        lbg += [0.0] * (nx * (N + 1))
        ubg += [0.0] * (nx * (N + 1))
        lbg += [c.r_sum] * (9 * (N + 1))
        ubg += [np.inf] * (9 * (N + 1))
        lbx, ubx = [], []
        for _ in range(N):
            lbx += [c.deltav_min, -np.inf]
            ubx += [c.deltav_max, c.a_max]
        for _ in range(N + 1):
            lo = [-np.inf, -np.inf, c.delta_min, c.v_min, -np.inf]
            hi = [np.inf, np.inf, c.delta_max, c.v_max, np.inf]
            if nx == 6:
                lo.append(-np.inf)
                hi.append(np.inf)
            lbx += lo
            ubx += hi
        return (np.array(lbg), np.array(ubg), np.array(lbx), np.array(ubx))


# --------------------------------------------------------------------------------------------------------
# synthetic instance generator; shared by tests and bench
# --------------------------------------------------------------------------------------------------------
def synthetic_instance(cfg: NLPConfig, b: int, v_range=(5.0, 25.0), heading=None):
    """Returns (x0_warm, p) for instance b: constant-curvature reference arc, perturbed initial state."""
    rng = np.random.default_rng(20240929 + b)
    N, nx, dt = cfg.N, cfg.nx, cfg.dt
    kappa = rng.uniform(-0.02, 0.02)
    v_ref = rng.uniform(*v_range)
    psi0 = rng.uniform(-math.pi, math.pi) if heading is None else heading
    lat = rng.uniform(-0.5, 0.5)
    dpsi = rng.uniform(-0.05, 0.05)
    vfac = rng.uniform(0.9, 1.1)
    ds = v_ref * dt
    Xr = np.zeros((N + 1, nx))
    # path points k = 0..N along the arc starting at the origin
    px = py = 0.0
    th = psi0
    pts = []
    for k in range(N + 1):
        pts.append((px, py, th))
        px += ds * math.cos(th)
        py += ds * math.sin(th)
        th += kappa * ds
    x_init = np.zeros(nx)
    x_init[0] = pts[0][0] - lat * math.sin(psi0)
    x_init[1] = pts[0][1] + lat * math.cos(psi0)
    x_init[2] = 0.0
    x_init[3] = v_ref * vfac
    x_init[4] = psi0 + dpsi
    Xr[0] = x_init
    for k in range(1, N + 1):
        Xr[k, 0], Xr[k, 1], Xr[k, 2], Xr[k, 3], Xr[k, 4] = pts[k][0], pts[k][1], 0.0, v_ref, pts[k][2]
        if nx == 6:
            Xr[k, 5] = 0.0
    p = np.concatenate([np.zeros(2 * N), Xr.ravel()])
    x0 = np.concatenate([np.zeros(2 * N), np.tile(x_init, N + 1)])
    return x0, p


def synthetic_batch(cfg: NLPConfig, B: int, start: int = 0, **kw):
    xs, ps = zip(*(synthetic_instance(cfg, start + b, **kw) for b in range(B)))
    return np.ascontiguousarray(np.stack(xs)), np.ascontiguousarray(np.stack(ps))
