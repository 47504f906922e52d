"""
Synthetic workloads of BASELINE.json's configurations -- used by bench.py, tools/configs_bench.py and the
tests.  Product-side code: imports nothing from oracle/ (tests/test_workloads.py checks that the rows equal the generator the
oracle tests use).

  family            BASELINE config   horizon  nx  weights (reference yaml)                           obstacle
  zamlf_n30_nx6     2 / headline      30       6   config_LF_ZAM_Over-1_1.yaml:19-31                  dummy (-100, 0)
  zamca_n30_nx5     3                 30       5   config_CA_ZAM_Over-1_1.yaml:38-50                  ZAM_Over-1_1.xml:3235-3258
  usalf_n50_nx5     4                 50       5   config_LF_USA_Lanker-2_18_T-1.yaml:19-31           dummy
  tutlf_n30_nx5     5 ("Tutorial")    30       5   config_LF_ZAM_Over-1_1.yaml:19-31 (the reference      dummy
                                                   ships no yaml for ZAM_Tutorial_Urban-3_2; its planning
                                                   problem 11 is slow urban lane following: 9 m/s start, 3.3 m/s desired)
  mixed sweep       5                 all four, dealt row by row (global row g -> family g % 4), 4096 rows per shard

Every instance b draws from numpy.random.default_rng(20240929 + b): constant-curvature reference arc, perturbed initial state,
warm start = [0 ; tile(X_ref[:, 0])] (the layout of optimizer.py:550).
"""
from __future__ import annotations

import importlib
import math
import os
import sys
from dataclasses import dataclass

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SEED0 = 20240929
EGO_LENGTH, EGO_WIDTH = 4.508, 1.610          # parameters_vehicle2


@dataclass(frozen=True)
class Family:
    name: str
    N: int
    nx: int
    Q: tuple
    R: tuple
    obstacle: tuple = (-100.0, 0.0, 0.0, 0.0, 0.0)      # (x, y, length, width, orientation); configuration.py:471-483 dummy
    dt: float = 0.1
    v_range: tuple = (5.0, 25.0)
    kind: str = "arc"                                     # "arc": lane following on a synthetic arc; "ca": ZAM_Over-1_1 cold starts

    @property
    def n_w(self):
        return 2 * self.N + self.nx * (self.N + 1)

    @property
    def n_g(self):
        return 1 + self.nx * (self.N + 1) + 9 * (self.N + 1)

    @property
    def Qdiag(self):
        q = np.zeros(self.nx)
        q[:5] = self.Q
        return q


FAMILIES = {
    "zamlf_n30_nx6": Family("zamlf_n30_nx6", 30, 6, (2.3, 2.3, 500.0, 0.1, 10.0), (2.0, 0.2)),
    "zamca_n30_nx5": Family("zamca_n30_nx5", 30, 5, (2.3, 2.3, 500.0, 0.1, 160.0), (0.8, 0.8),
                            obstacle=(59.948, 0.08323, 6.0, 3.5, 0.07759), kind="ca"),
    "usalf_n50_nx5": Family("usalf_n50_nx5", 50, 5, (200.0, 200.0, 150.0, 150.0, 1.0), (100.0, 10.0), v_range=(5.0, 9.0)),
    "tutlf_n30_nx5": Family("tutlf_n30_nx5", 30, 5, (2.3, 2.3, 500.0, 0.1, 10.0), (2.0, 0.2), v_range=(3.0, 9.0)),
}
MIXED_ORDER = ("zamlf_n30_nx6", "zamca_n30_nx5", "usalf_n50_nx5", "tutlf_n30_nx5")        # BASELINE.json configs[4]: ZAM_Over + USA_Lanker + Tutorial
MIXED_TOTAL, MIXED_SHARD = 32768, 4096


def _opt():
    return importlib.import_module("motion-planning-for-autonomous-driving-with-mpc_amd.optimizer")


def arc_instance(fam: Family, b: int, heading=None):
    """(x0_warm, p) of instance b: SURVEY.md section 8(d) generator"""
    rng = np.random.default_rng(SEED0 + b)
    N, nx, dt = fam.N, fam.nx, fam.dt
    kappa = rng.uniform(-0.02, 0.02)
    v_ref = rng.uniform(*fam.v_range)
    psi0 = rng.uniform(-math.pi, math.pi) if heading is None else heading
    lat = rng.uniform(-0.5, 0.5)
    dpsi = rng.uniform(-0.05, 0.05)
    vfac = rng.uniform(0.9, 1.1)
    ds = v_ref * dt
    Xr = np.zeros((N + 1, nx))
    px = py = 0.0
    th = psi0
    pts = []
    for _ in range(N + 1):
        pts.append((px, py, th))
        px += ds * math.cos(th)
        py += ds * math.sin(th)
        th += kappa * ds
    x_init = np.zeros(nx)
    x_init[0] = pts[0][0] - lat * math.sin(psi0)
    x_init[1] = pts[0][1] + lat * math.cos(psi0)
    x_init[3] = v_ref * vfac
    x_init[4] = psi0 + dpsi
    Xr[0] = x_init
    for k in range(1, N + 1):
        Xr[k, 0], Xr[k, 1], Xr[k, 3], Xr[k, 4] = pts[k][0], pts[k][1], v_ref, pts[k][2]
    return np.concatenate([np.zeros(2 * N), np.tile(x_init, N + 1)]), np.concatenate([np.zeros(2 * N), Xr.ravel()])


def ca_instance(fam: Family, b: int):
    """ZAM_Over-1_1 collision avoidance: perturbed ego initial states (ZAM_Over-1_1.xml:3260-3282), straight reference through the
    obstacle, cold start"""
    rng = np.random.default_rng(SEED0 + b)
    psi = 0.03495
    x_init = np.array([29.9948 + rng.uniform(-2, 2), -1.1501 + rng.uniform(-0.4, 0.4), 0.0, 20.0 * rng.uniform(0.9, 1.0), psi])
    Xr = np.zeros((fam.N + 1, fam.nx))
    Xr[0, :5] = x_init
    for k in range(1, fam.N + 1):
        Xr[k, :5] = [29.9948 + k * 20.0 * fam.dt * np.cos(psi), -1.1501 + k * 20.0 * fam.dt * np.sin(psi), 0.0, 20.0, psi]
    return np.concatenate([np.zeros(2 * fam.N), np.tile(Xr[0], fam.N + 1)]), np.concatenate([np.zeros(2 * fam.N), Xr.ravel()])


def instance(fam: Family, b: int):
    return ca_instance(fam, b) if fam.kind == "ca" else arc_instance(fam, b)


def batch(fam: Family, B: int, start: int = 0, indices=None):
    idx = range(start, start + B) if indices is None else indices
    xs, ps = zip(*(instance(fam, int(b)) for b in idx))
    return np.ascontiguousarray(np.stack(xs)), np.ascontiguousarray(np.stack(ps))


def bounds(fam: Family):
    """(lbx, ubx, lbg, ubg) of inequal_constraints() (optimizer.py:413-491) for the family"""
    o = _opt()
    r_ego, _ = o.compute_approximating_circle_radius(EGO_LENGTH, EGO_WIDTH)
    r_obs, _ = o.compute_approximating_circle_radius(fam.obstacle[2], fam.obstacle[3])
    N, nx = fam.N, fam.nx
    lbg = [0.0] + [0.0] * (nx * (N + 1)) + [r_ego + r_obs] * (9 * (N + 1))
    ubg = [11.5] + [0.0] * (nx * (N + 1)) + [np.inf] * (9 * (N + 1))
    lbx, ubx = [], []
    for _ in range(N):
        lbx += [-0.4, -np.inf]
        ubx += [0.4, 11.5]
    for _ in range(N + 1):
        lbx += [-np.inf, -np.inf, -1.066, 0.0, -np.inf] + ([-np.inf] if nx == 6 else [])
        ubx += [np.inf, np.inf, 1.066, 50.8, np.inf] + ([np.inf] if nx == 6 else [])
    return np.array(lbx), np.array(ubx), np.array(lbg), np.array(ubg)


def make_solver(fam: Family, **kw):
    """the product solver (one handle) for a family, bounds installed"""
    o = _opt()
    pkg = importlib.import_module("motion-planning-for-autonomous-driving-with-mpc_amd")
    centers = np.array(o.compute_centers_of_approximation_circles(*fam.obstacle), dtype=np.float64)
    _, dd = o.compute_approximating_circle_radius(EGO_LENGTH, EGO_WIDTH)
    s = pkg.BatchedMPCSolver(fam.N, fam.nx, dt=fam.dt, Q=fam.Qdiag, R=fam.R, obstacle_centers=centers, ego_offset=(dd / 2) / 2, **kw)
    s.set_bounds(*bounds(fam))
    return s


# ------------------------------------------------------------------------------------------------------------------------
# configuration 5: the mixed scenario sweep.  Global row g of 32 768 belongs to family MIXED_ORDER[g % 4] and is instance g of
# that family's generator; shard r (one per GPU) is the contiguous block of rows [4096 r, 4096 (r + 1)).
# ------------------------------------------------------------------------------------------------------------------------
def mixed_family_of(g):
    return np.asarray(g) % len(MIXED_ORDER)


def mixed_shard_rows(rank: int, world: int, total: int = MIXED_TOTAL):
    """global rows of shard `rank`, split by family: {family name: array of global rows}"""
    per = total // world
    lo, hi = rank * per, (rank + 1) * per if rank < world - 1 else total
    g = np.arange(lo, hi)
    f = mixed_family_of(g)
    return {name: g[f == i] for i, name in enumerate(MIXED_ORDER)}


def mixed_shard(rank: int, world: int, total: int = MIXED_TOTAL):
    """{family name: (global rows, x0, p)} of one shard"""
    return {name: (rows,) + batch(FAMILIES[name], 0, indices=rows) for name, rows in mixed_shard_rows(rank, world, total).items()}


MIXED_ROW_WIDTH = max(FAMILIES[n].n_w for n in MIXED_ORDER)       # result rows of different families are padded to one width for the final gather
