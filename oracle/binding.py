"""
ORACLE (test infrastructure, not product code) -- ctypes binding of oracle/libmpc_oracle.so and, when present,
of oracle/_ref/libforces_model_ref.so (the reference's own CasADi-generated stage functions compiled by
oracle/Makefile from /root/reference/test/FORCESNLPsolver/*.c).

Importers: tests/, bench.py (cpu_baseline leg), __graft_entry__.smoke(), tests/golden/make_golden.py.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from .nlp_numpy import NLPConfig

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmpc_oracle.so")
REF_PATH = os.path.join(HERE, "_ref", "libforces_model_ref.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class MpcoDesc(C.Structure):
    _fields_ = [
        ("N", C.c_int32), ("nx", C.c_int32), ("obst_mult", C.c_int32), ("max_iter", C.c_int32),
        ("fixed_iters", C.c_int32), ("reserved", C.c_int32),
        ("dt", C.c_double), ("wheelbase", C.c_double), ("friction_div", C.c_double), ("ego_offset", C.c_double),
        ("Q", C.c_double * 6), ("R", C.c_double * 2), ("obst", C.c_double * 6),
        ("fric_lo", C.c_double), ("fric_hi", C.c_double), ("obst_lo", C.c_double), ("obst_hi", C.c_double),
        ("tol", C.c_double),
    ]


def build(force: bool = False):
    """compile the oracle (and oracle/_ref when /root/reference exists) -- building the checker is not using it."""
    src = os.path.join(HERE, "mpc_oracle.c")
    stale = (not os.path.exists(LIB_PATH)) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src)
    if force or stale or (os.path.isdir("/root/reference") and not os.path.exists(REF_PATH)):
        subprocess.check_call(["make", "-s", "-C", HERE, "all"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.mpco_solve.argtypes = [C.POINTER(MpcoDesc), _dp, _dp, _dp, _dp, _dp, _ip, _ip, _dp, _dp]
        L.mpco_solve.restype = C.c_int
        L.mpco_solve_trace.argtypes = [C.POINTER(MpcoDesc), _dp, _dp, _dp, _dp, _dp, _ip, _ip, _dp, _dp, _dp, C.c_int32]
        L.mpco_solve_trace.restype = C.c_int
        L.mpco_solve_batch.argtypes = [C.POINTER(MpcoDesc), _dp, _dp, C.c_int32, _dp, _dp, _dp, _ip, _ip, _dp, C.c_int32]
        L.mpco_solve_batch.restype = C.c_int
        L.mpco_ode.argtypes = [C.POINTER(MpcoDesc), _dp, _dp, _dp]
        L.mpco_plant_step_euler.argtypes = [C.POINTER(MpcoDesc), _dp, _dp, _dp]
        L.mpco_plant_step_rk4.argtypes = [C.POINTER(MpcoDesc), _dp, _dp, _dp]
        L.mpco_objective.argtypes = [C.POINTER(MpcoDesc), _dp, _dp]
        L.mpco_objective.restype = C.c_double
        L.mpco_constraints.argtypes = [C.POINTER(MpcoDesc), _dp, _dp, _dp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(_dp)


def _pi(a):
    return a.ctypes.data_as(_ip)


def make_desc(cfg: NLPConfig, max_iter=100, tol=1e-8, fixed_iters=0, obst_mult=3, literal_friction_row=False) -> MpcoDesc:
    # literal_friction_row: True / 1 = the stage-0 friction row stays a row (no presolve); 2 or "ipopt" = as IPOPT sees it, the row's lower
    # bound lbg[0] = 0 with its log barrier too (the product's option friction_lb = ipopt)
    d = MpcoDesc()
    d.N, d.nx, d.obst_mult, d.max_iter, d.fixed_iters = cfg.N, cfg.nx, obst_mult, max_iter, fixed_iters
    d.reserved = 2 if literal_friction_row in (2, "ipopt") else (1 if literal_friction_row else 0)
    d.dt, d.wheelbase, d.friction_div, d.ego_offset = cfg.dt, cfg.wheelbase, cfg.friction_div, cfg.ego_offset
    for i, q in enumerate(cfg.Qdiag):
        d.Q[i] = q
    d.R[0], d.R[1] = cfg.R
    oc = cfg.obstacle_centers.ravel()
    for i in range(6):
        d.obst[i] = oc[i]
    d.fric_lo, d.fric_hi = 0.0, cfg.a_max
    d.obst_lo, d.obst_hi = cfg.r_sum, np.inf
    d.tol = tol
    return d


class OracleSolver:
    """`sol(x0, p, lbx=..., ubx=...)`-style front end of the C oracle for one problem template."""

    def __init__(self, cfg: NLPConfig, **kw):
        from .nlp_numpy import BicycleNLP
        self.cfg = cfg
        self.desc = make_desc(cfg, **kw)
        _, _, lbx, ubx = BicycleNLP(cfg).bounds()
        self.lbx = np.ascontiguousarray(lbx, dtype=np.float64)
        self.ubx = np.ascontiguousarray(ubx, dtype=np.float64)

    def solve(self, x0, p, trace=False):
        x0 = np.ascontiguousarray(x0, dtype=np.float64).ravel()
        p = np.ascontiguousarray(p, dtype=np.float64).ravel()
        out = np.zeros(self.cfg.n_w)
        st, it = np.zeros(1, np.int32), np.zeros(1, np.int32)
        kkt, obj = np.zeros(1), np.zeros(1)
        if trace:
            tr = np.zeros((self.desc.max_iter + 1, 8))
            rc = lib().mpco_solve_trace(C.byref(self.desc), _p(self.lbx), _p(self.ubx), _p(x0), _p(p), _p(out),
                                        _pi(st), _pi(it), _p(kkt), _p(obj), _p(tr), tr.shape[0])
        else:
            tr = None
            rc = lib().mpco_solve(C.byref(self.desc), _p(self.lbx), _p(self.ubx), _p(x0), _p(p), _p(out),
                                  _pi(st), _pi(it), _p(kkt), _p(obj))
        assert rc == 0, rc
        res = dict(x=out, status=int(st[0]), iters=int(it[0]), kkt=float(kkt[0]), f=float(obj[0]))
        if trace:
            res["trace"] = tr[: int(it[0]) + 1]
        return res

    def solve_batch(self, x0, p, nthreads=1, out=None):
        """out: the dict of an earlier call of the same shape -- its arrays are written again (a timing loop then measures the solves, not the
        page faults of four fresh arrays)"""
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        p = np.ascontiguousarray(p, dtype=np.float64)
        B = x0.shape[0]
        if out is not None:
            out, st, it, kkt = out["x"], out["status"], out["iters"], out["kkt"]
            assert out.shape == x0.shape and st.shape == (B,)
        else:
            out = np.zeros_like(x0)
            st, it, kkt = np.zeros(B, np.int32), np.zeros(B, np.int32), np.zeros(B)
        rc = lib().mpco_solve_batch(C.byref(self.desc), _p(self.lbx), _p(self.ubx), B, _p(x0), _p(p), _p(out),
                                    _pi(st), _pi(it), _p(kkt), int(nthreads))
        assert rc == 0, rc
        return dict(x=out, status=st, iters=it, kkt=kkt)

    # -- model pieces
    def ode(self, x, u):
        x = np.ascontiguousarray(x, dtype=np.float64)
        u = np.ascontiguousarray(u, dtype=np.float64)
        f = np.zeros(self.cfg.nx)
        lib().mpco_ode(C.byref(self.desc), _p(x), _p(u), _p(f))
        return f

    def plant_step(self, x, u, integrator="euler"):
        x = np.ascontiguousarray(x, dtype=np.float64)
        u = np.ascontiguousarray(u, dtype=np.float64)
        xn = np.zeros(self.cfg.nx)
        fn = lib().mpco_plant_step_euler if integrator == "euler" else lib().mpco_plant_step_rk4
        fn(C.byref(self.desc), _p(x), _p(u), _p(xn))
        return xn

    def objective(self, w, p):
        w = np.ascontiguousarray(w, dtype=np.float64)
        p = np.ascontiguousarray(p, dtype=np.float64)
        return lib().mpco_objective(C.byref(self.desc), _p(w), _p(p))

    def constraints(self, w, p):
        w = np.ascontiguousarray(w, dtype=np.float64)
        p = np.ascontiguousarray(p, dtype=np.float64)
        g = np.zeros(self.cfg.n_g)
        lib().mpco_constraints(C.byref(self.desc), _p(w), _p(p), _p(g))
        return g


# ----------------------------------------------------------------------------------------------------------
# oracle/_ref : the reference's CasADi-generated FORCES-mode stage functions (FORCESNLPsolver_model.c)
# ----------------------------------------------------------------------------------------------------------
class ForcesModelRef:
    """Calls `FORCESNLPsolver_casadi2forces` (FORCESNLPsolver_interface.c:41-198) of the compiled reference."""

    def __init__(self):
        if not os.path.exists(REF_PATH):
            raise FileNotFoundError(REF_PATH)
        self.lib = C.CDLL(REF_PATH)
        self.fn = self.lib.FORCESNLPsolver_casadi2forces
        self.fn.argtypes = [_dp] * 11 + [C.c_int32, C.c_int32, C.c_int32]
        self.fn.restype = None

    def eval(self, z, p, stage):
        """z (7,), p (10,) -> dict(f, grad_f(7), c(5), jac_c(5,7), h(10), jac_h(10,7)); terminal stage (9) has no c."""
        z = np.ascontiguousarray(z, dtype=np.float64)
        p = np.ascontiguousarray(p, dtype=np.float64)
        y, l = np.zeros(5), np.zeros(10)
        f = np.zeros(1)
        gf, c, jc, h, jh = np.zeros(7), np.zeros(5), np.zeros(35), np.zeros(10), np.zeros(70)
        self.fn(_p(z), _p(y), _p(l), _p(p), _p(f), _p(gf), _p(c), _p(jc), _p(h), _p(jh), None, stage, 0, 0)
        return dict(f=float(f[0]), grad_f=gf, c=c, jac_c=jc.reshape(7, 5).T.copy(), h=h,
                    jac_h=jh.reshape(7, 10).T.copy())
