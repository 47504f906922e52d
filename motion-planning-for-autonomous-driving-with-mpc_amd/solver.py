"""
BatchedMPCSolver -- thin Python owner of an `mpc_handle` (include/mpcgpu.h).

One handle = one problem template (horizon, weights, obstacle, bounds) + its device workspace; it solves B
independent instances of the NLP of MPC_Planner/optimizer.py:373-558 per call, i.e. B times the reference's
`sol(x0=..., p=..., lbg=..., lbx=..., ubg=..., ubx=...)` (optimizer.py:607).

All arithmetic happens in the HIP kernels of csrc/mpcgpu.hip.  No CPU fallback exists: constructing a solver
without the built library or without a GPU raises.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _abi
from ._abi import MpcLibraryError, MpcProblemDesc


RESCUE_FRACTIONS = (0.0, 0.4, 0.7, 0.9, 1.0)      # of the circle-distance lower bound, see rescue_failed()


class MpcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"mpcgpu error {code}: {msg}")
        self.code = code


@dataclass
class SolveResult:
    x: np.ndarray          # [B, n_w] rows in the reference's decision-vector order (optimizer.py:550)
    status: np.ndarray     # [B] int32, 1 converged / 0 max-iter / -6 NaN / -7 no progress
    iters: np.ndarray      # [B] int32
    kkt: np.ndarray        # [B] scaled KKT error at exit


class BatchedMPCSolver:
    def __init__(self, N, nx=5, *, dt=0.1, Q=None, R=None, P=None, obstacle_centers=None, wheelbase=2.5789128,
                 friction_div=2.578, ego_offset=0.75, max_iter=100, tol=1e-8, fixed_iters=0, obst_mult=3, device=0,
                 lib_path=None):
        self._lib = _abi.load_library(lib_path)
        d = MpcProblemDesc()
        self._lib.mpc_default_desc(C.byref(d), int(N), int(nx))
        d.dt, d.wheelbase, d.friction_div, d.ego_offset = float(dt), float(wheelbase), float(friction_div), float(ego_offset)
        d.max_iter, d.fixed_iters, d.obst_mult, d.device, d.tol = int(max_iter), int(fixed_iters), int(obst_mult), int(device), float(tol)
        if Q is not None:
            for i in range(8):
                d.Q[i] = float(Q[i]) if i < len(Q) else 0.0
        if R is not None:
            d.R[0], d.R[1] = float(R[0]), float(R[1])
        if P is not None:
            for i in range(8):
                d.P[i] = float(P[i]) if i < len(P) else 0.0
        if obstacle_centers is not None:
            oc = np.asarray(obstacle_centers, dtype=np.float64).ravel()
            assert oc.size == 6
            for i in range(6):
                d.obstacle[i] = oc[i]
        self.desc = d
        self.N, self.nx = int(N), int(nx)
        self.n_w = 2 * self.N + self.nx * (self.N + 1)
        self.n_g = 1 + self.nx * (self.N + 1) + 9 * (self.N + 1)
        self._h = C.c_void_p()
        rc = self._lib.mpc_create(C.byref(self._h), C.byref(d))
        if rc != _abi.MPC_OK:
            raise MpcError(rc, self._lib.mpc_last_error(None).decode())
        self._bounds_key = None

    # ------------------------------------------------------------------------------------------------
    def _check(self, rc):
        if rc != _abi.MPC_OK:
            raise MpcError(rc, self._lib.mpc_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.mpc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------------
    def set_bounds(self, lbx=None, ubx=None, lbg=None, ubg=None):
        """the four lists of `inequal_constraints()` (optimizer.py:413-491); all None = reference defaults."""
        if lbx is None and ubx is None and lbg is None and ubg is None:
            self._check(self._lib.mpc_set_bounds(self._h, None, None, None, None))
            self._bounds_key = None
            self._bounds = None
            return
        arrs = [_abi.f64(a).ravel() for a in (lbx, ubx, lbg, ubg)]
        if arrs[0].size != self.n_w or arrs[1].size != self.n_w or arrs[2].size != self.n_g or arrs[3].size != self.n_g:
            raise MpcError(_abi.MPC_ERR_BOUNDS, f"expected lbx/ubx of {self.n_w} and lbg/ubg of {self.n_g} entries")
        key = tuple(a.tobytes() for a in arrs)
        if key == self._bounds_key:
            return
        self._check(self._lib.mpc_set_bounds(self._h, *[_abi.as_dp(a) for a in arrs]))
        self._bounds_key = key
        self._bounds = tuple(a.copy() for a in arrs)            # (lbx, ubx, lbg, ubg), for rescue_failed()

    def solve(self, x0, p, obst=None) -> SolveResult:
        x0 = _abi.f64(x0)
        p = _abi.f64(p)
        if x0.ndim == 1:
            x0 = x0[None]
        if p.ndim == 1:
            p = p[None]
        B = x0.shape[0]
        if x0.shape != (B, self.n_w) or p.shape != (B, self.n_w):
            raise MpcError(_abi.MPC_ERR_INVALID, f"x0/p must be [B, {self.n_w}]")
        if obst is not None:
            obst = _abi.f64(obst, (B, 6))
        out = np.empty_like(x0)
        status = np.empty(B, np.int32)
        iters = np.empty(B, np.int32)
        kkt = np.empty(B, np.float64)
        self._check(self._lib.mpc_solve_batch(self._h, B, _abi.as_dp(x0), _abi.as_dp(p), _abi.as_dp(obst), _abi.as_dp(out),
                                              _abi.as_ip(status), _abi.as_ip(iters), _abi.as_dp(kkt)))
        return SolveResult(out, status, iters, kkt)

    def last_rescued(self):
        """instances of the last solve that took the second chance of the C-ABI (mpc_last_rescued)"""
        return int(self._lib.mpc_last_rescued(self._h))

    def solve_with_rescue(self, x0, p):
        """diagnostic form of solve(): the second chance for stalled instances (homotopy on the obstacle radius) runs on the
        device behind the C-ABI in every solve; this returns the result together with the mask of the instances that needed
        it, found by solving once with the option "rescue" off.  Returns (SolveResult, rescued mask)."""
        before = self.get_option("rescue")
        self.set_option("rescue", "0")
        try:
            plain = self.solve(x0, p)
        finally:
            self.set_option("rescue", before)          # what it was (the environment or the caller may have switched it off)
        if np.all(plain.status == 1):
            return plain, np.zeros(plain.status.shape[0], dtype=bool)
        res = self.solve(x0, p)
        return res, (plain.status != 1) & (res.status == 1)

    def solve_trace(self, x0, p, obst=None):
        x0 = _abi.f64(x0)
        p = _abi.f64(p)
        B = x0.shape[0]
        out = np.empty_like(x0)
        status, iters, kkt = np.empty(B, np.int32), np.empty(B, np.int32), np.empty(B)
        rows = int(self.desc.max_iter) + 1
        trace = np.zeros((rows, 8, B))
        n_it = np.zeros(1, np.int32)
        if obst is not None:
            obst = _abi.f64(obst, (B, 6))
        self._check(self._lib.mpc_solve_batch_trace(self._h, B, _abi.as_dp(x0), _abi.as_dp(p), _abi.as_dp(obst), _abi.as_dp(out),
                                                    _abi.as_ip(status), _abi.as_ip(iters), _abi.as_dp(kkt), _abi.as_dp(trace), rows,
                                                    _abi.as_ip(n_it)))
        return SolveResult(out, status, iters, kkt), trace[: int(n_it[0]) + 1]

    def solve_device(self, B, d_x0, d_p, d_x_out, d_status=0, d_iters=0, d_kkt=0, d_obst=0, stream=0):
        """device pointers (ints, e.g. torch.Tensor.data_ptr()) and a hipStream_t handle (int, 0 = default)."""
        vp = C.c_void_p
        self._check(self._lib.mpc_solve_batch_dev(self._h, int(B), vp(d_x0), vp(d_p), vp(d_obst or None), vp(d_x_out),
                                                  vp(d_status or None), vp(d_iters or None), vp(d_kkt or None), vp(stream or None)))

    def plant_step(self, x, u, integrator="euler"):
        x = _abi.f64(x)
        u = _abi.f64(u)
        single = x.ndim == 1
        x2 = x.reshape(-1, self.nx)
        u2 = u.reshape(-1, 2)
        out = np.empty_like(x2)
        self._check(self._lib.mpc_plant_step(self._h, x2.shape[0], 0 if integrator == "euler" else 1, _abi.as_dp(x2), _abi.as_dp(u2),
                                             _abi.as_dp(out)))
        return out[0] if single else out

    def closed_loop(self, init_state, path, orient, vdes, steps, noise_mode=0, sigma=0.0, seed=0):
        """B egos through `steps` receding-horizon steps on the device (include/mpcgpu.h: mpc_closed_loop_batch_ex; the loop
        body of CasadiOptimizer.optimize, optimizer.py:596-631).  init_state [B,5], path [B,Lp,2], orient [B,Lp],
        vdes [B] -> (traj [B,steps,5], ctrl [B,steps,2], step_status [B,steps]).  noise_mode / sigma / seed: the reference's
        `noised: True` with a counter-based generator (noise.py holds the Python mirror of the samples)."""
        init_state = _abi.f64(init_state)
        if init_state.ndim == 1:
            init_state = init_state[None]
        B = init_state.shape[0]
        path = _abi.f64(path).reshape(B, -1, 2)
        Lp = path.shape[1]
        orient = _abi.f64(orient).reshape(B, Lp)
        vdes = _abi.f64(np.broadcast_to(np.asarray(vdes, dtype=np.float64), (B,)))
        steps = int(steps)
        traj = np.empty((B, steps, 5))
        ctrl = np.empty((B, steps, 2))
        st = np.empty((B, steps), np.int32)
        self._check(self._lib.mpc_closed_loop_batch_ex(self._h, B, steps, Lp, _abi.as_dp(init_state), _abi.as_dp(path), _abi.as_dp(orient),
                                                       _abi.as_dp(vdes), int(noise_mode), float(sigma), int(seed) & (2 ** 64 - 1), _abi.as_dp(traj),
                                                       _abi.as_dp(ctrl), _abi.as_ip(st)))
        return traj, ctrl, st

    def closed_loop_device(self, B, d_init_state, d_path, d_orient, d_vdes, steps, Lp, d_traj, d_ctrl, d_step_status=0, noise_mode=0, sigma=0.0,
                           seed=0, stream=0):
        """device-pointer form (ints): the whole loop is enqueued on `stream`; see mpc_closed_loop_batch_dev_ex"""
        vp = C.c_void_p
        self._check(self._lib.mpc_closed_loop_batch_dev_ex(self._h, int(B), int(steps), int(Lp), vp(d_init_state), vp(d_path), vp(d_orient), vp(d_vdes),
                                                           int(noise_mode), float(sigma), int(seed) & (2 ** 64 - 1), vp(d_traj), vp(d_ctrl),
                                                           vp(d_step_status or None), vp(stream or None)))

    def last_loop_replayed(self):
        return bool(self._lib.mpc_last_loop_replayed(self._h))

    def metrics(self, traj, ref_path=None, origin_path=None, r_sum=None, all_pairs=False):
        """Post-hoc metrics of B planned trajectories [B,L,5] on the device (mpc_metrics_batch; mpc_planner.py:184-199,
        279-292): dict(deviation [B,L] | None, rmsd [B,2] | None, clearance [B] | None)."""
        traj = _abi.f64(traj)
        if traj.ndim == 2:
            traj = traj[None]
        B, L = traj.shape[0], traj.shape[1]
        dev = rm = cl = None
        Lo = 0
        if origin_path is not None:
            origin_path = _abi.f64(np.broadcast_to(np.asarray(origin_path, dtype=np.float64), (B,) + np.asarray(origin_path).shape[-2:]))
            Lo = origin_path.shape[1]
            dev = np.empty((B, L))
        if ref_path is not None:
            ref_path = _abi.f64(np.broadcast_to(np.asarray(ref_path, dtype=np.float64)[..., :L, :], (B, L, 2)))
            rm = np.empty((B, 2))
        if r_sum is not None:
            cl = np.empty(B)
        self._check(self._lib.mpc_metrics_batch(self._h, B, L, Lo, _abi.as_dp(traj), _abi.as_dp(ref_path), _abi.as_dp(origin_path),
                                                C.c_double(0.0 if r_sum is None else float(r_sum)), 1 if all_pairs else 0, _abi.as_dp(dev), _abi.as_dp(rm),
                                                _abi.as_dp(cl)))
        return dict(deviation=dev, rmsd=rm, clearance=cl)

    def validity(self, traj, obstacles=None, left=None, right=None, ego_length=4.3, ego_width=1.8):
        """collision / road verdict of B planned trajectories [B,L,5] on the device (mpc_validity_batch; the check of the
        reference's test, test/test_mpc_planner.py:37-47).  obstacles: [n,5] static rectangles (x, y, length, width, orientation)
        or [n,L,5] per time step; left / right: boundary polylines [m,2] of the drivable corridor in driving direction.
        Returns dict(first_collision [B], first_off_road [B]) -- step index or -1."""
        traj = _abi.f64(traj)
        if traj.ndim == 2:
            traj = traj[None]
        B, L = traj.shape[0], traj.shape[1]
        n_obst, ob = 0, None
        if obstacles is not None and len(obstacles):
            ob = np.asarray(obstacles, dtype=np.float64)
            if ob.ndim == 2:
                ob = np.repeat(ob[:, None, :], L, axis=1)
            ob = _abi.f64(ob[:, :L])
            n_obst = ob.shape[0]
        lf = None if left is None else _abi.f64(left).reshape(-1, 2)
        rt = None if right is None else _abi.f64(right).reshape(-1, 2)
        fc, fo = np.empty(B, np.int32), np.empty(B, np.int32)
        self._check(self._lib.mpc_validity_batch(self._h, B, L, _abi.as_dp(traj), float(ego_length), float(ego_width), n_obst, _abi.as_dp(ob),
                                                 0 if lf is None else lf.shape[0], _abi.as_dp(lf), 0 if rt is None else rt.shape[0], _abi.as_dp(rt),
                                                 _abi.as_ip(fc), _abi.as_ip(fo)))
        return dict(first_collision=fc, first_off_road=fo)

    def forces_stage_eval(self, z, p, terminal=False):
        """FORCES-mode stage functions of B (z, p) pairs on the device (mpc_forces_stage_eval; the generated
        FORCESNLPsolver_model.c of the reference): dict(f, grad_f, c, jac_c, h, jac_h); c / jac_c are None at the last stage."""
        z = _abi.f64(z).reshape(-1, 7)
        p = _abi.f64(p).reshape(-1, 10)
        B = z.shape[0]
        f, gf = np.empty(B), np.empty((B, 7))
        c, jc = (None, None) if terminal else (np.empty((B, 5)), np.empty((B, 5, 7)))
        hv, jh = np.empty((B, 10)), np.empty((B, 10, 7))
        self._check(self._lib.mpc_forces_stage_eval(self._h, B, 1 if terminal else 0, _abi.as_dp(z), _abi.as_dp(p), _abi.as_dp(f),
                                                    _abi.as_dp(gf), _abi.as_dp(c), _abi.as_dp(jc), _abi.as_dp(hv), _abi.as_dp(jh)))
        return dict(f=f, grad_f=gf, c=c, jac_c=jc, h=hv, jac_h=jh)

    def forces_solve(self, x0, xinit, all_parameters, lb, ub, hl, hu, hessian_mode=0):
        """FORCES-mode SQP step for B problems (mpc_forces_solve_batch): x0 [B,N,7], xinit [B,5], all_parameters [B,N,10]
        -> (x [B,N,7], exitflag [B], it [B], res [B])."""
        N = self.N
        x0 = _abi.f64(x0).reshape(-1, N, 7)
        B = x0.shape[0]
        xinit = _abi.f64(xinit).reshape(B, 5)
        par = _abi.f64(all_parameters).reshape(B, N, 10)

        def big(a, n):
            a = np.asarray(a, dtype=np.float64).reshape(n)
            return _abi.f64(np.where(np.isfinite(a), a, np.sign(a) * 1e308))
        out = np.empty_like(x0)
        flag, it, res = np.empty(B, np.int32), np.empty(B, np.int32), np.empty(B)
        self._check(self._lib.mpc_forces_solve_batch(self._h, B, _abi.as_dp(x0), _abi.as_dp(xinit), _abi.as_dp(par), _abi.as_dp(big(lb, 7)),
                                                     _abi.as_dp(big(ub, 7)), _abi.as_dp(big(hl, 10)), _abi.as_dp(big(hu, 10)), int(hessian_mode), _abi.as_dp(out),
                                                     _abi.as_ip(flag), _abi.as_ip(it), _abi.as_dp(res)))
        return out, flag, it, res

    def forces_solve_device(self, B, d_x0, d_xinit, d_par, lb, ub, hl, hu, d_x_out, d_flag=0, d_it=0, d_res=0, hessian_mode=0, stream=0):
        """mpc_forces_solve_batch_dev: device pointers (ints) for x0 [B,N,7], xinit [B,5], all_parameters [B,N,10] and the outputs;
        lb / ub / hl / hu are small host arrays.  Enqueues on `stream`; nothing is synchronised."""
        def big(a, n):
            a = np.asarray(a, dtype=np.float64).reshape(n)
            return _abi.f64(np.where(np.isfinite(a), a, np.sign(a) * 1e308))
        vp = C.c_void_p
        self._check(self._lib.mpc_forces_solve_batch_dev(self._h, int(B), vp(d_x0), vp(d_xinit), vp(d_par), _abi.as_dp(big(lb, 7)), _abi.as_dp(big(ub, 7)),
                                                         _abi.as_dp(big(hl, 10)), _abi.as_dp(big(hu, 10)), int(hessian_mode), vp(d_x_out), vp(d_flag or None),
                                                         vp(d_it or None), vp(d_res or None), vp(stream or None)))

    def forces_closed_loop(self, init_state, path, orient, vdes, steps, lb, ub, hl, hu, init_acc=None, hessian_mode=0, noise_mode=0, sigma=0.0, seed=0):
        """B egos through `steps` steps of ForcesproOptimizer.optimize (optimizer.py:246-366) on the device
        (mpc_forces_closed_loop_batch): init_state [B,5], path [B,Lp,2], orient [B,Lp], vdes [B] -> (traj [B,steps,5],
        ctrl [B,steps,2], exitflag [B,steps])."""
        def big(a, n):
            a = np.asarray(a, dtype=np.float64).reshape(n)
            return _abi.f64(np.where(np.isfinite(a), a, np.sign(a) * 1e308))
        init_state = _abi.f64(init_state)
        if init_state.ndim == 1:
            init_state = init_state[None]
        B = init_state.shape[0]
        path = _abi.f64(path).reshape(B, -1, 2)
        Lp = path.shape[1]
        orient = _abi.f64(orient).reshape(B, Lp)
        vdes = _abi.f64(np.broadcast_to(np.asarray(vdes, dtype=np.float64), (B,)))
        acc = None if init_acc is None else _abi.f64(np.broadcast_to(np.asarray(init_acc, dtype=np.float64), (B,)))
        steps = int(steps)
        traj, ctrl, fl = np.empty((B, steps, 5)), np.empty((B, steps, 2)), np.empty((B, steps), np.int32)
        self._check(self._lib.mpc_forces_closed_loop_batch(self._h, B, steps, Lp, _abi.as_dp(init_state), _abi.as_dp(acc), _abi.as_dp(path), _abi.as_dp(orient),
                                                           _abi.as_dp(vdes), _abi.as_dp(big(lb, 7)), _abi.as_dp(big(ub, 7)), _abi.as_dp(big(hl, 10)),
                                                           _abi.as_dp(big(hu, 10)), int(hessian_mode), int(noise_mode), float(sigma), int(seed) & (2 ** 64 - 1),
                                                           _abi.as_dp(traj), _abi.as_dp(ctrl), _abi.as_ip(fl)))
        return traj, ctrl, fl

    def set_option(self, name, value=None):
        """run-time switch of the handle (include/mpcgpu.h: mpc_set_option); value None restores the default.  The
        environment (MPCGPU_<NAME>) is only read when the handle is created."""
        v = None if value is None else str(value).encode()
        self._check(self._lib.mpc_set_option(self._h, str(name).encode(), v))

    def get_option(self, name):
        """current value of a run-time switch (include/mpcgpu.h: mpc_get_option)"""
        import ctypes
        v = ctypes.c_int64(0)
        self._check(self._lib.mpc_get_option(self._h, str(name).encode(), ctypes.byref(v)))
        return int(v.value)

    def set_profiling(self, enable=True):
        """True / 1: HIP events around every kernel; 2: the iteration loop of a hybrid solve as one span (no marker between its two kernels)."""
        self._check(self._lib.mpc_set_profiling(self._h, int(enable) if enable in (0, 1, 2) else (1 if enable else 0)))

    def get_profile(self):
        out = np.zeros(6)
        self._check(self._lib.mpc_get_profile(self._h, _abi.as_dp(out)))
        return dict(riccati_ms=out[0], riccati_launches=int(out[1]), stage_ms=out[2], stage_launches=int(out[3]),
                    other_ms=out[4], iterations=int(out[5]))

    def get_resident_profile(self):
        """Figures of the workgroup-resident kernel of the last solve (k_solve_wg: finishes the instances of the tiles that left the
        pipeline -- the hybrid solve -- or solves a small batch alone); `ran` is False when only streaming paths ran.
        `instance_iterations`: interior-point iterations it performed (the others ran in the pipeline)."""
        out = np.zeros(8)
        self._check(self._lib.mpc_get_resident_profile(self._h, _abi.as_dp(out)))
        return dict(ms=out[0], ran=bool(out[1]), rounds=int(out[2]), workgroups=int(out[3]), workgroup_rounds=int(out[4]), sweeps=int(out[5]),
                    instance_iterations=int(out[6]))

    def measure_copy_bandwidth(self, nbytes=1 << 30, reps=5):
        """GB/s (read + write) of the library's own streaming copy kernel on this device (roofline denominator of bench.py)"""
        out = np.zeros(1)
        self._check(self._lib.mpc_measure_copy_bandwidth(self._h, nbytes, reps, _abi.as_dp(out)))
        return float(out[0])

    def get_pipeline_profile(self):
        """Figures of the single-launch pipeline (k_pipeline) for the last solve; `ran` is False when the solve used one
        launch per kernel instead (small or very large batches, MPCGPU_PIPELINE=0)."""
        out = np.zeros(8)
        self._check(self._lib.mpc_get_pipeline_profile(self._h, _abi.as_dp(out)))
        return dict(ms=out[0], ran=bool(out[1]), rounds=int(out[2]), riccati_wait_ms=out[3], stage_wait_ms=out[4],
                    stage_busy_ms=out[5], items=int(out[6]), stage_workers=int(out[7]), riccati_workers=int(round((out[7] % 1) * 1000)))


def rescue_failed(backend, x0, p, result, bounds, fractions=RESCUE_FRACTIONS):
    """Second chance for the instances of a batch that did not converge (status != 1), by homotopy on the obstacle radius.

    IPOPT hands a start that is locally infeasible -- typically a guess that runs straight through the obstacle, where
    the linearised circle constraints cannot be met within the fraction-to-the-boundary rule -- to its restoration phase,
    which is not restated here.  Instead the failed instances are re-solved on the device with the
    lower bound of the circle-distance rows (`lbg` of the 9 (N+1) obstacle rows, optimizer.py:426-428) raised in steps
    from 0 to its true value, each solve warm-started from the previous solution; the last solve is the ORIGINAL NLP, so
    what comes back is a KKT point of the original problem to the original tolerance, or the original failure.

    backend: anything with set_bounds(lbx, ubx, lbg, ubg) and solve(x0, p) -> SolveResult (the BatchedMPCSolver).
    bounds: the (lbx, ubx, lbg, ubg) of the original problem.  Returns (result, rescued mask)."""
    status = np.asarray(result.status)
    bad = np.nonzero(status != 1)[0]
    rescued = np.zeros(status.shape[0], dtype=bool)
    if bad.size == 0:
        return result, rescued
    lbx, ubx, lbg, ubg = [np.asarray(a, dtype=np.float64).ravel().copy() for a in bounds]
    n_obst = 9 * (int(backend.N) + 1) if hasattr(backend, "N") else 9 * ((lbg.size - 1) // 14)
    xs = np.asarray(x0, dtype=np.float64)[bad].copy()
    ps = np.asarray(p, dtype=np.float64)[bad]
    iters = np.zeros(bad.size, dtype=np.int64)
    last = None
    try:
        for frac in fractions:
            lbg_f = lbg.copy()
            lbg_f[-n_obst:] = frac * lbg[-n_obst:]
            backend.set_bounds(lbx, ubx, lbg_f, ubg)
            last = backend.solve(xs, ps)
            ok = np.asarray(last.status) == 1
            xs[ok] = np.asarray(last.x)[ok]                       # warm start of the next, tighter problem
            iters += np.asarray(last.iters)
    finally:
        backend.set_bounds(lbx, ubx, lbg, ubg)
    ok = np.asarray(last.status) == 1
    x, st, it, kkt = [np.array(a, copy=True) for a in (result.x, result.status, result.iters, result.kkt)]
    sel = bad[ok]
    x[sel], st[sel], kkt[sel] = np.asarray(last.x)[ok], 1, np.asarray(last.kkt)[ok]
    it[sel] = it[sel] + iters[ok]
    rescued[sel] = True
    return SolveResult(x, st, it, kkt), rescued


__all__ = ["BatchedMPCSolver", "SolveResult", "MpcError", "MpcLibraryError", "rescue_failed", "RESCUE_FRACTIONS"]
