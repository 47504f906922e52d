"""
Host-side mirror of the reference's `MPC_Planner/optimizer.py` call surface, backed by the HIP solver.

What is mirrored (same names, argument meaning, return shapes, and the behaviour-defining quirks of SURVEY.md
App. C), so that `mpc_planner.py:301-309` works unchanged against this module:

    CasadiOptimizer(configuration=, init_values=, predict_horizon=)        optimizer.py:369-371
        .inequal_constraints() -> lbg, ubg, lbx, ubx                        optimizer.py:413-491
        .solver() -> (sol, f)                                               optimizer.py:513-560
              sol(x0=, p=, lbg=, lbx=, ubg=, ubx=)['x'].full()              optimizer.py:607-609
              f(x, u).full()                                                optimizer.py:649-650
        .optimize() -> (states[L,5], controls[L,2], solve_time[L])          optimizer.py:562-643
        .shift_movement(...), .desired_command_and_trajectory(...)          optimizer.py:645-702
    ForcesproOptimizer(...)  -- call surface of `solver.solve(problem)`     optimizer.py:197-366

The reference rebuilds the CasADi graph and the IPOPT object on every MPC step (optimizer.py:605); here
`solver()` returns a cached handle to the device workspace.  All NLP arithmetic runs in csrc/mpcgpu.hip through
the C-ABI of include/mpcgpu.h; this file only packs arguments.  `ca` below is a tiny numeric stand-in for the
three CasADi functions that `mpc_planner.py` reaches through its star import (mpc_planner.py:288-290).
"""
from __future__ import annotations

import math
import time
import types

import numpy as np

from . import noise as _noise
from .solver import BatchedMPCSolver, rescue_failed

# ------------------------------------------------------------------------------------------------------------
# numeric stand-ins for names the reference's planner pulls in via `from MPC_Planner.optimizer import *`
# ------------------------------------------------------------------------------------------------------------
ca = types.SimpleNamespace(
    sqrt=np.sqrt,
    vertcat=lambda *a: np.concatenate([np.atleast_1d(np.asarray(x, dtype=np.float64)).ravel() for x in a]).reshape(-1, 1),
    reshape=lambda a, r, c: DMLike(np.asarray(a.full() if hasattr(a, "full") else a, dtype=np.float64).reshape((r, c), order="F")),
    cos=np.cos, sin=np.sin, tan=np.tan,
)


class DMLike:
    """the two things the reference does with a casadi.DM: `.full()` and reshaping."""

    def __init__(self, a):
        self._a = np.array(a, dtype=np.float64)

    def full(self):
        return self._a.copy()

    def __array__(self, dtype=None, copy=None):
        return self._a if dtype is None else self._a.astype(dtype)

    @property
    def shape(self):
        return self._a.shape


def find_closest_point(path_points, current_point):
    """configuration.py:26-37."""
    diff = np.transpose(np.transpose(path_points) - current_point.reshape(2, 1))
    squared_dist = np.power(diff, 2)[:, 0] + np.power(diff, 2)[:, 1]
    return np.argmin(squared_dist)


def compute_approximating_circle_radius(ego_length, ego_width):
    """configuration.py:40-66 (radius rounding rule included)."""
    assert ego_length >= 0 and ego_width >= 0, "Invalid vehicle dimensions = {}".format([ego_length, ego_width])
    if np.isclose(ego_length, 0.0) and np.isclose(ego_width, 0.0):
        return 0.0, 0.0
    square_length = ego_length / 3
    diagonal_square = np.sqrt((square_length / 2) ** 2 + (ego_width / 2) ** 2)
    if diagonal_square > round(diagonal_square, 1):
        approx_radius = round(diagonal_square, 1) + 0.1
    else:
        approx_radius = round(diagonal_square, 1)
    return approx_radius, round(square_length * 2, 1)


def compute_centers_of_approximation_circles(x_position, y_position, v_length, v_width, orientation):
    """configuration.py:69-93: centre, front and rear circle centres."""
    _, disc_distance = compute_approximating_circle_radius(v_length, v_width)
    distance_centers = disc_distance / 2
    center = [x_position, y_position]
    center_fw = [x_position + (distance_centers / 2) * np.cos(orientation), y_position + (distance_centers / 2) * np.sin(orientation)]
    center_rw = [x_position - (distance_centers / 2) * np.cos(orientation), y_position - (distance_centers / 2) * np.sin(orientation)]
    return center, center_fw, center_rw


class VehicleDynamics:
    """configuration.py:342-368 (kinematic single track); l = a + b of parameters_vehicle2."""
    l = 2.5789128

    @classmethod
    def KS_casadi(cls, x, u):
        x = np.asarray(x, dtype=np.float64).ravel()
        u = np.asarray(u, dtype=np.float64).ravel()
        return np.array([x[3] * math.cos(x[4]), x[3] * math.sin(x[4]), u[0], u[1], x[3] / cls.l * math.tan(x[2])])


class _DynamicsFunction:
    """`f = ca.Function('f', [states, controls], [rhs])` look-alike (optimizer.py:537): f(x, u).full() -> (5,1)."""

    def __call__(self, x, u):
        return DMLike(VehicleDynamics.KS_casadi(x, u).reshape(-1, 1))


class NlpSolution(dict):
    """what `sol(...)` returns: res['x'].full() is the optimal decision vector (optimizer.py:609)."""


class NlpSolverHandle:
    """`ca.nlpsol('solver', 'ipopt', nlp_prob, opts)` look-alike (optimizer.py:558): callable with the same
    keyword arguments as optimizer.py:607.  Accepts a single instance ((n_w,1) / (n_w,)) or a batch [B, n_w]."""

    # failed instances get a second chance by homotopy on the obstacle radius: on the device, behind the C-ABI
    # (mpc_solve_batch; include/mpcgpu.h).  A backend without that (the stand-in backends of the tests) gets the same
    # procedure from solver.rescue_failed.
    rescue = True

    def __init__(self, backend: BatchedMPCSolver):
        self._backend = backend
        self._stats = {}

    def __call__(self, x0=None, p=None, lbg=None, lbx=None, ubg=None, ubx=None, **_unused):
        be = self._backend
        x0a = np.asarray(x0.full() if hasattr(x0, "full") else x0, dtype=np.float64)
        pa = np.asarray(p.full() if hasattr(p, "full") else p, dtype=np.float64)
        # a batch is [B, n_w]; the reference's own column vector is (n_w, 1) (optimizer.py:602) -- B == n_w is a batch too
        batched = x0a.ndim == 2 and x0a.shape[1] == be.n_w and x0a.shape != (be.n_w, 1)
        if not batched:
            x0a = x0a.reshape(1, -1)
            pa = pa.reshape(1, -1)
        if lbg is not None or lbx is not None or ubg is not None or ubx is not None:
            be.set_bounds(lbx, ubx, lbg, ubg)
        res = be.solve(x0a, pa)
        rescued = np.zeros(res.status.shape[0], dtype=bool)
        n_rescued = int(be.last_rescued()) if hasattr(be, "last_rescued") else 0      # the second chance behind the C-ABI (a count, not a mask)
        if (self.rescue and not isinstance(be, BatchedMPCSolver) and not np.all(res.status == 1) and lbg is not None and lbx is not None
                and ubg is not None and ubx is not None):
            res, rescued = rescue_failed(be, x0a, pa, res, (lbx, ubx, lbg, ubg))      # stands in for IPOPT's restoration phase
        self._stats = dict(status=res.status.copy(), iter_count=res.iters.copy(), kkt=res.kkt.copy(), rescued=rescued, n_rescued=n_rescued + int(rescued.sum()),
                           success=bool(np.all(res.status == 1)),
                           return_status="Solve_Succeeded" if np.all(res.status == 1) else "Not_Converged")
        x = res.x if batched else res.x.reshape(-1, 1)
        out = NlpSolution(x=DMLike(x))
        out["status"] = res.status
        return out

    def stats(self):
        return dict(self._stats)


class Optimizer(object):
    """optimizer.py:33-83 -- pulls limits, weights, obstacle circles out of the planning configuration."""

    def __init__(self, configuration, init_values, predict_horizon):
        self.configuration = configuration
        self.delta_min = configuration.p.steering.min
        self.delta_max = configuration.p.steering.max
        self.deltav_min = configuration.p.steering.v_min
        self.deltav_max = configuration.p.steering.v_max
        self.v_min = 0
        self.v_max = configuration.p.longitudinal.v_max
        self.a_max = configuration.p.longitudinal.a_max
        self.init_position, self.init_velocity, self.init_acceleration, self.init_orientation = \
            init_values[0], init_values[1], init_values[2], init_values[3]
        self.iter_length = configuration.iter_length
        self.delta_t = configuration.delta_t
        self.desired_velocity = configuration.desired_velocity
        self.resampled_path_points = configuration.reference_path
        self.orientation = configuration.orientation
        self.predict_horizon = predict_horizon
        self.weights_setting = configuration.weights_setting
        so = configuration.static_obstacle
        self.obstacle_circles_centers_tuple = compute_centers_of_approximation_circles(
            so["position_x"], so["position_y"], so["length"], so["width"], so["orientation"])
        self.radius_obstacle, _ = compute_approximating_circle_radius(so["length"], so["width"])
        self.radius_ego, _ = compute_approximating_circle_radius(configuration.p.l, configuration.p.w)

    def equal_constraints(self, *args, **kwargs):
        pass

    def inequal_constraints(self, *args, **kwargs):
        pass

    def cost_function(self, *args, **kwargs):
        pass

    def solver(self):
        pass

    def optimize(self):
        pass


class CasadiOptimizer(Optimizer):
    """optimizer.py:369-702 with the NLP solved on the GPU."""

    num_states = 5
    num_controls = 2

    def __init__(self, configuration, init_values, predict_horizon, device=0):
        super(CasadiOptimizer, self).__init__(configuration, init_values, predict_horizon)
        self._device = device
        self._sol = None
        self._f = _DynamicsFunction()

    # -- bounds: optimizer.py:413-491 ---------------------------------------------------------------------------
    def inequal_constraints(self):
        N = self.predict_horizon
        lbg = [0.0]
        ubg = [self.a_max]
        for _ in range(N + 1):
            lbg += [0.0] * 5
            ubg += [0.0] * 5
        for _ in range(N + 1):
            lbg += [(self.radius_ego + self.radius_obstacle)] * 9
            ubg += [np.inf] * 9
        lbx, ubx = [], []
        for _ in range(N):
            lbx += [self.deltav_min, -np.inf]
            ubx += [self.deltav_max, self.a_max]
        for _ in range(N + 1):
            lbx += [-np.inf, -np.inf, self.delta_min, self.v_min, -np.inf]
            ubx += [np.inf, np.inf, self.delta_max, self.v_max, np.inf]
        return lbg, ubg, lbx, ubx

    # -- solver object: optimizer.py:513-560 ---------------------------------------------------------------------
    def solver(self):
        if self._sol is None:
            w = self.weights_setting
            Q = [w["weight_x"], w["weight_y"], w["weight_steering_angle"], w["weight_velocity"], w["weight_heading_angle"]]
            R = [w["weight_velocity_steering_angle"], w["weight_long_acceleration"]]
            Pt = [w["weight_x_terminate"], w["weight_y_terminate"], w["weight_steering_angle_terminate"],
                  w["weight_velocity_terminate"], w["weight_heading_angle_terminate"]]
            _, disc_distance = compute_approximating_circle_radius(self.configuration.p.l, self.configuration.p.w)
            centers = np.array(self.obstacle_circles_centers_tuple, dtype=np.float64)
            backend = BatchedMPCSolver(self.predict_horizon, 5, dt=self.delta_t, Q=Q, R=R, P=Pt, obstacle_centers=centers,
                                       ego_offset=(disc_distance / 2) / 2, max_iter=100, tol=1e-8, device=self._device)
            # the lower bound 0 of the stage-0 friction row (optimizer.py:378, 424): "nlp" (default: implied by the absolute value) or
            # "ipopt" (a barrier on it too, as IPOPT has -- a solve can then end at the kink of |.|; include/mpcgpu.h, option friction_lb)
            backend.set_option("friction_lb", str(getattr(self.configuration, "friction_lb", "nlp")))
            self._sol = NlpSolverHandle(backend)
        return self._sol, self._f

    # -- closed loop: optimizer.py:562-643 -----------------------------------------------------------------------
    # set to False to run the step-by-step host loop below even when the backend has an on-device driver
    use_device_loop = True

    def optimize(self):
        num_states, num_controls, N = 5, 2, self.predict_horizon
        lbg, ubg, lbx, ubx = self.inequal_constraints()
        sol, _ = self.solver()
        backend = getattr(sol, "_backend", None)
        noised = bool(self.configuration.noised)
        sigma = 0.1 if self.configuration.use_case == "lane_following" else 0.05          # optimizer.py:612-615
        # `noise_seed` (not in the reference, whose draws are unseeded and therefore not reproducible): with it the samples come
        # from the counter-based generator of noise.py / csrc/mpc_closed_loop.h and the loop can run on the device
        seed = getattr(self.configuration, "noise_seed", None)
        if self.use_device_loop and hasattr(backend, "closed_loop") and (not noised or seed is not None):
            # the whole loop of optimizer.py:596-631 on the device (mpc_closed_loop_batch_ex): no host round trip per step
            t_ = time.time()
            backend.set_bounds(lbx, ubx, lbg, ubg)
            init_state = np.array([self.init_position[0], self.init_position[1], 0.0, self.init_velocity, self.init_orientation])
            traj, ctrl, st = backend.closed_loop(init_state, self.resampled_path_points, self.orientation, self.desired_velocity,
                                                 self.iter_length, noise_mode=1 if noised else 0, sigma=sigma if noised else 0.0,
                                                 seed=0 if seed is None else int(seed))
            ok = bool(np.all(st == 1))
            sol._stats = dict(status=st[0].copy(), success=ok, return_status="Solve_Succeeded" if ok else "Not_Converged")
            return traj[0], ctrl[0], np.full(self.iter_length, (time.time() - t_) / self.iter_length)
        # ---- host-side loop: one solve per step through `sol(...)`, the plant and the warm start on the host.  Not the default path (the device
        #      loop above is); it is what runs the reference's unseeded noise and what tests/ check the device loop against.  State of the loop:
        #      the measured state, the last plan (controls 2 x N, states nx x (N + 1)) and whether a plan exists yet.
        x_now = np.array([self.init_position[0], self.init_position[1], 0.0, self.init_velocity, self.init_orientation], dtype=float)
        plan_u, plan_x = np.zeros((num_controls, N)), np.tile(x_now[:, None], (1, N + 1))
        window = np.tile(x_now, (N + 1, 1))                       # step 0 tracks the initial state itself
        applied, visited, seconds = [], [x_now.copy()], []
        for step in range(self.iter_length):
            p_vec = np.concatenate((np.zeros(num_controls * N), window.ravel()))[:, None]           # [U_ref = 0 | X_ref], optimizer.py:552
            w0 = self.warm_start(plan_u, plan_x, first=(step == 0))
            tic = time.time()
            sol, f = self.solver()
            w = sol(x0=w0, p=p_vec, lbg=lbg, lbx=lbx, ubg=ubg, ubx=ubx)["x"].full().ravel()
            seconds.append(time.time() - tic)
            plan_u = w[:num_controls * N].reshape(N, num_controls).T
            plan_x = w[num_controls * N:].reshape(N + 1, num_states).T
            if noised:
                # optimizer.py:611-615 perturbs the WHOLE predicted input sequence (20 samples at N = 10; 2 N here), applies its first column and
                # shifts the perturbed sequence into the next warm start
                eps = np.random.normal(0, sigma, 2 * N).reshape(num_controls, N) if seed is None else _noise.sequence_noise(int(seed), 0, step, N, sigma)
                plan_u = plan_u + eps
            applied.append(plan_u[:, 0].copy())
            x_now = x_now + self.delta_t * np.asarray(f(x_now, plan_u[:, 0]).full()).ravel()          # forward Euler plant, optimizer.py:649-651
            plan_u, plan_x = self.shift(plan_u), self.shift(plan_x)
            window, _ = self.desired_command_and_trajectory(step, x_now, N)
            visited.append(x_now.copy())
        # (what the reference returns: the states the solver was called AT -- the last propagated one is dropped, optimizer.py:639-641)
        return np.array(visited[:-1]), np.array(applied), np.array(seconds)

    @staticmethod
    def shift(plan):
        """a plan (columns = stages) moved one step ahead: the first column leaves, the last one is repeated (optimizer.py:653-654)"""
        return np.concatenate((plan[:, 1:], plan[:, -1:]), axis=1)

    @staticmethod
    def warm_start(plan_u, plan_x, first):
        """The vector handed to `sol(x0=...)`, with the layouts the reference's reshapes produce (optimizer.py:602) -- they are
        part of the behaviour: IPOPT starts where they put it.  The solver expects [u_0 .. u_{N-1} | x_0 .. x_N] stage by stage.
          states:   step 0 hands them over COMPONENT-major ([sx of all stages, sy of all stages, ...]: `next_states` is (N + 1, nx) there and is
                    transposed before it is flattened); from step 1 on stage-major, as expected
          controls: from step 1 on COMPONENT-major ([all steering rates, all accelerations]: `shift_movement` returns the transpose); at step 0 they
                    are zeros either way"""
        u_part = plan_u.ravel() if not first else np.zeros(plan_u.size)              # (2, N) row-major = component-major
        x_part = plan_x.ravel() if first else plan_x.T.ravel()                       # (nx, N + 1) row-major = component-major; transposed = stage-major
        return np.concatenate((u_part, x_part))[:, None]

    def shift_movement(self, t0, x0, u, x_f, f):
        """the reference's method of that name (optimizer.py:645-655), kept for callers that use it: Euler plant step + shifted plan; note the
        TRANSPOSED controls it returns (App. C-7)"""
        x_next = x0 + self.delta_t * f(x0, u[:, 0]).full()
        return t0 + self.delta_t, x_next, self.shift(u).T, self.shift(x_f)

    def desired_command_and_trajectory(self, i, x0_, N_):
        """Reference window of step i (optimizer.py:657-702): row 0 = the measured state, row 1 + k = (path point, delta = 0, desired velocity,
        path heading) at index i + 1 + k -- and once i >= L - N the window stops moving: it stays on the LAST N points of the path (App. C-10)."""
        L = self.iter_length
        first = min(i, L - self.predict_horizon - 1) + 1 if i >= L - self.predict_horizon else i + 1
        idx = first + np.arange(N_)
        rows = np.column_stack((self.resampled_path_points[idx, 0], self.resampled_path_points[idx, 1], np.zeros(N_),
                                np.full(N_, float(self.desired_velocity)), np.asarray(self.orientation)[idx]))
        return np.vstack((np.asarray(x0_, dtype=float).reshape(1, -1), rows)), np.zeros((N_, 2))


class ForcesInfo(object):
    """`info` of `solver.solve(problem)` (struct FORCESNLPsolver_info, FORCESNLPsolver.h:151-203): the fields the reference
    reads are `it` and `solvetime` (optimizer.py:331,355)."""

    def __init__(self, it, solvetime, res_eq, pobj=float("nan")):
        self.it = int(it)
        self.it2opt = int(it)
        self.solvetime = float(solvetime)
        self.fevalstime = 0.0
        self.QPtime = float(solvetime)
        self.res_eq = float(res_eq)
        self.rsnorm = float(res_eq)
        self.pobj = float(pobj)


class ForcesModel(object):
    """the fields of `forcespro.nlp.SymbolicModel` that optimizer.py:203-222 sets and the closed loop reads"""

    def __init__(self, N, backend, lb, ub, hl, hu):
        self.N, self.nvar, self.neq, self.nh, self.npar = N, 7, 5, 10, 10
        self.lb, self.ub, self.hl, self.hu = lb, ub, hl, hu
        self.E = np.concatenate([np.zeros((5, 2)), np.eye(5)], axis=1)
        self.xinitidx = range(2, 7)
        self._backend = backend

    def eq(self, z):
        """one RK4 step of the vehicle ODE from x = z[2:7] with u = z[0:2] (optimizer.py:91-98), on the device"""
        z = np.asarray(z, dtype=np.float64).ravel()
        return self._backend.plant_step(z[2:7], z[0:2], "rk4")


class ForcesSolverHandle(object):
    """`solver` of `model, solver = self.solver()`: `solver.solve(problem)` -> (output, exitflag, info)
    (test/FORCESNLPsolver/interface/FORCESNLPsolver_py.py:181-300), one SQP step per call on the device."""

    def __init__(self, backend, model, hessian_mode=0):
        self._backend, self._model = backend, model
        self.hessian_mode = int(hessian_mode)

    def solve(self, problem):
        N = self._model.N
        x0 = np.asarray(problem["x0"], dtype=np.float64)
        batched = x0.ndim == 3
        x0 = x0.reshape(-1, N, 7) if batched else x0.reshape(1, N, 7)
        B = x0.shape[0]
        xinit = np.asarray(problem["xinit"], dtype=np.float64).reshape(B, 5)
        par = np.asarray(problem["all_parameters"], dtype=np.float64).reshape(B, N, 10)
        t_ = time.time()
        m = self._model
        x, flag, it, res = self._backend.forces_solve(x0, xinit, par, m.lb, m.ub, m.hl, m.hu, hessian_mode=self.hessian_mode)
        dt = time.time() - t_
        fmt = "x{0:02d}" if N >= 10 else "x{0:1d}"
        if batched:
            output = {fmt.format(i + 1): x[:, i, :] for i in range(N)}
            return output, flag, [ForcesInfo(it[b], dt / B, res[b]) for b in range(B)]
        output = {fmt.format(i + 1): x[0, i, :] for i in range(N)}
        return output, int(flag[0]), ForcesInfo(it[0], dt, res[0])


class ForcesproOptimizer(Optimizer):
    """optimizer.py:86-366 with the FORCES-mode solve on the GPU (scope row f3).

    The formulation is the reference's: z = [deltaDot, aLong, x, y, delta, v, psi], RK4 shooting, friction circle and nine
    squared circle distances per stage, terminal weights, `xinit` on the first stage, ten run-time parameters per stage, ONE
    quadratic programme per call (SQP with `maxqps = 1`, BFGS initialised to 2.5 I).  The reference's solver for it is a
    generated, licence-locked binary (FORCESNLPsolver.h:209-210): its numerics cannot be matched, only its formulation
    (stage functions pinned against the generated C, row a11) and its call surface."""

    # QP Hessian of the SQP step: 0 = exact Hessian of the least-squares cost (Gauss-Newton; the default HERE), 1 = the literal
    # `bfgs_init = 2.5 I` of optimizer.py:234-237 (with one QP per call and a guess that is never refreshed it would stay
    # 2.5 I; see csrc/mpc_forces_qp.h: forces_hessian_diag and INTEGRATION.md for why that is not the default)
    hessian_mode = 0

    def __init__(self, configuration, init_values, predict_horizon, device=0, hessian_mode=None):
        super(ForcesproOptimizer, self).__init__(configuration, init_values, predict_horizon)
        self._device = device
        self._pair = None
        if hessian_mode is not None:
            self.hessian_mode = int(hessian_mode)

    def inequal_constraint(self):
        """optimizer.py:100-119."""
        z_low_bound = np.array([self.deltav_min, -self.a_max, -np.inf, -np.inf, self.delta_min, self.v_min, -np.inf])
        z_upper_bound = np.array([self.deltav_max, self.a_max, np.inf, np.inf, self.delta_max, self.v_max, np.inf])
        lo = np.concatenate((np.array([0]), np.tile(np.array([(self.radius_ego + self.radius_obstacle) ** 2]), 9)))
        hi = np.concatenate((np.array([self.a_max ** 2]), np.tile(np.array([np.inf]), 9)))
        return z_low_bound, z_upper_bound, lo, hi

    def solver(self):
        """optimizer.py:196-245: (model, solver); the device handle is created once and cached."""
        if self._pair is None:
            w = self.weights_setting
            Q = [w["weight_x"], w["weight_y"], w["weight_steering_angle"], w["weight_velocity"], w["weight_heading_angle"]]
            R = [w["weight_velocity_steering_angle"], w["weight_long_acceleration"]]
            Pt = [w["weight_x_terminate"], w["weight_y_terminate"], w["weight_steering_angle_terminate"],
                  w["weight_velocity_terminate"], w["weight_heading_angle_terminate"]]
            _, disc_distance = compute_approximating_circle_radius(self.configuration.p.l, self.configuration.p.w)
            # (the obstacle circle centres are run-time parameters 4..9 of every stage in the reference, optimizer.py:319-323; the
            #  device loop fills them from the handle's description, the host loop from runtime_parameters(): the same centres)
            backend = BatchedMPCSolver(self.predict_horizon, 5, dt=0.1, Q=Q, R=R, P=Pt,            # integrator_stepsize = 0.1, optimizer.py:97
                                       friction_div=self.configuration.wheelbase, ego_offset=(disc_distance / 2) / 2,
                                       obstacle_centers=np.array(self.obstacle_circles_centers_tuple, dtype=np.float64), device=self._device)
            lb, ub, hl, hu = self.inequal_constraint()
            model = ForcesModel(self.predict_horizon, backend, lb, ub, hl, hu)
            self._pair = (model, ForcesSolverHandle(backend, model, self.hessian_mode))
        return self._pair

    def runtime_parameters(self, k, N):
        """the (10, N) parameter block of step k (optimizer.py:292-318): next N path points, desired velocities ramping to 0
        over the last N steps, next N path orientations (all replenished with the last entry), obstacle circle centres"""
        L = self.iter_length
        v_all = np.hstack((np.ones(L - N) * self.desired_velocity, np.linspace(self.desired_velocity, 0, N)))
        v = v_all[k + 1:k + 1 + N]
        pts = self.resampled_path_points.T[:, k + 1:k + 1 + N]
        ori = np.asarray(self.orientation)[k + 1:k + 1 + N]
        while pts.shape[1] != N:
            pts = np.hstack((pts, self.resampled_path_points[-1].reshape(2, -1)))
            ori = np.hstack((ori, self.orientation[-1]))
            v = np.hstack((v, v_all[-1]))
        oc = np.array(self.obstacle_circles_centers_tuple, dtype=np.float64).reshape(6, 1)
        return np.vstack((pts, v, ori, np.tile(oc, (1, N))))

    use_device_loop = True      # the whole loop on the device (mpc_forces_closed_loop_batch); False: the step-by-step host loop below

    def optimize(self):
        """optimizer.py:246-366"""
        model, solver = self.solver()
        L = self.iter_length
        noised = bool(self.configuration.noised)
        seed = getattr(self.configuration, "noise_seed", None)
        backend = getattr(solver, "_backend", None)
        if self.use_device_loop and hasattr(backend, "forces_closed_loop") and (not noised or seed is not None):
            t_ = time.time()
            sigma = 0.1 if self.configuration.use_case == "lane_following" else 0.05
            init_state = np.array([self.init_position[0], self.init_position[1], 0.0, self.init_velocity, self.init_orientation])
            traj, ctrl, flag = backend.forces_closed_loop(init_state, self.resampled_path_points, self.orientation, self.desired_velocity, L,
                                                          model.lb, model.ub, model.hl, model.hu, init_acc=self.init_acceleration,
                                                          hessian_mode=self.hessian_mode, noise_mode=2 if noised else 0,
                                                          sigma=sigma if noised else 0.0, seed=0 if seed is None else int(seed))
            assert np.all(flag == 1), "bad exitflag"                               # optimizer.py:330
            return traj[0], ctrl[0], np.full(L, (time.time() - t_) / L)
        x = np.zeros((5, L + 1))
        u = np.zeros((2, L))
        solve_time = np.zeros(L)
        x0i = np.array([0.0, self.init_acceleration, self.init_position[0], self.init_position[1], 0.0, self.init_velocity, self.init_orientation])
        x0 = np.transpose(np.tile(x0i, (1, model.N)))
        xinit = np.array([self.init_position[0], self.init_position[1], 0.0, self.init_velocity, self.init_orientation])
        x[:, 0] = xinit
        # (the reference never refreshes problem["x0"]: every call linearises at the tiled initial guess, optimizer.py:264-274)
        problem = {"x0": x0.reshape(model.N, 7), "xinit": xinit, "ToleranceEqualities": 1e-1, "ToleranceInequalities": 1e-1}
        for k in range(L):
            problem["xinit"] = x[:, k]
            params = self.runtime_parameters(k, model.N)
            problem["all_parameters"] = np.reshape(np.transpose(params), (10 * model.N, 1))
            output, exitflag, info = solver.solve(problem)
            assert exitflag == 1, "bad exitflag"                                  # optimizer.py:330
            temp = np.zeros((model.nvar, model.N))
            for i in range(model.N):
                key = "x{0:1d}".format(i + 1)
                temp[:, i] = output[key] if key in output else output["x{0:02d}".format(i + 1)]
            pred_u = temp[0:2, :]
            if not self.configuration.noised:
                u[:, k] = pred_u[:, 0]
            else:
                sigma = 0.1 if self.configuration.use_case == "lane_following" else 0.05
                seed = getattr(self.configuration, "noise_seed", None)
                u[:, k] = pred_u[:, 0] + (np.random.normal(np.array([0, 0]), np.array([sigma, sigma]), (2,)) if seed is None
                                          else _noise.applied_noise(int(seed), 0, k, sigma))
            x[:, k + 1] = np.transpose(model.eq(np.concatenate((u[:, k], x[:, k]))))
            solve_time[k] = info.solvetime
        x = np.delete(x, -1, axis=1)
        return x.T, u.T, solve_time


__all__ = ["CasadiOptimizer", "ForcesproOptimizer", "ForcesSolverHandle", "ForcesModel", "ForcesInfo", "Optimizer", "NlpSolverHandle", "DMLike", "VehicleDynamics", "ca", "np",
           "find_closest_point", "compute_approximating_circle_radius", "compute_centers_of_approximation_circles"]
