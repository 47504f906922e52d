"""
The caller of the hot path: look-alike of the reference's `MPC_Planner/mpc_planner.py::MPCPlanner` without its plotting / GIF
output (out of scope).

    MPCPlanner(scenario, planning_problem, configuration, predict_horizon)          mpc_planner.py:21-28
        .get_init_values() -> (position ndarray(2,), velocity, acceleration, orientation)        :30-59
        .plan() -> (ego_vehicle_trajectory, ego_vehicle)                                          :296-314

`plan()` instantiates the optimizer the configuration names (`framework_name`: casadi | forcespro, :301-306), runs `optimize()`
-- the NLP solves, the closed loop and the metrics all on the GPU -- and, like `plot_and_create_gif` (:77-182), turns the planned
states into the ego vehicle's trajectory (time steps 1 .. L-1, a 4.3 x 1.8 m rectangle, :84-109) and writes the reference's result
files when `save_dir` is given: `deviation.txt` (:184-199), `control inputs.txt` (:207), `solve time.txt` (:232), `planned states.txt`
(:248), `RMSD.txt` for lane following (:279-292).  `collision_check()` is what the reference's only test does with the result
(test/test_mpc_planner.py:37-47).  Scenario / planning problem: the objects of `scenario.read_scenario` (or anything with the
CommonRoad attribute names `initial_state.position / velocity / acceleration / orientation`).
"""
from __future__ import annotations

import os
from types import SimpleNamespace

import numpy as np

from . import scenario as _scenario
from .optimizer import CasadiOptimizer, ForcesproOptimizer

EGO_SHAPE = SimpleNamespace(length=4.3, width=1.8)          # mpc_planner.py:99


# ---- the planner's post-hoc metrics (scope row f4): `plot_deviation_euclidean_dis` (mpc_planner.py:184-199, the array it saves as
# deviation.txt) and `compute_rmsd` (:279-292, RMSD.txt) without the plotting, the clearance of the 3 x 3 approximation circles the NLP
# constrains (optimizer.py:395-411) and the collision / road verdict of the reference's test.  `solver`: a BatchedMPCSolver; the arithmetic
# runs on the device (mpc_metrics_batch / mpc_validity_batch of include/mpcgpu.h), there is no CPU path.
def deviation_euclidean_dis(solver, x, origin_reference_path):
    """deviation.txt of mpc_planner.py:184-199 for one trajectory (L,5) or a batch (B,L,5)."""
    x = np.asarray(x, dtype=np.float64)
    out = solver.metrics(x, origin_path=origin_reference_path)["deviation"]
    return out[0] if x.ndim == 2 else out


def compute_rmsd(solver, x, reference_path):
    """(rmsd_x, rmsd_y) of mpc_planner.py:279-292 (divisor L - 1) for one trajectory or a batch."""
    x = np.asarray(x, dtype=np.float64)
    out = solver.metrics(x, ref_path=reference_path)["rmsd"]
    return out[0] if x.ndim == 2 else out


def min_clearance(solver, x, r_sum, all_pairs=False):
    """min over steps and circle pairs of (centre distance - r_sum).  Default: the three pairs (ego circle j, obstacle
    circle j) the reference constrains (optimizer.py:395-403); all_pairs=True: all nine."""
    x = np.asarray(x, dtype=np.float64)
    out = solver.metrics(x, r_sum=r_sum, all_pairs=all_pairs)["clearance"]
    return out[0] if x.ndim == 2 else out


def collision_verdict(solver, x, obstacles=None, left_boundary=None, right_boundary=None, ego_length=4.3, ego_width=1.8):
    """what test/test_mpc_planner.py:37-47 prints -- does the ego vehicle (mpc_planner.py:99: a 4.3 x 1.8 m rectangle on the planned
    states) collide with an obstacle of the scenario or with the road boundary? -- for one trajectory (L,5) or a batch (B,L,5).
    Returns (collides, first_collision_step, leaves_road, first_off_road_step); scenario.obstacle_rectangles / road_corridor build
    the obstacle rows and boundary polylines from a scenario."""
    x = np.asarray(x, dtype=np.float64)
    r = solver.validity(x, obstacles=obstacles, left=left_boundary, right=right_boundary, ego_length=ego_length, ego_width=ego_width)
    fc, fo = r["first_collision"], r["first_off_road"]
    if x.ndim == 2:
        return bool(fc[0] >= 0), int(fc[0]), bool(fo[0] >= 0), int(fo[0])
    return fc >= 0, fc, fo >= 0, fo


class MPCPlanner(object):
    def __init__(self, scenario, planning_problem, configuration, predict_horizon, device=0):
        self.scenario = scenario
        self.planning_problem = planning_problem
        self.configuration = configuration
        self.init_values = self.get_init_values()
        self.predict_horizon = predict_horizon
        self.results = None
        self._device = device
        self._optimizer = None

    def get_init_values(self):
        """mpc_planner.py:30-59 (missing attributes default to 0, position to the origin)"""
        pp = self.planning_problem
        st = getattr(pp, "initial_state", None)
        if st is not None:
            return (np.asarray(getattr(st, "position", np.array([0, 0])), dtype=np.float64), getattr(st, "velocity", 0), getattr(st, "acceleration", 0.),
                    getattr(st, "orientation", 0))
        return (np.asarray(getattr(pp, "initial_position", np.array([0, 0])), dtype=np.float64), getattr(pp, "initial_velocity", 0),
                getattr(pp, "initial_acceleration", 0.), getattr(pp, "initial_orientation", 0))

    def plan(self, save_dir=None):
        """mpc_planner.py:296-314"""
        conf = self.configuration
        if conf.framework_name == "casadi":
            optimizer = CasadiOptimizer(configuration=conf, init_values=self.init_values, predict_horizon=self.predict_horizon, device=self._device)
        elif conf.framework_name == "forcespro":
            optimizer = ForcesproOptimizer(configuration=conf, init_values=self.init_values, predict_horizon=self.predict_horizon, device=self._device)
        else:
            raise ValueError("Only casadi and forcespro are available!")
        self._optimizer = optimizer
        final_states, final_control_inputs, final_solve_time = optimizer.optimize()
        return self.create_ego_vehicle(final_states, final_control_inputs, final_solve_time, save_dir)

    def _backend(self):
        pair = self._optimizer.solver()
        return pair[0]._backend if self.configuration.framework_name == "casadi" else pair[1]._backend

    def create_ego_vehicle(self, x, u, solve_time, save_dir=None):
        """what plot_and_create_gif (mpc_planner.py:77-182) produces besides pictures"""
        conf = self.configuration
        initial_state = SimpleNamespace(position=np.array([self.init_values[0][0], self.init_values[0][1]]), velocity=self.init_values[2],
                                        orientation=self.init_values[3], time_step=0)                       # (:81-84; `velocity` is the acceleration there too)
        state_list = [SimpleNamespace(position=np.array([x[i, 0], x[i, 1]]), velocity=x[i, 3], orientation=x[i, 4], time_step=i)
                      for i in range(1, conf.iter_length)]
        trajectory = SimpleNamespace(initial_time_step=1, state_list=state_list)
        ego_vehicle = SimpleNamespace(obstacle_shape=EGO_SHAPE, initial_state=initial_state,
                                      prediction=SimpleNamespace(trajectory=trajectory, shape=EGO_SHAPE), obstacle_type="car")
        be = self._backend()
        deviation = deviation_euclidean_dis(be, x, conf.origin_reference_path)
        rmsd = compute_rmsd(be, x, conf.reference_path) if conf.use_case == "lane_following" else None
        self.results = dict(states=x, controls=u, solve_time=solve_time, deviation=deviation, rmsd=rmsd)
        if save_dir is not None:
            os.makedirs(save_dir, exist_ok=True)
            np.savetxt(os.path.join(save_dir, "deviation.txt"), deviation)
            np.savetxt(os.path.join(save_dir, "control inputs.txt"), u)
            np.savetxt(os.path.join(save_dir, "solve time.txt"), solve_time)
            np.savetxt(os.path.join(save_dir, "planned states.txt"), x)
            if rmsd is not None:
                np.savetxt(os.path.join(save_dir, "RMSD.txt"), np.asarray(rmsd).reshape(2, 1))
        return trajectory, ego_vehicle

    def collision_check(self):
        """test/test_mpc_planner.py:37-47: does the planned ego vehicle collide with an obstacle of the scenario or leave the road?
        -> (collides, first step, leaves road, first step); plan() first"""
        assert self.results is not None, "plan() first"
        conf = self.configuration
        x = self.results["states"]
        # every obstacle of the scenario, static and moving, whatever the use case (the reference's test builds its collision checker from
        # the whole scenario, test_mpc_planner.py:40; the lane-following variant of ZAM_Over-1_1 simply has none)
        obst = _scenario.obstacle_rectangles(self.scenario, x.shape[0])
        left, right = _scenario.road_corridor(self.scenario, conf.lanelets_leading_to_goal)
        return collision_verdict(self._backend(), x, obst, left, right, EGO_SHAPE.length, EGO_SHAPE.width)


__all__ = ["MPCPlanner", "deviation_euclidean_dis", "compute_rmsd", "min_clearance", "collision_verdict"]
