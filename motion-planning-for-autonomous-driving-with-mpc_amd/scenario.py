"""Scenario -> planning configuration without the CommonRoad stack (scope row f2).

The reference builds its `PlanningConfiguration` in `MPC_Planner/configuration.py:400-623` on top of four third-party
packages that are not in this image (commonroad-io, commonroad-route-planner, commonroad-drivability-checker,
commonroad-vehicle-models).  This module restates that pipeline on plain numpy so that the inputs of
`CasadiOptimizer` (reference path, orientation, desired velocity, iteration length, obstacle, limits) can be produced
from a CommonRoad 2018b/2020a scenario XML and a settings yaml directly:

    scenario  = read_scenario("ZAM_Over-1_1.xml")                 # commonroad.common.file_reader (subset)
    settings  = yaml.safe_load(open("config_LF_ZAM_Over-1_1.yaml"))
    conf      = Configuration(settings, scenario, planning_problem_id=1).configuration
    states, controls, t = CasadiOptimizer(conf, init_values(scenario, 1), conf.predict_horizon).optimize()

What is restated from where (file:line of /root/reference, or the third-party function by name):
  * `Configuration.create_optimization_configuration_vehicle`   configuration.py:414-487
  * `find_reference_path_and_desired_velocity`                  configuration.py:500-552  (desired-velocity rule :538-544)
  * `clip_reference_path`                                       configuration.py:584-623
  * `find_closest_point`                                        configuration.py:26-37
  * `chaikins_corner_cutting`, `resample_polyline`, `compute_polyline_length`, `compute_orientation_from_polyline`:
    commonroad_dc.geometry.util -- source not in the tree; restated from the library's documented behaviour
  * route: commonroad_route_planner `RoutePlanner(...).plan_routes().retrieve_first_route().reference_path` -- source not
    in the tree; restated from the library's published behaviour (plan_route below): lanelet sequence start -> first goal
    lanelet over successor AND same-direction adjacent lanelets (lane changes), reference path from portions of the centre
    lines (a run of lane changes shares its stretch equally), a few vertices dropped around every hand-over, 2 m
    resampling, one Chaikin refinement
  * `parameters_vehicle2` (commonroad-vehicle-models): only the fields optimizer.py:34-46 reads

PARITY: unpinned against the third-party pieces (versions unknown).  What can be checked is checked in
tests/test_scenario.py against the reference's recorded runs: for ZAM_Over-1_1 (one lanelet) the pipeline reproduces the run
length (30 steps) and the recorded RMSD.txt from the recorded `planned states.txt` to 0.2 %; for USA_Lanker-2_18_T-1 (route
3672 -> 3452 -> lane changes over 3454 to 3456) the run length (70 steps), the first state and the recorded RMSD to +12 % / +5 %
(x / y) -- that figure moves by 50 ... 200 % when the lane-change construction is altered (portions, dropped vertices,
resampling step), so the recorded run does discriminate; the residual is the noise of the recorded run and whatever the
library version of 2021 did differently in detail.  ZAM_Tutorial-1_2_T-1 carries a planning problem WITHOUT a goal: the
reference's Configuration cannot plan it either (configuration.py:529 reads goal.state_list[0]); it raises here.
"""
from __future__ import annotations

import xml.etree.ElementTree as ET
from collections import deque
from types import SimpleNamespace

import numpy as np


# ----------------------------------------------------------------------------------------------------------------
# geometry utilities (commonroad_dc.geometry.util, restated)
# ----------------------------------------------------------------------------------------------------------------
def find_closest_point(path_points, current_point):
    """configuration.py:26-37"""
    diff = np.transpose(np.transpose(path_points) - np.asarray(current_point, dtype=np.float64).reshape(2, 1))
    sq = np.power(diff, 2)
    return int(np.argmin(sq[:, 0] + sq[:, 1]))


def chaikins_corner_cutting(polyline, refinements=1):
    """Chaikin's corner cutting: every refinement replaces each inner vertex by the 1/4 and 3/4 points of its edges
    (end points are kept)."""
    polyline = np.asarray(polyline, dtype=np.float64)
    for _ in range(refinements):
        L = polyline.repeat(2, axis=0)
        R = np.empty_like(L)
        R[0] = L[0]
        R[2::2] = L[1:-1:2]
        R[1:-1:2] = L[2::2]
        R[-1] = L[-1]
        polyline = L * 0.75 + R * 0.25
    return polyline


def resample_polyline(polyline, step=2.0):
    """points at arc-length multiples of `step` along the polyline; the last vertex is appended unless it coincides."""
    polyline = np.asarray(polyline, dtype=np.float64)
    if len(polyline) < 2:
        return np.array(polyline)
    new = [polyline[0]]
    current_position = step
    current_length = np.linalg.norm(polyline[0] - polyline[1])
    current_idx = 0
    while current_idx < len(polyline) - 1:
        if current_position >= current_length:
            current_position = current_position - current_length
            current_idx += 1
            if current_idx > len(polyline) - 2:
                break
            current_length = np.linalg.norm(polyline[current_idx + 1] - polyline[current_idx])
        else:
            rel = current_position / current_length
            new.append((1 - rel) * polyline[current_idx] + rel * polyline[current_idx + 1])
            current_position += step
    if np.linalg.norm(new[-1] - polyline[-1]) >= 1e-6:
        new.append(polyline[-1])
    return np.array(new)


def compute_polyline_length(polyline):
    polyline = np.asarray(polyline, dtype=np.float64)
    return float(np.sum(np.linalg.norm(np.diff(polyline, axis=0), axis=1)))


def compute_orientation_from_polyline(polyline):
    """heading of every segment; the last point repeats the heading of the last segment"""
    polyline = np.asarray(polyline, dtype=np.float64)
    d = np.diff(polyline, axis=0)
    o = np.arctan2(d[:, 1], d[:, 0])
    return np.concatenate((o, o[-1:]))


# ----------------------------------------------------------------------------------------------------------------
# CommonRoad XML (the subset the planner reads)
# ----------------------------------------------------------------------------------------------------------------
def _points(node):
    return np.array([[float(p.find("x").text), float(p.find("y").text)] for p in node.findall("point")], dtype=np.float64)


def _value(node, default=0.0):
    """<exact>v</exact> or the mid point of <intervalStart>/<intervalEnd>"""
    if node is None:
        return default
    e = node.find("exact")
    if e is not None:
        return float(e.text)
    a, b = node.find("intervalStart"), node.find("intervalEnd")
    return 0.5 * (float(a.text) + float(b.text))


def read_scenario(path):
    """lanelets, static obstacles and planning problems of a CommonRoad XML file (commonroad.common.file_reader subset)."""
    root = ET.parse(path).getroot()
    sc = SimpleNamespace(scenario_id=root.attrib.get("benchmarkID", ""), dt=float(root.attrib.get("timeStepSize", 0.1)),
                         lanelets={}, obstacles=[], dynamic_obstacles=[], planning_problems={})
    for l in root.findall("lanelet"):
        left, right = _points(l.find("leftBound")), _points(l.find("rightBound"))
        sc.lanelets[int(l.attrib["id"])] = SimpleNamespace(
            lanelet_id=int(l.attrib["id"]), left_vertices=left, right_vertices=right, center_vertices=0.5 * (left + right),
            successor=[int(s.attrib["ref"]) for s in l.findall("successor")],
            predecessor=[int(s.attrib["ref"]) for s in l.findall("predecessor")],
            adj_same=[int(a.attrib["ref"]) for a in (l.find("adjacentLeft"), l.find("adjacentRight"))
                      if a is not None and a.attrib.get("drivingDir") == "same"],
            adj_left=(int(l.find("adjacentLeft").attrib["ref"]) if l.find("adjacentLeft") is not None
                      and l.find("adjacentLeft").attrib.get("drivingDir") == "same" else None),
            adj_right=(int(l.find("adjacentRight").attrib["ref"]) if l.find("adjacentRight") is not None
                       and l.find("adjacentRight").attrib.get("drivingDir") == "same" else None),
            adj_left_opposite=(int(l.find("adjacentLeft").attrib["ref"]) if l.find("adjacentLeft") is not None
                               and l.find("adjacentLeft").attrib.get("drivingDir") == "opposite" else None))
    for tag in ("obstacle", "dynamicObstacle"):
        for o in root.findall(tag):                     # moving obstacles: rectangle + one (x, y, orientation) per time step
            role = o.find("role")
            if tag == "obstacle" and (role is None or role.text != "dynamic"):
                continue
            rect, ini, trj = o.find("shape/rectangle"), o.find("initialState"), o.find("trajectory")
            if rect is None or ini is None:
                continue
            states = {}
            for st in [ini] + (list(trj.findall("state")) if trj is not None else []):
                tt = int(round(_value(st.find("time"))))
                states[tt] = (float(st.find("position/point/x").text), float(st.find("position/point/y").text), _value(st.find("orientation")))
            sc.dynamic_obstacles.append(SimpleNamespace(obstacle_id=int(o.attrib["id"]), length=float(rect.find("length").text),
                                                        width=float(rect.find("width").text), states=states))
    for tag in ("obstacle", "staticObstacle"):
        for o in root.findall(tag):
            role = o.find("role")
            if tag == "obstacle" and role is not None and role.text != "static":
                continue
            rect = o.find("shape/rectangle")
            ini = o.find("initialState")
            if rect is None or ini is None:
                continue
            sc.obstacles.append(SimpleNamespace(
                obstacle_id=int(o.attrib["id"]), length=float(rect.find("length").text), width=float(rect.find("width").text),
                position=np.array([float(ini.find("position/point/x").text), float(ini.find("position/point/y").text)]),
                orientation=_value(ini.find("orientation"))))
    for pp in root.findall("planningProblem"):
        ini = pp.find("initialState")
        goal = pp.find("goalState")
        gpos = goal.find("position") if goal is not None else None
        center, goal_lanelets = None, []
        if gpos is not None:
            for shape in ("rectangle", "circle"):
                c = gpos.find(shape + "/center")
                if c is not None:
                    center = np.array([float(c.find("x").text), float(c.find("y").text)])
            goal_lanelets = [int(g.attrib["ref"]) for g in gpos.findall("lanelet")]
        time_end = None
        if goal is not None:
            t = goal.find("time")
            t_exact = t.find("exact")
            time_end = int(float(t_exact.text)) if t_exact is not None else int(float(t.find("intervalEnd").text))
        sc.planning_problems[int(pp.attrib["id"])] = SimpleNamespace(
            planning_problem_id=int(pp.attrib["id"]),
            initial_position=np.array([float(ini.find("position/point/x").text), float(ini.find("position/point/y").text)]),
            initial_velocity=_value(ini.find("velocity")), initial_orientation=_value(ini.find("orientation")),
            initial_acceleration=_value(ini.find("acceleration")),
            goal_center=center, goal_lanelets=goal_lanelets, goal_time_end=time_end)
    return sc


# ----------------------------------------------------------------------------------------------------------------
# route (commonroad_route_planner, restated as "centre line of the first lanelet sequence start -> goal")
# ----------------------------------------------------------------------------------------------------------------
def _lanelet_of_point(lanelets, pt):
    best, dist = None, np.inf
    for lid, l in lanelets.items():
        d = np.min(np.linalg.norm(l.center_vertices - pt, axis=1))
        if d < dist:
            best, dist = lid, d
    return best


def plan_route(scenario, planning_problem, step_resample=1.0, num_vertices_lane_change_max=6, percentage_vertices_lane_change_max=0.1):
    """returns (reference_path ndarray(n,2), list of lanelet ids).

    Lanelet sequence: breadth-first from the lanelet of the initial position to the FIRST goal lanelet of the planning problem
    (the reference takes `retrieve_first_route()`), moving to successors and to adjacent lanelets of the same driving direction
    (a lane change).  Reference path, as the route planner builds it: every lanelet contributes the portion [a, b] of its centre
    line (resampled to `step_resample`) -- [0, 1] normally, while the m lanelets of a run of lane changes share the stretch,
    [q/m, (q+1)/m] -- minus a few vertices on either side of every hand-over, so that the change of lane becomes a diagonal
    instead of a jump; then 2 m resampling and FOUR Chaikin refinements (the route planner's own `chaikins_corner_cutting`, whose default
    is `num_refinements = 4` -- not commonroad_dc's, whose default is 1: pinned by the recorded deviation.txt of the USA_Lanker run, which
    this path reproduces to 1e-14, tests/test_scenario.py)."""
    lan = scenario.lanelets
    start = _lanelet_of_point(lan, planning_problem.initial_position)
    if planning_problem.goal_lanelets:
        goals = {planning_problem.goal_lanelets[0]}
    elif planning_problem.goal_center is not None:
        goals = {_lanelet_of_point(lan, planning_problem.goal_center)}
    else:
        goals = set()
    prev = {start: None}
    queue = deque([start])
    hit = start if (start in goals or not goals) else None
    while queue and hit is None:
        cur = queue.popleft()
        for s in list(lan[cur].successor) + list(getattr(lan[cur], "adj_same", [])):
            if s in lan and s not in prev:
                prev[s] = cur
                if s in goals:
                    hit = s
                    break
                queue.append(s)
    if hit is None:                       # goal not reachable: stay on the start lanelet
        hit = start
    ids = []
    while hit is not None:
        ids.append(hit)
        hit = prev[hit]
    ids.reverse()
    n = len(ids)
    change = [1 if (i + 1 < n and ids[i + 1] in getattr(lan[ids[i]], "adj_same", [])) else 0 for i in range(n)]
    portions, i = [], 0
    while i < n:
        if not change[i]:
            portions.append((0.0, 1.0))
            i += 1
            continue
        j = i
        while j < n and change[j]:
            j += 1
        m = j - i + 1                     # lanelets of this run of lane changes
        portions += [(q / m, (q + 1) / m) for q in range(m)]
        i = j + 1
    path = None
    for k, lid in enumerate(ids):
        v = resample_polyline(lan[lid].center_vertices, step_resample)
        nv = len(v)
        n_lc = min(int(nv * percentage_vertices_lane_change_max) + 1, num_vertices_lane_change_max)
        i0 = int(portions[k][0] * nv)
        i1 = int(portions[k][1] * nv)
        if path is None:
            i1 = max(i1, 1)
        else:
            i0 = min(i0 + n_lc, nv - 1)
        if k != n - 1:
            i1 = max(i1 - n_lc, 1)
        part = v[i0:i1]
        path = part if path is None else np.concatenate((path, part), axis=0)
    return chaikins_corner_cutting(resample_polyline(path, 2.0), refinements=4), ids


# ----------------------------------------------------------------------------------------------------------------
# configuration (configuration.py:400-623)
# ----------------------------------------------------------------------------------------------------------------
def parameters_vehicle2():
    """commonroad-vehicle-models `parameters_vehicle2` (BMW 320i): the fields optimizer.py:34-46 and configuration.py read."""
    return SimpleNamespace(l=4.508, w=1.610, a=1.1561957, b=1.4227171,
                           steering=SimpleNamespace(min=-1.066, max=1.066, v_min=-0.4, v_max=0.4),
                           longitudinal=SimpleNamespace(v_min=-13.6, v_max=50.8, v_switch=7.319, a_max=11.5))


def clip_reference_path(origin_reference_path, init_position, goal_position):
    """configuration.py:584-623"""
    start_index = find_closest_point(origin_reference_path, init_position)
    end_index = find_closest_point(origin_reference_path, goal_position)
    if goal_position[0] >= init_position[0]:
        diff_init = (origin_reference_path[start_index] - init_position) >= 0
        diff_goal = (origin_reference_path[end_index] - goal_position) <= 0
    else:
        diff_init = (origin_reference_path[start_index] - init_position) <= 0
        diff_goal = (origin_reference_path[end_index] - goal_position) >= 0
    if diff_init.sum() != 2:
        start_index = start_index + 1
    if diff_goal.sum() != 2:
        end_index = end_index - 1
    return np.concatenate((init_position.reshape(1, 2), origin_reference_path[start_index:end_index + 1], goal_position.reshape(1, 2)), axis=0)


class Configuration(object):
    """configuration.py:400-487: `.configuration` carries what Optimizer.__init__ (optimizer.py:34-68) reads."""

    def __init__(self, settings, scenario, planning_problem_id=None):
        self.settings = settings
        self.scenario = scenario
        if planning_problem_id is None:
            planning_problem_id = sorted(scenario.planning_problems)[0]
        self.planning_problem = scenario.planning_problems[planning_problem_id]
        self.configuration = self.create_optimization_configuration_vehicle()

    def find_reference_path_and_desired_velocity(self):
        """configuration.py:500-552"""
        pp = self.planning_problem
        vehicle_settings = self.settings["vehicle_settings"][pp.planning_problem_id]
        if pp.goal_time_end is None:
            raise ValueError("planning problem {} has no goal state: the desired velocity is defined by the goal's time limit "
                             "(configuration.py:529-538), the reference cannot plan it either".format(pp.planning_problem_id))
        origin_reference_path, lanelets_leading_to_goal = plan_route(self.scenario, pp)
        goal_position = pp.goal_center if pp.goal_center is not None else origin_reference_path[-1]
        clipped = clip_reference_path(origin_reference_path, pp.initial_position, goal_position)
        length_clipped_path = compute_polyline_length(clipped)
        delta_t = self.scenario.dt
        desired_velocity = length_clipped_path / ((pp.goal_time_end - 1) * delta_t)
        if desired_velocity > round(desired_velocity, 4):                       # configuration.py:540-544
            desired_velocity = round(desired_velocity, 4) + 0.0001
        else:
            desired_velocity = round(desired_velocity, 4)
        if vehicle_settings["resampling_reference_path"]:
            resampled = resample_polyline(np.array(chaikins_corner_cutting(clipped)), step=desired_velocity * delta_t)
        else:
            resampled = clipped
        return origin_reference_path, resampled, lanelets_leading_to_goal, desired_velocity, delta_t

    def create_optimization_configuration_vehicle(self):
        """configuration.py:414-487"""
        pp = self.planning_problem
        assert pp.planning_problem_id in self.settings["vehicle_settings"], \
            "Cannot find settings for planning problem {}".format(pp.planning_problem_id)
        vehicle_settings = self.settings["vehicle_settings"][pp.planning_problem_id]
        c = SimpleNamespace()
        origin, reference_path, lanelets, desired_velocity, delta_t = self.find_reference_path_and_desired_velocity()
        c.origin_reference_path = origin
        c.reference_path = np.array(reference_path)
        c.lanelets_leading_to_goal = lanelets
        c.desired_velocity = desired_velocity
        c.delta_t = delta_t
        c.iter_length = reference_path.shape[0]
        c.orientation = compute_orientation_from_polyline(reference_path)
        c.predict_horizon = self.settings["general_planning_settings"]["predict_horizon"]
        c.reference_point = vehicle_settings.get("reference_point", "rear")
        c.vehicle_id = pp.planning_problem_id
        if vehicle_settings["vehicle_model"] != "parameters_vehicle2":
            raise ValueError("only parameters_vehicle2 is restated (the reference's config files use no other model)")
        c.p = parameters_vehicle2()
        c.wheelbase = vehicle_settings["wheelbase"]
        c.framework_name = self.settings["general_planning_settings"]["framework_name"]
        c.noised = self.settings["general_planning_settings"]["noised"]
        c.weights_setting = self.settings["weights_setting"]
        c.use_case = self.settings["scenario_settings"]["use_case"]
        if c.use_case == "collision_avoidance":
            o = self.scenario.obstacles[0]
            c.static_obstacle = {"position_x": o.position[0], "position_y": o.position[1], "length": o.length, "width": o.width,
                                 "orientation": o.orientation}
        elif c.use_case == "lane_following":
            c.static_obstacle = {"position_x": -100.0, "position_y": 0.0, "length": 0.0, "width": 0.0, "orientation": 0.0}
        else:
            raise ValueError("use_case can only be lane_following and collision_avoidance!")
        return c


def obstacle_rectangles(scenario, steps):
    """[n, steps, 5] rows (x, y, length, width, orientation) of every obstacle of the scenario at time steps 0 .. steps-1 -- the
    input of mpc_validity_batch / mpc_planner.collision_verdict.  Static obstacles repeat their row; a moving obstacle that does not
    exist at a time step gets length = width = 0 there."""
    rows = []
    for o in scenario.obstacles:
        rows.append(np.tile([o.position[0], o.position[1], o.length, o.width, o.orientation], (steps, 1)))
    for o in scenario.dynamic_obstacles:
        r = np.zeros((steps, 5))
        for i in range(steps):
            if i in o.states:
                r[i] = [o.states[i][0], o.states[i][1], o.length, o.width, o.states[i][2]]
        rows.append(r)
    return np.array(rows) if rows else np.zeros((0, steps, 5))


def road_corridor(scenario, lanelet_ids, include_oncoming=True):
    """(left, right) boundary polylines, in driving direction, of the road along a lanelet sequence (a route): for every lanelet
    of the sequence the RIGHT bound of its rightmost neighbour of the same driving direction and, on the other side, the LEFT
    bound of its leftmost one -- or, with include_oncoming (what commonroad_dc's road boundary of the whole network amounts to
    on a two-way road, and what configuration.py:432-433 picks for ZAM_Over-1_1), the far bound of the oncoming lanes next to
    it.  Consecutive lanelets of one lane-change run share a cross-section and contribute it once."""
    lan = scenario.lanelets
    left, right, seen = [], [], set()
    for lid in lanelet_ids:
        lm = lid
        while getattr(lan[lm], "adj_left", None) in lan:
            lm = lan[lm].adj_left
        rm = lid
        while getattr(lan[rm], "adj_right", None) in lan:
            rm = lan[rm].adj_right
        if (lm, rm) in seen:
            continue
        seen.add((lm, rm))
        left_pts = lan[lm].left_vertices
        opp = getattr(lan[lm], "adj_left_opposite", None)
        if include_oncoming and opp in lan:
            while getattr(lan[opp], "adj_right", None) in lan:          # rightmost lane of the oncoming direction
                opp = lan[opp].adj_right
            left_pts = lan[opp].right_vertices[::-1]
        for acc, pts in ((left, left_pts), (right, lan[rm].right_vertices)):
            acc.append(pts if not acc or not np.allclose(acc[-1][-1], pts[0]) else pts[1:])
    return np.concatenate(left, axis=0), np.concatenate(right, axis=0)


def init_values(scenario, planning_problem_id=None):
    """(position, velocity, acceleration, orientation) as MPCPlanner hands them to the optimizers (mpc_planner.py:30-59)"""
    if planning_problem_id is None:
        planning_problem_id = sorted(scenario.planning_problems)[0]
    pp = scenario.planning_problems[planning_problem_id]
    return pp.initial_position, pp.initial_velocity, pp.initial_acceleration, pp.initial_orientation


__all__ = ["read_scenario", "Configuration", "init_values", "plan_route", "obstacle_rectangles", "road_corridor", "clip_reference_path", "find_closest_point",
           "chaikins_corner_cutting", "resample_polyline", "compute_polyline_length", "compute_orientation_from_polyline",
           "parameters_vehicle2"]
