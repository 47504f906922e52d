"""
ORACLE (test infrastructure, not product code) -- numpy restatement of the reference's post-hoc trajectory metrics
(SURVEY.md section 8 row f4):

  deviation_euclidean   MPCPlanner.plot_deviation_euclidean_dis, MPC_Planner/mpc_planner.py:184-199, with
                        find_closest_point, MPC_Planner/configuration.py:26-37 (first index of the minimum squared distance)
  rmsd_xy               MPCPlanner.compute_rmsd, mpc_planner.py:279-292 -- note the divisor L - 1 over L squared terms
  min_clearance         the quantity the NLP constrains (optimizer.py:395-403): distance between ego circle j
                        (configuration.py:69-93) and obstacle circle j, minus r_ego + r_obstacle (or all 9 pairs); the
                        reference's own check goes through commonroad_dc (test_mpc_planner.py:37-47, not in the image)

Parity: the recorded runs (test/2D_plots_*) keep deviation.txt / RMSD.txt but not the paths they were computed against,
so these functions are pinned by the formulas only ("parity unpinned" for the numbers; the GPU path is held bit-exact
to this restatement).
"""
import numpy as np


def find_closest_point(path_points, current_point):
    diff = np.transpose(np.transpose(path_points) - np.asarray(current_point, dtype=np.float64).reshape(2, 1))
    sq = np.power(diff, 2)
    return int(np.argmin(sq[:, 0] + sq[:, 1]))


def deviation_euclidean(x, origin_path):
    x = np.asarray(x, dtype=np.float64)
    origin_path = np.asarray(origin_path, dtype=np.float64)
    nearest = np.zeros((x.shape[0], 2))
    for i in range(x.shape[0]):
        nearest[i] = origin_path[find_closest_point(origin_path, x[i, 0:2])]
    dx = nearest[:, 0] - x[:, 0]
    dy = nearest[:, 1] - x[:, 1]
    return np.sqrt(dx ** 2 + dy ** 2)


def rmsd_xy(x, reference_path):
    x = np.asarray(x, dtype=np.float64)
    reference_path = np.asarray(reference_path, dtype=np.float64)
    L = x.shape[0]
    sum_x = 0.0
    sum_y = 0.0
    for i in range(L):
        sum_x += (reference_path[i, 0] - x[i, 0]) ** 2
        sum_y += (reference_path[i, 1] - x[i, 1]) ** 2
    return np.array([np.sqrt(sum_x / (L - 1)), np.sqrt(sum_y / (L - 1))])


def min_clearance(x, obstacle_centers, ego_offset, r_sum, all_pairs=False):
    """x: (L,5) states; obstacle_centers: (3,2); min over steps and circle pairs of dist - r_sum.  The reference constrains
    only the pairs (ego circle j, obstacle circle j), each three times (optimizer.py:395-403); all_pairs=True: all nine."""
    x = np.asarray(x, dtype=np.float64)
    oc = np.asarray(obstacle_centers, dtype=np.float64).reshape(3, 2)
    best = np.inf
    for i in range(x.shape[0]):
        sx, sy, psi = x[i, 0], x[i, 1], x[i, 4]
        c, s = np.cos(psi), np.sin(psi)
        for e, sg in enumerate((0.0, 1.0, -1.0)):
            ex, ey = sx + sg * ego_offset * c, sy + sg * ego_offset * s
            for j in range(3):
                if not all_pairs and j != e:
                    continue
                d = np.sqrt((ex - oc[j, 0]) * (ex - oc[j, 0]) + (ey - oc[j, 1]) * (ey - oc[j, 1])) - r_sum
                best = min(best, d)
    return best


def _seg_side(poly, px, py):
    """signed side of (px, py) w.r.t. the nearest segment of a polyline: > 0 left of it, < 0 right of it"""
    best, side = np.inf, 0.0
    for q in range(len(poly) - 1):
        ax, ay, bx, by = poly[q, 0], poly[q, 1], poly[q + 1, 0], poly[q + 1, 1]
        ex, ey = bx - ax, by - ay
        l2 = ex * ex + ey * ey
        t = ((px - ax) * ex + (py - ay) * ey) / l2 if l2 > 0.0 else 0.0
        t = min(1.0, max(0.0, t))
        dx, dy = px - (ax + t * ex), py - (ay + t * ey)
        d2 = dx * dx + dy * dy
        if d2 < best:
            best, side = d2, ex * (py - ay) - ey * (px - ax)
    return side


def validity(x, obstacles=None, left=None, right=None, ego_length=4.3, ego_width=1.8):
    """collision / road verdict of one trajectory (L,5) -- the check of test/test_mpc_planner.py:37-47 restated without
    commonroad_dc: ego rectangle (mpc_planner.py:99) of step i against the obstacle rectangles [n,L,5] of step i (separating-axis
    test) and against the corridor between the boundary polylines.  Returns (first collision step | -1, first off-road step | -1)."""
    x = np.asarray(x, dtype=np.float64)
    L = x.shape[0]
    hl, hw = 0.5 * ego_length, 0.5 * ego_width
    first_c, first_o = -1, -1
    for i in range(L):
        cx, cy, psi = x[i, 0], x[i, 1], x[i, 4]
        sn, cs = np.sin(psi), np.cos(psi)
        if first_c < 0 and obstacles is not None:
            for o in range(len(obstacles)):
                r = obstacles[o][i]
                if not (r[2] > 0.0 and r[3] > 0.0):
                    continue
                so, co = np.sin(r[4]), np.cos(r[4])
                ol, ow, dx, dy = 0.5 * r[2], 0.5 * r[3], r[0] - cx, r[1] - cy
                c00, c01 = abs(cs * co + sn * so), abs(-cs * so + sn * co)
                sep = (abs(dx * cs + dy * sn) > hl + ol * c00 + ow * c01 or abs(-dx * sn + dy * cs) > hw + ol * c01 + ow * c00 or
                       abs(dx * co + dy * so) > ol + hl * c00 + hw * c01 or abs(-dx * so + dy * co) > ow + hl * c01 + hw * c00)
                if not sep:
                    first_c = i
                    break
        if first_o < 0 and (left is not None or right is not None):
            for q in range(4):
                sl, sw = (-hl if q & 1 else hl), (-hw if q & 2 else hw)
                px, py = cx + sl * cs - sw * sn, cy + sl * sn + sw * cs
                if (left is not None and len(left) > 1 and _seg_side(left, px, py) > 0.0) or \
                        (right is not None and len(right) > 1 and _seg_side(right, px, py) < 0.0):
                    first_o = i
                    break
    return first_c, first_o
