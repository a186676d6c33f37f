"""Motion-primitive hybrid-A* planner with GPU successor generation.

Call surface of the reference's five search variants:
  main/lib/motion_primitive_search.py             (heuristic: distance to the goal BOX + 2.7*excess heading error)
  main/lib/motion_primitive_search_modified.py    (heuristic: distance to the goal POINT + 2.7*(|dtheta| - tol/2))
  main/lib/motion_primitive_search_multi_lane.py  (weighted heuristic / edge cost terms)
  main/lib/motion_primitive_search_roundabout.py  (modified heuristic; edge = length + 0.1/obstacle distance + 5*|dtheta|)
  main/lib/motion_primitive_search_single_lane.py (heuristic + 15*|dtheta to goal|; edge = length + 5*|dtheta| + 0.1/obstacle distance)
selected by `variant`; the sibling modules `motion_primitive_search_modified` / `_multi_lane` / `_roundabout` /
`_single_lane` export the same class under the reference's names.

Successors, collision flags and edge costs come from mpcx_expand_batch (one launch expands the popped node together
with the best open nodes and, in a second launch, all of their free children, so most pops hit the cache); the
queue, heuristic and goal test stay on the host (exact float semantics of a_star.py).
"""
from typing import Dict, Iterable, List, Tuple

import numpy as np
import torch

from ._session import context
from .a_star import AStar
from .trajectories import car_trajectory_to_collision_point_trajectories, resample_curve

NodeType = Tuple[float, float, float]


class MotionPrimitiveSearch:
    variant = 'base'
    PREFETCH = 48          # open nodes expanded speculatively per cache miss

    def __init__(self, scenario, car_dimensions, mps: Dict[str, object], margin: float,
                 wh_dist: float = 1.0, wh_theta: float = 2.7, wh_steering: float = 15.0, wh_obstacle: float = 0.0,
                 wh_center: float = 0.0, wc_dist: float = 1.0, wc_steering: float = 5.0, wc_obstacle: float = 0.1,
                 wc_center: float = 0.0, variant: str = None, ctx=None):
        if variant is not None:
            self.variant = variant
        if self.variant not in ('base', 'modified', 'multi_lane', 'roundabout', 'single_lane'):
            raise ValueError('unknown search variant %r' % (self.variant,))
        self._mps = mps
        self._car_dimensions = car_dimensions
        self._points_to_mp_names: Dict[Tuple[NodeType, NodeType], str] = {}
        self._start = scenario.start
        self._goal_area = scenario.goal_area
        self._goal_point = scenario.goal_point
        self._allowed_goal_theta_difference = scenario.allowed_goal_theta_difference
        self._obstacles_hp: List[np.ndarray] = [o.to_convex(margin=margin) for o in scenario.obstacles]
        self._gx, self._gy, self._gtheta = scenario.goal_point
        self._wh = (wh_dist, wh_theta, wh_steering, wh_obstacle, wh_center)
        self._wc = (wc_dist, wc_steering, wc_obstacle, wc_center)
        self._wh_dist, self._wh_theta, self._wh_steering, self._wh_obstacle, self._wh_center = self._wh
        self._wc_dist, self._wc_steering, self._wc_obstacle, self._wc_center = self._wc
        self._a_star: AStar[NodeType] = AStar(neighbor_function=self.neighbor_function)
        self._mp_collision_points: Dict[str, np.ndarray] = self._create_collision_points()
        self.visited_nodes = 0

        # device tables: primitive ids follow the dict's iteration order (the order neighbour_function yields in)
        self._ctx = ctx if ctx is not None else context()
        self._names = list(mps.keys())
        hp_off = np.cumsum([0] + [len(h) for h in self._obstacles_hp]).astype(np.int32)
        hp = np.concatenate(self._obstacles_hp, axis=0) if self._obstacles_hp else np.zeros((0, 3))
        self._model = self._ctx.search_model([self._mp_collision_points[n] for n in self._names],
                                             np.array([mps[n].points[-1] for n in self._names]),
                                             np.array([mps[n].total_length for n in self._names]), hp, hp_off)
        self._cache: Dict[NodeType, List[Tuple[int, NodeType]]] = {}
        self.kernel_launches = 0

    # ------------------------------------------------------------------ set-up (motion_primitive_search.py:35-52)
    def _create_collision_points(self) -> Dict[str, np.ndarray]:
        spacing = self._car_dimensions.radius
        table = {}
        for name, mp in self._mps.items():
            sparse = resample_curve(mp.points.copy(), dl=spacing, keep_last_point=True)
            discs = car_trajectory_to_collision_point_trajectories(sparse, self._car_dimensions)
            table[name] = np.concatenate(discs, axis=0)
        return table

    # ------------------------------------------------------------------ public API
    def run(self, debug=False):
        cost, path = self._a_star.run(self._start, is_goal_function=self.is_goal,
                                      heuristic_function=self.distance_to_goal, debug=debug)
        return cost, path, self.path_to_full_trajectory(path)

    def run_gen(self, debug=False):
        """`run` as a generator for plan_many: yields every node whose expansion is not cached yet; value = (cost, path)"""
        return self._a_star.run_gen(self._start, is_goal_function=self.is_goal, heuristic_function=self.distance_to_goal, debug=debug,
                                    ready_function=lambda node: node in self._cache)

    @property
    def debug_data(self):
        return self._a_star.debug_data

    def is_goal(self, node: NodeType) -> bool:        # motion_primitive_search.py:64-69 (no angle wrapping)
        return bool(self._goal_area.distance_to_point(node[:2]) <= 1e-5
                    and abs(node[2] - self._gtheta) <= self._allowed_goal_theta_difference)

    def calculate_steering_change_cost(self, current_node, next_node, steering_angle_weight: float = 1.0) -> float:
        d = next_node[2] - current_node[2]
        d = (d + np.pi) % (2 * np.pi) - np.pi          # _multi_lane.py:56-76
        return abs(d) * steering_angle_weight

    def calculate_distance_point_to_halfplane(self, point, half_planes: np.ndarray) -> float:
        x0, y0 = point
        return min(abs(a * x0 + b * y0 + c) / (a ** 2 + b ** 2) ** 0.5 for a, b, c in half_planes)

    def distance_to_nearest_obstacle(self, node: NodeType) -> float:
        best = float('inf')
        for hp in self._obstacles_hp:
            d = self.calculate_distance_point_to_halfplane((node[0], node[1]), hp)
            if d < best:
                best = d
        return best

    def distance_to_goal(self, node: NodeType) -> float:
        x, y, theta = node
        if self.variant == 'base':                     # motion_primitive_search.py:71-75
            d_xy = self._goal_area.distance_to_point(node[:2])
            d_th = max(0., abs(theta - self._gtheta) - self._allowed_goal_theta_difference)
            return d_xy + 2.7 * d_th
        gx, gy, gth = self._goal_point
        d_xy = np.sqrt((x - gx) ** 2 + (y - gy) ** 2)
        d_th = min(abs(theta - gth), abs(theta - gth) - self._allowed_goal_theta_difference / 2)
        if self.variant in ('modified', 'roundabout'):  # _modified.py:80-89, _roundabout.py:131-157
            return d_xy + 2.7 * d_th
        if self.variant == 'single_lane':              # _single_lane.py:145-162
            return d_xy + 2.7 * d_th + 15 * self.calculate_steering_change_cost(node, self._goal_point, steering_angle_weight=1.0)
        steer = self.calculate_steering_change_cost(node, self._goal_point, steering_angle_weight=1.0)   # _multi_lane.py:155-181
        obst = 0.0
        centre = 0.0
        if self._wh_obstacle != 0.0:
            d = self.distance_to_nearest_obstacle(node)
            obst = 1 / d if d else float('inf')
        if self._wh_center != 0.0:
            centre = np.sqrt(x ** 2 + y ** 2)
        return (self._wh_dist * d_xy + self._wh_theta * d_th + self._wh_steering * steer
                + self._wh_obstacle * obst + self._wh_center * centre)

    # ------------------------------------------------------------------ device expansion
    def _uncached(self, nodes: List[NodeType]) -> List[NodeType]:
        return [n for n in dict.fromkeys(nodes) if n not in self._cache]

    @staticmethod
    def _node_arrays(nodes: List[NodeType]):
        arr = np.array(nodes, dtype=np.float64).reshape(-1, 3)
        # cos/sin from the host's numpy -- the library the reference's create_2d_transform_mtx calls -- so successor
        # coordinates (hence exact-equality node identity and exact-tie pop order) are bit-identical to the reference
        return arr, np.column_stack([np.cos(arr[:, 2]), np.sin(arr[:, 2])])

    def _expand(self, nodes: List[NodeType]):
        """expand uncached nodes on the GPU; returns the free children that became known"""
        nodes = self._uncached(nodes)
        if not nodes:
            return []
        arr, cs = self._node_arrays(nodes)
        out = self._ctx.expand(self._model, self._ctx.f64(arr), nodes_cs=self._ctx.f64(cs))
        self.kernel_launches += 1
        return self._store(nodes, out['nbr'].cpu().numpy(), out['collide'].cpu().numpy())

    def _store(self, nodes: List[NodeType], nbr: np.ndarray, col: np.ndarray) -> List[NodeType]:
        children = []
        for n, nb, cl in zip(nodes, nbr, col):
            free = [(k, (float(nb[k, 0]), float(nb[k, 1]), float(nb[k, 2]))) for k in range(len(self._names)) if not cl[k]]
            self._cache[n] = free
            children.extend(c for _, c in free)
        return children

    def neighbor_function(self, node: NodeType) -> Iterable[Tuple[float, NodeType]]:
        if node not in self._cache:
            kids = self._expand([node] + self._a_star.peek_open(self.PREFETCH))
            self._expand(kids)                         # one level of look-ahead
        self.visited_nodes += 1
        for k, nb in self._cache[node]:
            name = self._names[k]
            self._points_to_mp_names[node, nb] = name
            length = self._mps[name].total_length
            if self.variant in ('base', 'modified'):
                yield length, nb
                continue
            steer = self.calculate_steering_change_cost(node, nb, steering_angle_weight=1.0)     # _multi_lane.py:226-237
            if self.variant in ('roundabout', 'single_lane'):
                d = self.distance_to_nearest_obstacle(nb)
                obst = 1 / d if d else float('inf')
                if self.variant == 'roundabout':       # _roundabout.py:212 (term order matters for the rounding)
                    yield length + 0.1 * obst + 5 * steer, nb
                else:                                  # _single_lane.py:218
                    yield length + 5 * steer + 0.1 * obst, nb
                continue
            obst = 0.0
            centre = 0.0
            if self._wh_obstacle != 0.0:               # (sic) the reference gates the COST term on the heuristic weight
                d = self.distance_to_nearest_obstacle(nb)
                obst = 1 / d if d else float('inf')
            if self._wc_center != 0.0:
                centre = np.linalg.norm([nb[0], nb[1]])
            yield (self._wc_dist * length + self._wc_steering * steer + self._wc_obstacle * obst
                   + self._wc_center * centre), nb

    # ------------------------------------------------------------------ geometry at a configuration (device transform)
    def _transform(self, points: np.ndarray, configurations: List[NodeType]) -> np.ndarray:
        n = len(configurations)
        pts = self._ctx.f64(np.ascontiguousarray(points, dtype=np.float64))
        nodes = self._ctx.f64(np.array(configurations, dtype=np.float64).reshape(-1, 3))
        off = self._ctx.i32(np.zeros(n)); cnt = self._ctx.i32(np.full(n, len(points)))
        return self._ctx.transform(nodes, off, cnt, pts, len(points)).cpu().numpy()

    def collision_checking_points_at(self, mp_name: str, configuration: NodeType) -> np.ndarray:
        return self._transform(self._mp_collision_points[mp_name], [configuration])[0]

    def motion_primitive_at(self, mp_name: str, configuration: NodeType) -> np.ndarray:
        return self._transform(self._mps[mp_name].points, [configuration])[0]

    def path_to_full_trajectory(self, path: List[NodeType]) -> np.ndarray:
        """all path edges in one launch (motion_primitive_search.py:123-135): each primitive transformed at its
        start pose, last point dropped, concatenated"""
        if len(path) < 2:
            return np.zeros((0, 3))
        names = [self._points_to_mp_names[a, b] for a, b in zip(path[:-1], path[1:])]
        uniq = list(dict.fromkeys(names))
        offs, cur = {}, 0
        for nme in uniq:
            offs[nme] = cur
            cur += len(self._mps[nme].points)
        pts = self._ctx.f64(np.concatenate([self._mps[nme].points for nme in uniq], axis=0))
        cnts = np.array([len(self._mps[nme].points) for nme in names])
        nodes = self._ctx.f64(np.array(path[:-1], dtype=np.float64))
        out = self._ctx.transform(nodes, self._ctx.i32(np.array([offs[nme] for nme in names])), self._ctx.i32(cnts), pts,
                                  int(cnts.max())).cpu().numpy()
        return np.concatenate([out[i, :cnts[i] - 1] for i in range(len(names))], axis=0)


def plan_many(searches: List[MotionPrimitiveSearch], debug=False):
    """Run several independent searches concurrently: every search keeps its own exact queue, heuristic and
    expansion cache on the host -- pop order, costs and paths are those of `search.run()` -- while the nodes all of them are
    waiting for (each popped node + its best open nodes, then all their free children) are expanded by ONE
    mpcx_expand_multi_batch launch per level, whatever the number of searches.  Returns [(cost, path, trajectory), ...]."""
    if not searches:
        return []
    ctx = searches[0]._ctx
    gens = [s.run_gen(debug=debug) for s in searches]
    results = [None] * len(searches)
    waiting = {}

    def advance(i):
        try:
            waiting[i] = next(gens[i])
        except StopIteration as done:
            waiting.pop(i, None)
            results[i] = done.value

    def expand_level(requests):
        """requests: {search index: [nodes]} -> {search index: children}; one launch"""
        idx = [i for i, nodes in requests.items() if nodes]
        if not idx:
            return {}
        off = np.cumsum([0] + [len(requests[i]) for i in idx])
        flat = [n for i in idx for n in requests[i]]
        arr, cs = MotionPrimitiveSearch._node_arrays(flat)
        out = ctx.expand_multi([searches[i]._model for i in idx], off, ctx.f64(arr), nodes_cs=ctx.f64(cs))
        nbr, col = out['nbr'].cpu().numpy(), out['collide'].cpu().numpy()
        kids = {}
        for j, i in enumerate(idx):
            searches[i].kernel_launches += 1
            kids[i] = searches[i]._store(requests[i], nbr[off[j]:off[j + 1]], col[off[j]:off[j + 1]])
        return kids

    for i in range(len(searches)):
        advance(i)
    while waiting:
        level1 = {i: searches[i]._uncached([node] + searches[i]._a_star.peek_open(searches[i].PREFETCH)) for i, node in waiting.items()}
        kids = expand_level(level1)
        expand_level({i: searches[i]._uncached(c) for i, c in kids.items()})        # one level of look-ahead
        for i in list(waiting):
            advance(i)
    out = []
    for s, (cost, path) in zip(searches, results):
        out.append((cost, path, s.path_to_full_trajectory(path)))
    return out
