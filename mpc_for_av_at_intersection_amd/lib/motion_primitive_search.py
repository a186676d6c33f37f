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
import os
from typing import Dict, Iterable, List, Tuple

import numpy as np
import torch

from ._session import context
from .a_star import AStar, AStarDebugData
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
        self._mps_key = hash(tuple((n, np.ascontiguousarray(mps[n].points, dtype=np.float64).tobytes()) for n in self._names))   # same primitives, same key
        hp_off = np.cumsum([0] + [len(h) for h in self._obstacles_hp]).astype(np.int32)
        hp = np.concatenate(self._obstacles_hp, axis=0) if self._obstacles_hp else np.zeros((0, 3))
        self._hp_key = hash(np.ascontiguousarray(hp, dtype=np.float64).tobytes())      # same obstacle rows, same key
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

    def _reference_h(self, nodes: np.ndarray) -> np.ndarray:
        """`distance_to_goal` (every variant but 'base') for an array of nodes with the reference's bits -- what plan_many_device checks the
        device's values against: the squares through Python's `**` (libm pow), the float % through np.mod (the same fmod-and-sign rule);
        everything else is IEEE arithmetic numpy evaluates element by element in the order written, like the interpreter"""
        gx, gy, gth = self._goal_point
        x, y, th = nodes[:, 0], nodes[:, 1], nodes[:, 2]
        d_xy = np.sqrt(_py_square(x - gx) + _py_square(y - gy))
        ad = np.abs(th - gth)
        d_th = np.minimum(ad, ad - self._allowed_goal_theta_difference / 2)
        if self.variant in ('modified', 'roundabout'):
            return d_xy + 2.7 * d_th
        steer = _steering_change_np(th, gth)
        if self.variant == 'single_lane':
            return d_xy + 2.7 * d_th + 15 * steer
        obst = self._dev_check.inv_obstacle_distance(x, y) if self._wh_obstacle != 0.0 else 0.0
        centre = np.sqrt(_py_square(x) + _py_square(y)) if self._wh_center != 0.0 else 0.0
        return (self._wh_dist * d_xy + self._wh_theta * d_th + self._wh_steering * steer
                + self._wh_obstacle * obst + self._wh_center * centre)

    def _reference_edge(self, parents: np.ndarray, children: np.ndarray, prim_ids: np.ndarray) -> np.ndarray:
        """the edge values `neighbor_function` yields for parents[i] -> children[i] through primitive prim_ids[i] ('multi_lane', 'roundabout',
        'single_lane'), with the reference's bits (see _reference_h)"""
        c = self._dev_check
        length = c.lengths[prim_ids]
        steer = _steering_change_np(parents[:, 2], children[:, 2])
        if self.variant in ('roundabout', 'single_lane'):
            obst = c.inv_obstacle_distance(children[:, 0], children[:, 1])
            return length + 0.1 * obst + 5 * steer if self.variant == 'roundabout' else length + 5 * steer + 0.1 * obst
        obst = c.inv_obstacle_distance(children[:, 0], children[:, 1]) if self._wh_obstacle != 0.0 else 0.0     # (sic) gated on the HEURISTIC weight
        centre = 0.0
        if self._wc_center != 0.0:
            # np.linalg.norm([x, y]) goes through BLAS ddot (whose rounding -- fused or not -- is the library's business): evaluated with that
            # very call, once per DISTINCT point (a batch of replicated or overlapping searches repeats its successors many times over)
            keys = np.ascontiguousarray(children[:, :2]).view(np.complex128).ravel()       # (x, y) as one sortable scalar: a 1-D unique, not a row sort
            uniq, inv = np.unique(keys, return_inverse=True)
            centre = np.array([float(np.linalg.norm([px, py])) for px, py in zip(uniq.real.tolist(), uniq.imag.tolist())])[inv.ravel()]
        return self._wc_dist * length + self._wc_steering * steer + self._wc_obstacle * obst + self._wc_center * centre

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


# ------------------------------------------------------------------ device-resident searches (csrc/mpcx_astar.hip)
_CLOSURES = {}

# arguments at which libm pow(x, 2.0) -- Python's x ** 2 -- is NOT the correctly rounded x * x (found by search; about 0.08 % of doubles are)
_POW_WITNESSES = np.array([float.fromhex(h) for h in (
    '-0x1.3280af53a2210p+5', '0x1.ce81dbd32a3f0p+4', '-0x1.c1cd5db74d5e9p+4', '-0x1.04a4662f0b126p+4', '-0x1.3383e88441076p+5',
    '0x1.55a43676965dcp+4', '-0x1.aa91da030b6b1p+5', '-0x1.d61918d4ec37dp+4', '-0x1.df53d17adc2e4p+3', '-0x1.e31e226b77888p+2',
    '0x1.34beb16fc7620p+0', '0x1.c3549947b95f0p+3', '0x1.be484163e1b0cp+5', '-0x1.a0f9ce21b6308p+2', '-0x1.4a0ac419b5fccp+3',
    '-0x1.1e3fb9578dfaap+5', '-0x1.b137254d890a4p+4', '0x1.1a9a606a9e01cp+4', '0x1.1ab2836c40a3ap+5', '0x1.626847f670c0ep+5',
    '-0x1.f8bb263499278p+3', '-0x1.f772a86155e6ap+4', '-0x1.a17028dfef434p+3', '0x1.8b17a6f896034p+5')])
_POW_VECTOR_OK = None


def _py_square(x: np.ndarray) -> np.ndarray:
    """x ** 2 as Python floats compute it (libm pow: NOT always the correctly rounded x * x), for an array.  numpy.float_power goes
    through the same pow on this image; that is CHECKED once per process on a fixed list of arguments at which pow(x, 2) and x * x
    differ (a float_power with a square fast path, or another pow, cannot pass it); should it fail, the element-by-element loop runs."""
    global _POW_VECTOR_OK
    if _POW_VECTOR_OK is None:
        want = np.array([v ** 2 for v in _POW_WITNESSES.tolist()])
        _POW_VECTOR_OK = bool(np.array_equal(np.float_power(_POW_WITNESSES, 2), want) and not np.array_equal(_POW_WITNESSES * _POW_WITNESSES, want))
    x = np.asarray(x, dtype=np.float64)
    if _POW_VECTOR_OK:
        return np.float_power(x, 2)
    return np.array([v ** 2 for v in x.ravel().tolist()]).reshape(x.shape)


def _steering_change_np(current_theta, next_theta):
    """calculate_steering_change_cost(current, next, 1.0) for arrays: np.mod on float64 IS Python's float % (fmod, sign of the divisor)"""
    d = next_theta - current_theta
    return np.abs(np.mod(d + np.pi, 2 * np.pi) - np.pi)


def _check_key(s: 'MotionPrimitiveSearch'):
    """searches with the same key evaluate h and the edge value with the same expressions and constants (one vectorised check per key);
    a search whose reference expressions were replaced on the instance (tests do that) is a group of its own"""
    patched = id(s) if ('_reference_h' in s.__dict__ or '_reference_edge' in s.__dict__) else None
    return (s.variant, s._goal_point, s._allowed_goal_theta_difference, s._wh, s._wc, s._mps_key, s._hp_key, patched)


class _DeviceCheck:
    """What plan_many_device needs per group of like searches besides their models: the reference's own expressions for h and the edge
    value over arrays live on the representative search (`_reference_h`, `_reference_edge`); this object holds their tables."""

    def __init__(self, s: 'MotionPrimitiveSearch'):
        from .. import _lib
        self.s = s
        self.variant = _lib.ASTAR_VARIANTS[s.variant]
        self.needs_obst = s.variant in ('roundabout', 'single_lane') or (s.variant == 'multi_lane' and s._wh_obstacle != 0.0)
        self.log_all = s.variant in ('multi_lane', 'roundabout', 'single_lane')
        self.lengths = np.array([float(s._mps[n].total_length) for n in s._names])
        self.rows = self.norm = self._norm_dev = None
        if self.needs_obst:
            self.rows = np.concatenate(s._obstacles_hp, axis=0).astype(np.float64) if s._obstacles_hp else np.zeros((0, 3))
            # (a**2 + b**2)**0.5 exactly as calculate_distance_point_to_halfplane evaluates it: numpy float64 scalars, `**` = libm pow
            self.norm = np.array([float((r[0] ** 2 + r[1] ** 2) ** 0.5) for r in self.rows], dtype=np.float64)

    def norm_dev(self):
        if self._norm_dev is None and self.needs_obst and len(self.norm):
            self._norm_dev = self.s._ctx.f64(self.norm)
        return self._norm_dev

    def inv_obstacle_distance(self, x, y):
        """1 / distance_to_nearest_obstacle (inf where the distance is 0) at points (x, y), element-wise IEEE arithmetic in the reference's order:
        abs((a * x + b * y) + c) / norm per half-plane row, minimum over the rows.  points x rows is tens of millions of elements for a batch
        of searches: evaluated in cache-sized chunks of points, in place, on a few threads (numpy releases the GIL inside its loops)."""
        r = self.rows
        n = len(x)
        if r.shape[0] == 0:
            return np.zeros(n)                               # min over nothing = inf, 1 / inf = 0
        x = np.ascontiguousarray(x, dtype=np.float64); y = np.ascontiguousarray(y, dtype=np.float64)
        a, b, c, nrm = r[None, :, 0], r[None, :, 1], r[None, :, 2], self.norm[None, :]
        d = np.empty(n)

        def work(lo):
            hi = min(lo + 2048, n)
            t = a * x[lo:hi, None]
            t += b * y[lo:hi, None]
            t += c
            np.abs(t, out=t)
            t /= nrm
            d[lo:hi] = t.min(axis=1)
        starts = range(0, n, 2048)
        if n > 8192:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
                list(pool.map(work, starts))
        else:
            for lo in starts:
                work(lo)
        with np.errstate(divide='ignore'):
            return np.where(d != 0.0, 1.0 / d, np.inf)


def paths_to_full_trajectories(searches: List['MotionPrimitiveSearch'], paths: List[List[NodeType]], prim_ids=None) -> List[np.ndarray]:
    """`path_to_full_trajectory` of many searches that share their primitives, all edges in ONE transform launch and one copy back.
    prim_ids[i][j] (optional) = primitive id of the edge that ENDS in paths[i][j] (entry 0 unused): saves the dictionary look-ups."""
    if not searches:
        return []
    s0 = searches[0]
    ctx = s0._ctx
    names0 = list(s0._mps)
    n_pts = np.array([len(s0._mps[nme].points) for nme in names0])
    p_off = np.concatenate([[0], np.cumsum(n_pts)])[:-1]
    ids, nodes, n_edges = [], [], []
    for i, (s, path) in enumerate(zip(searches, paths)):
        if prim_ids is not None:
            k = np.asarray(prim_ids[i][1:], dtype=np.int64)
        else:
            k = np.array([names0.index(s._points_to_mp_names[a, b]) for a, b in zip(path[:-1], path[1:])], dtype=np.int64)
        ids.append(k); n_edges.append(len(k))
        if len(k):
            nodes.append(np.asarray(path[:-1], dtype=np.float64).reshape(-1, 3))
    if not sum(n_edges):
        return [np.zeros((0, 3)) for _ in searches]
    ids = np.concatenate(ids)
    cnts = n_pts[ids]
    pts = ctx.f64(np.concatenate([s0._mps[nme].points for nme in names0], axis=0))
    out = ctx.transform(ctx.f64(np.concatenate(nodes, axis=0)), ctx.i32(p_off[ids]), ctx.i32(cnts), pts, int(cnts.max()))
    # each primitive without its last point (motion_primitive_search.py:123-135), packed on the device, one copy back
    keep = torch.arange(out.shape[1], device=out.device)[None, :] < (ctx.i32(cnts).to(torch.int64) - 1)[:, None]
    flat = out[keep].cpu().numpy()
    per_edge = np.concatenate([[0], np.cumsum(cnts - 1)])
    e_off = np.concatenate([[0], np.cumsum(n_edges)])
    return [flat[per_edge[e_off[i]]:per_edge[e_off[i + 1]]] for i in range(len(searches))]


def _edges_to_trajectories(s0: 'MotionPrimitiveSearch', edges) -> List[np.ndarray]:
    """path_to_full_trajectory for many paths over the primitives of s0 from arrays: edges[i] = (start nodes (e, 3), primitive ids (e,)) of
    path i.  One transform launch, the primitives' last points dropped on the device, one copy back."""
    ctx = s0._ctx
    names0 = s0._names
    n_pts = np.array([len(s0._mps[nme].points) for nme in names0])
    p_off = np.concatenate([[0], np.cumsum(n_pts)])[:-1]
    n_edges = np.array([len(e[1]) for e in edges])
    if not n_edges.sum():
        return [np.zeros((0, 3)) for _ in edges]
    ids = np.concatenate([e[1] for e in edges]).astype(np.int64)
    nodes = np.concatenate([e[0] for e in edges], axis=0)
    cnts = n_pts[ids]
    pts = ctx.f64(np.concatenate([s0._mps[nme].points for nme in names0], axis=0))
    cnt_t = ctx.i32(cnts)
    out = ctx.transform(ctx.f64(nodes), ctx.i32(p_off[ids]), cnt_t, pts, int(cnts.max()))
    keep = torch.arange(out.shape[1], device=out.device)[None, :] < (cnt_t.to(torch.int64) - 1)[:, None]
    flat = out[keep].cpu().numpy()
    per_edge = np.concatenate([[0], np.cumsum(cnts - 1)])
    bounds = per_edge[np.concatenate([[0], np.cumsum(n_edges)])].tolist()
    return [flat[bounds[i]:bounds[i + 1]] for i in range(len(edges))]


def heading_closure(start_thetas, dthetas, depth: int) -> np.ndarray:
    """Every heading a search can reach within `depth` primitives: the closure of the start headings under theta -> normalize_angle(dtheta +
    theta) (transform_2d_pts adds the primitive's end heading to the node's, linalg.py:48-50; maths.py:4-10), evaluated with numpy float64
    arithmetic, which is the reference's (np.mod has Python's float % semantics).  Sorted ascending."""
    key = (tuple(sorted(set(float(t) for t in start_thetas))), tuple(float(d) for d in dthetas), int(depth))
    if key in _CLOSURES:
        return _CLOSURES[key]
    tau = 2 * np.pi
    seen = np.unique(np.asarray(start_thetas, dtype=np.float64))
    frontier = seen
    dth = np.asarray(dthetas, dtype=np.float64)
    for _ in range(depth):
        nxt = np.mod(dth[None, :] + frontier[:, None], tau).ravel()
        nxt = np.where(nxt >= np.pi, nxt - tau, nxt)
        nxt = np.unique(nxt)
        frontier = np.setdiff1d(nxt, seen, assume_unique=True)
        if frontier.size == 0:
            break
        seen = np.union1d(seen, frontier)
    if len(_CLOSURES) > 8:
        _CLOSURES.clear()
    _CLOSURES[key] = seen
    return seen


def plan_many_device(searches: List['MotionPrimitiveSearch'], max_expansions: int = 4096, closure_depth: int = 16, max_rounds: int = 12,
                     debug=False, first_expansions: int = 512, path_cap: int = 64):
    """Run independent searches -- ALL FIVE variants -- with open list, closed set and successor generation RESIDENT ON THE DEVICE
    (mpcx_astar_batch: one wavefront per search, no host work between expansions) and the reference's exact pop order.  The host's part
    is what must carry the reference's bits and cannot be evaluated on the device: the cos / sin table (numpy, over the closure of the
    start headings under each primitive set's heading changes) before the launch, and afterwards a check of every heuristic and edge
    value the searches used against the reference's own expressions (`**` is libm pow: one ulp from x * x for ~0.08 % of arguments;
    np.linalg.norm is BLAS): values that differ go into that search's override slice and the search runs again -- typically one extra
    launch for a few of the searches.  Capacities grow on demand: the first launch holds `first_expansions` expansions per search, a
    search that needs more runs again with four times as many (up to max_expansions, then RuntimeError), a path longer than path_cap
    nodes runs again with twice the room.  The goal area must be a box.
    Returns ([(cost, path, trajectory), ...], info) with info = dict(launches, rounds per search, overrides, expansions, timings)."""
    from .. import _lib
    import time as _time
    if not searches:
        return [], dict(launches=0)
    ctx = searches[0]._ctx
    t_start = _time.perf_counter()
    n = len(searches)
    for s in searches:
        if not hasattr(s._goal_area, 'xy1'):
            raise NotImplementedError('plan_many_device: the goal area must be a box')
    gkey = [_check_key(s) for s in searches]
    checks = {}
    for s, k in zip(searches, gkey):
        if k not in checks:
            checks[k] = s._dev_check = _DeviceCheck(s)
    chk = [checks[k] for k in gkey]
    # heading table: the union over the primitive sets present of the closure of THEIR searches' start headings
    prim_groups = {}
    for i, s in enumerate(searches):
        prim_groups.setdefault(s._mps_key, []).append(i)
    dth_of = {k: [float(searches[idx[0]]._mps[nm].points[-1][2]) for nm in searches[idx[0]]._names] for k, idx in prim_groups.items()}
    theta_tab = None
    for k, idx in prim_groups.items():
        cl = heading_closure([searches[i]._start[2] for i in idx], dth_of[k], closure_depth)
        theta_tab = cl if theta_tab is None else np.union1d(theta_tab, cl)
    # the kernel's struct rows, column by column
    rows = np.zeros(n, dtype=_lib.ASTAR_SEARCH_DTYPE)
    rows['start'] = [s._start for s in searches]
    rows['goal_box'] = [(*s._goal_area.xy1, *s._goal_area.xy2) for s in searches]
    rows['goal_point'] = [s._goal_point for s in searches]
    rows['allowed_dtheta'] = [s._allowed_goal_theta_difference for s in searches]
    rows['wh'] = [s._wh for s in searches]; rows['wc'] = [s._wc for s in searches]
    rows['variant'] = [c.variant for c in chk]
    norm_ptr = {k: (c.norm_dev().data_ptr() if c.norm_dev() is not None else 0) for k, c in checks.items()}
    rows['hp_norm'] = [norm_ptr[k] for k in gkey]
    overrides = [dict() for _ in range(n)]      # per search: (x, y, theta, kind) -> value with the reference's bits
    level = np.zeros(n, dtype=np.int64)         # capacity level of every search: first_expansions * 4**level expansions
    results = [None] * n
    edge_src = [None] * n       # per finished search: (start node of every path edge, primitive id of every path edge)
    todo = list(range(n))
    info = dict(launches=0, rounds=[0] * n, overrides=0, expansions=[0] * n, table_headings=int(theta_tab.size),
                t_closure=_time.perf_counter() - t_start, t_device=0.0, t_check=0.0, t_results=0.0)
    cs_t = cs_v = None
    for _ in range(max_rounds):
        if not todo:
            break
        if cs_t is None:
            # the table on the device, kept per context and closure (2.1 M headings = 50 MB for the stock start poses: cos / sin and the
            # upload cost more than the searches)
            tkey = (theta_tab.size, float(theta_tab[0]), float(theta_tab[-1]), float(theta_tab[::257].sum()))
            tabs = ctx.__dict__.setdefault('_astar_heading_tables', {})      # the tensors live and die with their context
            if tkey not in tabs:
                if len(tabs) > 4:
                    tabs.clear()
                tabs[tkey] = (ctx.f64(theta_tab), ctx.f64(np.column_stack([np.cos(theta_tab), np.sin(theta_tab)])))
            cs_t, cs_v = tabs[tkey]
        m = len(todo)
        E = int(min(max_expansions, first_expansions * 4 ** int(level[todo].max())))
        top = E >= max_expansions
        P = max(searches[i]._model.n_prim for i in todo)
        any_all = any(chk[i].log_all for i in todo)
        # successors per expansion: at most P; searches that only log their pushes (base / modified) stay far below that
        push_cap = heap_cap = None if (top or any_all) else 5 * E + 64
        sub = rows[todo].copy()
        sub['max_expansions'] = E
        ovk = ovv = None
        cnt = np.array([len(overrides[i]) for i in todo], dtype=np.int64)
        if cnt.sum():
            sub['ov_off'] = np.concatenate([[0], np.cumsum(cnt)])[:-1]; sub['ov_cnt'] = cnt
            keys, vals = [], []
            for i in todo:
                ks = sorted(overrides[i])
                keys.extend(ks); vals.extend(overrides[i][k] for k in ks)
            ovk, ovv = ctx.f64(np.array(keys, dtype=np.float64).reshape(-1, 4)), ctx.f64(np.array(vals, dtype=np.float64))
        t0 = _time.perf_counter()
        out = ctx.astar_batch([searches[i]._model for i in todo], sub, cs_t, cs_v, ovk, ovv, max_expansions=E, path_cap=path_cap,
                              heap_cap=heap_cap, push_cap=push_cap)
        ints = torch.stack([out['status'], out['n_exp'], out['n_push'], out['path_len']]).cpu().numpy()       # (the copy waits for the kernel)
        info['t_device'] += _time.perf_counter() - t0
        t0 = _time.perf_counter()
        info['launches'] += 1
        status, n_exp, n_push, path_len = ints
        flts = torch.stack([out['cost'], out['miss']]).cpu().numpy()
        costs, misses = flts
        done = (status == _lib.ASTAR_FOUND) | (status == _lib.ASTAR_EXHAUSTED)
        # the successor logs of the searches that ran to their end, in one gather and one copy (variable-length rows)
        need_log = debug or any_all
        pl_t = out['push_log']
        cnt_t = torch.where(torch.as_tensor(done, device=pl_t.device), out['n_push'], torch.zeros_like(out['n_push'])).to(torch.int64)
        keep = torch.arange(pl_t.shape[1], device=pl_t.device)[None, :] < cnt_t[:, None]
        push_all = (pl_t if any_all else pl_t[:, :, :4])[keep].cpu().numpy()
        push_off = np.concatenate([[0], np.cumsum(np.where(done, n_push, 0))]).astype(np.int64)
        if need_log:
            lg_t = out['log']
            ecnt = torch.where(torch.as_tensor(done, device=lg_t.device), out['n_exp'], torch.zeros_like(out['n_exp'])).to(torch.int64)
            log_all_rows = lg_t[torch.arange(lg_t.shape[1], device=lg_t.device)[None, :] < ecnt[:, None]].cpu().numpy()
            log_off = np.concatenate([[0], np.cumsum(np.where(done, n_exp, 0))]).astype(np.int64)
        again, new_thetas, grow_path = [], [], False
        round_level = int(level[todo].max())
        # ---- the values the searches used, against the reference's arithmetic: one vectorised evaluation per group of like searches
        bad = np.zeros(m, dtype=bool)
        groups = {}
        for j, i in enumerate(todo):
            if done[j] and chk[i].variant != _lib.ASTAR_BASE and n_push[j]:
                groups.setdefault(gkey[i], []).append(j)
        for js in groups.values():
            c = chk[todo[js[0]]]
            s = c.s
            seg = [push_all[push_off[j]:push_off[j + 1]] for j in js]
            owner = np.repeat(np.arange(len(js)), [len(x) for x in seg])
            pl = np.concatenate(seg, axis=0)
            pushed = ~np.isnan(pl[:, 3])
            ref_h = s._reference_h(pl[pushed, :3])
            wrong = np.nonzero(ref_h != pl[pushed, 3])[0]
            rows_p = np.nonzero(pushed)[0]
            for w in wrong:
                r = rows_p[w]
                j = js[owner[r]]
                overrides[todo[j]][(float(pl[r, 0]), float(pl[r, 1]), float(pl[r, 2]), -1.0)] = float(ref_h[w])
                bad[j] = True
            if c.log_all:
                par = np.concatenate([log_all_rows[log_off[j]:log_off[j + 1]][pl_j[:, 5].astype(np.int64), :3] for j, pl_j in zip(js, seg)], axis=0)
                kk = pl[:, 6].astype(np.int64)
                ref_e = s._reference_edge(par, pl[:, :3], kk)
                for r in np.nonzero(ref_e != pl[:, 4])[0]:
                    j = js[owner[r]]
                    overrides[todo[j]][(float(par[r, 0]), float(par[r, 1]), float(par[r, 2]), float(kk[r]))] = float(ref_e[r])
                    bad[j] = True
        info['t_check'] += _time.perf_counter() - t0
        t0 = _time.perf_counter()
        pmax = int(path_len[done].max()) if done.any() else 0
        path_all = out['path'][:, :pmax].cpu().numpy(); prim_all = out['path_prim'][:, :pmax].cpu().numpy()
        info_rounds, info_exp = info['rounds'], info['expansions']
        final = []
        for j, i in enumerate(todo):
            info_rounds[i] += 1
            info_exp[i] = int(n_exp[j])
            st = status[j]
            if st == _lib.ASTAR_MISS:
                new_thetas.append((searches[i]._mps_key, float(misses[j])))
                again.append(i)
            elif st == _lib.ASTAR_CAPACITY:
                if top:
                    raise RuntimeError('plan_many_device: search %d exceeds %d expansions' % (i, max_expansions))
                level[i] = round_level + 1
                again.append(i)
            elif st == _lib.ASTAR_PATH_CAPACITY:
                grow_path = True
                again.append(i)
            elif bad[j]:
                again.append(i)
            elif st == _lib.ASTAR_EXHAUSTED:
                raise Exception("No solution found.")
            else:
                final.append(j)
        if final:
            # every finished path start -> goal, flattened: one fancy index, one tolist for all searches
            fj = np.array(final)
            L = path_len[fj].astype(np.int64)
            owner = np.repeat(np.arange(len(fj)), L)
            first = np.concatenate([[0], np.cumsum(L)])
            pos = np.arange(first[-1]) - first[:-1][owner]
            src = L[owner] - 1 - pos
            flat_nodes = path_all[fj[owner], src]
            flat_prims = prim_all[fj[owner], src]
            tuples = list(map(tuple, flat_nodes.tolist()))
            prim_list = flat_prims.tolist()
            for q, j in enumerate(final):
                i = todo[j]
                s = searches[i]
                a, b = int(first[q]), int(first[q + 1])
                path = tuples[a:b]
                assert path[0] == s._start or path[0] == tuple(float(v) for v in s._start), 'plan_many_device: the path does not start at the start node'
                names = s._names
                s._points_to_mp_names.update(zip(zip(path[:-1], path[1:]), [names[k] for k in prim_list[a + 1:b]]))
                s.visited_nodes = int(n_exp[j])
                if debug:
                    lg = log_all_rows[log_off[j]:log_off[j + 1]].tolist()
                    s._a_star._debug_data = [AStarDebugData(g=r[3], h=r[4], node=tuple(r[0:3]), predecessor=tuple(r[5:8])) for r in lg]
                results[i] = (float(costs[j]), path)
                edge_src[i] = (flat_nodes[a:b - 1], flat_prims[a + 1:b])
        if new_thetas:      # headings beyond the closure: add them and what is reachable from them
            for k in {k for k, _ in new_thetas}:
                theta_tab = np.union1d(theta_tab, heading_closure([t for kk, t in new_thetas if kk == k], dth_of[k], 4))
            cs_t = None
        if grow_path:
            path_cap *= 2
        info['t_results'] += _time.perf_counter() - t0
        info['overrides'] = sum(len(o) for o in overrides)
        todo = again
    if todo:
        raise RuntimeError('plan_many_device: %d searches did not settle in %d rounds' % (len(todo), max_rounds))
    t0 = _time.perf_counter()
    trajs = [None] * n
    for idx in prim_groups.values():        # one transform launch per set of primitives
        for i, t in zip(idx, _edges_to_trajectories(searches[idx[0]], [edge_src[i] for i in idx])):
            trajs[i] = t
    info['t_results'] += _time.perf_counter() - t0
    return [(c, p, t) for (c, p), t in zip(results, trajs)], info
