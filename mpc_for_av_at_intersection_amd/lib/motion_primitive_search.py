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
        """`distance_to_goal` of the 'modified' variant for an array of nodes with the reference's bits (what plan_many_device checks
        the device's values against): the squares through Python's `**`, the rest is IEEE arithmetic numpy evaluates identically"""
        gx, gy, gth = self._goal_point
        ex, ey = nodes[:, 0] - gx, nodes[:, 1] - gy
        sq = _py_square(ex) + _py_square(ey)
        ad = np.abs(nodes[:, 2] - gth)
        return np.sqrt(sq) + 2.7 * np.minimum(ad, ad - self._allowed_goal_theta_difference / 2)

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


def _py_square(x: np.ndarray) -> np.ndarray:
    """x ** 2 as Python floats compute it (libm pow: NOT always the correctly rounded x * x), for an array.  numpy.float_power goes
    through the same pow; the first elements are checked against the interpreter every time, and should the two ever part (another
    numpy, another libm) the element-by-element loop takes over."""
    out = np.float_power(x, 2)
    k = min(len(x), 64)
    if k and not np.array_equal(out[:k], np.array([v ** 2 for v in x[:k].tolist()])):
        return np.array([v ** 2 for v in x.tolist()])
    return out


def paths_to_full_trajectories(searches: List['MotionPrimitiveSearch'], paths: List[List[NodeType]]) -> List[np.ndarray]:
    """`path_to_full_trajectory` of many searches that share their primitives, all edges in ONE transform launch and one copy back"""
    if not searches:
        return []
    s0 = searches[0]
    ctx = s0._ctx
    names0 = list(s0._mps)
    offs, cur = {}, 0
    for nme in names0:
        offs[nme] = cur
        cur += len(s0._mps[nme].points)
    edges_off, edges_cnt, edge_nodes, spans = [], [], [], []
    for s, path in zip(searches, paths):
        first = len(edges_off)
        for a, b in zip(path[:-1], path[1:]):
            nme = s._points_to_mp_names[a, b]
            edges_off.append(offs[nme]); edges_cnt.append(len(s0._mps[nme].points)); edge_nodes.append(a)
        spans.append((first, len(edges_off)))
    if not edges_off:
        return [np.zeros((0, 3)) for _ in searches]
    pts = ctx.f64(np.concatenate([s0._mps[nme].points for nme in names0], axis=0))
    cnts = np.array(edges_cnt)
    out = ctx.transform(ctx.f64(np.array(edge_nodes, dtype=np.float64)), ctx.i32(np.array(edges_off)), ctx.i32(cnts), pts, int(cnts.max())).cpu().numpy()
    res = []
    for a, b in spans:
        res.append(np.concatenate([out[i, :cnts[i] - 1] for i in range(a, b)], axis=0) if b > a else np.zeros((0, 3)))
    return res


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


def plan_many_device(searches: List['MotionPrimitiveSearch'], max_expansions: int = 4096, closure_depth: int = 16, max_rounds: int = 6, debug=False):
    """Run independent searches with open list, closed set and successor generation RESIDENT ON THE DEVICE (mpcx_astar_batch: one
    wavefront per search, no host work between expansions) and the reference's exact pop order.  The host's part is what must carry the
    reference's bits and cannot be evaluated on the device: the cos / sin table (numpy, over the closure of the start headings) before
    the launch, and afterwards a check of every heuristic value the searches used against Python-float arithmetic (`**` is libm pow:
    one ulp from x * x for ~0.08 % of arguments): values that differ go into an override table and the searches concerned run again --
    typically one extra launch for a few of the searches.  Variants 'base' and 'modified' with box goal areas; anything else raises.
    Returns ([(cost, path, trajectory), ...], info) with info = dict(launches, rounds per search, overrides, expansions)."""
    from .. import _lib
    if not searches:
        return [], dict(launches=0)
    ctx = searches[0]._ctx
    for s in searches:
        if s.variant not in ('base', 'modified'):
            raise NotImplementedError("plan_many_device: variant %r (only 'base' and 'modified' run on the device)" % (s.variant,))
        if not hasattr(s._goal_area, 'xy1'):
            raise NotImplementedError('plan_many_device: the goal area must be a box')
    import time as _time
    t_start = _time.perf_counter()
    dth = [float(searches[0]._mps[n].points[-1][2]) for n in searches[0]._names]
    theta_tab = heading_closure([s._start[2] for s in searches], dth, closure_depth)
    specs = [dict(start=s._start, goal_box=(*s._goal_area.xy1, *s._goal_area.xy2), goal_point=s._goal_point,
                  allowed_dtheta=s._allowed_goal_theta_difference, variant=_lib.ASTAR_BASE if s.variant == 'base' else _lib.ASTAR_MODIFIED)
             for s in searches]
    overrides = {}                    # node -> h with the reference's bits
    results = [None] * len(searches)
    todo = list(range(len(searches)))
    info = dict(launches=0, rounds=[0] * len(searches), overrides=0, expansions=[0] * len(searches), table_headings=int(theta_tab.size),
                t_closure=_time.perf_counter() - t_start, t_device=0.0, t_check=0.0)
    cs_t = cs_v = None
    for _ in range(max_rounds):
        if not todo:
            break
        if cs_t is None:
            # the table on the device, kept per context and closure (2.1 M headings = 50 MB for the stock start poses: cos / sin and the
            # upload cost more than the searches)
            tkey = (theta_tab.size, float(theta_tab[0]), float(theta_tab[-1]), float(theta_tab.sum()))
            tabs = ctx.__dict__.setdefault('_astar_heading_tables', {})      # the tensors live and die with their context
            if tkey not in tabs:
                if len(tabs) > 4:
                    tabs.clear()
                tabs[tkey] = (ctx.f64(theta_tab), ctx.f64(np.column_stack([np.cos(theta_tab), np.sin(theta_tab)])))
            cs_t, cs_v = tabs[tkey]
        hov_n = hov_h = None
        if overrides:
            keys = sorted(overrides)
            hov_n, hov_h = ctx.f64(np.array(keys, dtype=np.float64).reshape(-1, 3)), ctx.f64(np.array([overrides[k] for k in keys]))
        t0 = _time.perf_counter()
        out = ctx.astar_batch([searches[i]._model for i in todo], [specs[i] for i in todo], cs_t, cs_v, hov_n, hov_h, max_expansions=max_expansions)
        ctx.synchronize()
        info['t_device'] += _time.perf_counter() - t0
        t0 = _time.perf_counter()
        info['launches'] += 1
        status = out['status'].cpu().numpy(); n_exp = out['n_exp'].cpu().numpy(); n_push = out['n_push'].cpu().numpy()
        path_len = out['path_len'].cpu().numpy(); costs = out['cost'].cpu().numpy(); misses = out['miss'].cpu().numpy()
        # the pushes of all searches in one gather and one copy (variable-length rows), paths and primitive ids likewise
        pl_t = out['push_log']
        keep = torch.arange(pl_t.shape[1], device=pl_t.device)[None, :] < out['n_push'][:, None].to(torch.int64)
        push_all = pl_t[keep].cpu().numpy()
        push_off = np.concatenate([[0], np.cumsum(n_push)]).astype(np.int64)
        pmax = int(path_len.max()) if len(path_len) else 0
        path_all = out['path'][:, :pmax].cpu().numpy(); prim_all = out['path_prim'][:, :pmax].cpu().numpy()
        again, new_thetas = [], []
        for j, i in enumerate(todo):
            s = searches[i]
            info['rounds'][i] += 1
            info['expansions'][i] = int(n_exp[j])
            if status[j] == _lib.ASTAR_MISS:
                new_thetas.append(float(misses[j]))
                again.append(i)
                continue
            if status[j] == _lib.ASTAR_CAPACITY:
                raise RuntimeError('plan_many_device: search %d exceeds %d expansions' % (i, max_expansions))
            # the heuristic values the search used, against the reference's arithmetic (Python floats)
            pl = push_all[push_off[j]:push_off[j + 1]]
            bad = 0
            if s.variant != 'base' and len(pl):
                # motion_primitive_search_modified.py:80-89 with Python-float squares (`**` = libm pow); everything else is IEEE arithmetic that
                # numpy evaluates identically, so only the squares are taken element by element
                ref = s._reference_h(pl[:, :3])
                for r in np.nonzero(ref != pl[:, 3])[0]:
                    overrides[(float(pl[r, 0]), float(pl[r, 1]), float(pl[r, 2]))] = float(ref[r])
                    bad += 1
            if bad:
                again.append(i)
                continue
            if status[j] == _lib.ASTAR_EXHAUSTED:
                raise Exception("No solution found.")
            n = int(path_len[j])
            nodes = path_all[j, :n][::-1]
            prims = prim_all[j, :n][::-1]
            path = [tuple(float(v) for v in p) for p in nodes]
            for a, b, k in zip(path[:-1], path[1:], prims[1:]):
                s._points_to_mp_names[a, b] = s._names[int(k)]
            s.visited_nodes = int(n_exp[j])
            if debug:
                lg = out['log'][j, :int(n_exp[j])].cpu().numpy()
                s._a_star._debug_data = [AStarDebugData(g=float(r[3]), h=float(r[4]), node=tuple(map(float, r[0:3])), predecessor=tuple(map(float, r[5:8]))) for r in lg]
            results[i] = (float(costs[j]), path)
        if new_thetas:      # headings beyond the closure: add them and what is reachable from them
            theta_tab = np.union1d(theta_tab, heading_closure(new_thetas, dth, 4))
            cs_t = None
        info['t_check'] += _time.perf_counter() - t0
        info['overrides'] = len(overrides)
        todo = again
    if todo:
        raise RuntimeError('plan_many_device: %d searches did not settle in %d rounds' % (len(todo), max_rounds))
    trajs = [None] * len(searches)
    groups = {}
    for i, s in enumerate(searches):        # one transform launch per set of primitives
        groups.setdefault(s._mps_key, []).append(i)
    for idx in groups.values():
        for i, t in zip(idx, paths_to_full_trajectories([searches[i] for i in idx], [results[i][1] for i in idx])):
            trajs[i] = t
    return [(c, p, t) for (c, p), t in zip(results, trajs)], info
