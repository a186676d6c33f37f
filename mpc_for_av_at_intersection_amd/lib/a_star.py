"""Best-first search with the reference's queue semantics (reference: main/lib/a_star.py:17-78).

The search itself is inherently sequential and stays on the host; what runs on the GPU is the successor
generation behind `neighbor_function` (mpcx_expand_batch). To let that run in batches this class exposes the open
list through `peek_open(k)`: a batching neighbour function can look at the k best open nodes and expand them
speculatively -- expansion is a pure function of the node, so the pop order (and hence the result) is unchanged.

Queue contract (a_star.py:34,43-49,73-76): heap entries are tuples (g+h, g, node, predecessor) compared
lexicographically; a popped node is skipped if it was closed with g >= the stored g; a neighbour is pushed only
when unseen or reached with a strictly smaller g; the goal test happens at pop time.
"""
import heapq
from dataclasses import dataclass
from typing import Callable, Dict, Generic, Hashable, Iterable, List, Tuple, TypeVar

TNode = TypeVar("TNode", bound=Hashable)


@dataclass
class AStarDebugData(Generic[TNode]):
    g: float
    h: float
    node: TNode
    predecessor: TNode


class AStar(Generic[TNode]):
    def __init__(self, neighbor_function: Callable[[TNode], Iterable[Tuple[float, TNode]]]):
        self.neighbor_function = neighbor_function
        self._debug_data: List[AStarDebugData[TNode]] = []
        self._open: List[Tuple[float, float, TNode, TNode]] = []
        self._closed: Dict[TNode, Tuple[float, TNode]] = {}

    @property
    def debug_data(self):
        return self._debug_data

    def peek_open(self, k: int) -> List[TNode]:
        """The (up to) k best nodes still on the open list that are not closed yet, best first."""
        out = []
        for f, g, node, _ in heapq.nsmallest(k, self._open):
            seen = self._closed.get(node)
            if seen is None or g < seen[0]:
                out.append(node)
        return out

    def run(self, start: TNode, is_goal_function: Callable[[TNode], bool],
            heuristic_function: Callable[[TNode], float], debug=False) -> Tuple[float, List[TNode]]:
        gen = self.run_gen(start, is_goal_function, heuristic_function, debug=debug)
        try:
            while True:
                next(gen)              # no ready_function: never yields
        except StopIteration as done:
            return done.value

    def run_gen(self, start: TNode, is_goal_function: Callable[[TNode], bool], heuristic_function: Callable[[TNode], float],
                debug=False, ready_function: Callable[[TNode], bool] = None):
        """The same search as a generator: before a node is expanded it is yielded if `ready_function(node)` is false, so a driver
        can run many searches in lock-step and expand the nodes they wait for in one batch.  Returns (g, path) as the
        generator's value.  Pop order, tie-breaking and the debug log are those of `run` (this IS `run`)."""
        self._open = [(0, 0, start, start)]
        self._closed = {}
        if debug:
            self._debug_data = []
        open_, closed = self._open, self._closed
        while open_:
            f, g, node, parent = heapq.heappop(open_)
            known = closed.get(node)
            if known is not None and g >= known[0]:
                continue
            if debug:
                self._debug_data.append(AStarDebugData(g=g, h=f - g, node=node, predecessor=parent))
            closed[node] = (g, parent)
            if is_goal_function(node):
                chain = [node]
                while node != start:
                    chain.append(parent)
                    node, parent = parent, closed[parent][1]
                chain.reverse()
                return g, chain
            if ready_function is not None and not ready_function(node):
                yield node
            for edge_cost, nb in self.neighbor_function(node):
                nb_g = g + edge_cost
                known = closed.get(nb)
                if known is None or nb_g < known[0]:
                    heapq.heappush(open_, (nb_g + heuristic_function(nb), nb_g, nb, node))
        raise Exception("No solution found.")
