"""main/lib/motion_primitive_search_single_lane.py of the reference: same class, `single_lane` cost / heuristic terms."""
from .motion_primitive_search import MotionPrimitiveSearch as _Base, NodeType  # noqa: F401


class MotionPrimitiveSearch(_Base):
    variant = 'single_lane'
