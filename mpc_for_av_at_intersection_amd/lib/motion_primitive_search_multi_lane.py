"""reference: main/lib/motion_primitive_search_multi_lane.py (weighted heuristic :155-181 and edge cost :226-237)."""
from .motion_primitive_search import MotionPrimitiveSearch as _Base, NodeType  # noqa: F401


class MotionPrimitiveSearch(_Base):
    variant = 'multi_lane'
