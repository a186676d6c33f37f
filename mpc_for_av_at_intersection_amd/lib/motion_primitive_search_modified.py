"""reference: main/lib/motion_primitive_search_modified.py (goal-POINT heuristic, :80-89) -- what every stock MPC
scenario imports."""
from .motion_primitive_search import MotionPrimitiveSearch as _Base, NodeType  # noqa: F401


class MotionPrimitiveSearch(_Base):
    variant = 'modified'
