"""main/lib/motion_primitive_search_roundabout.py of the reference: same class, `roundabout` cost / heuristic terms."""
from .motion_primitive_search import MotionPrimitiveSearch as _Base, NodeType  # noqa: F401


class MotionPrimitiveSearch(_Base):
    variant = 'roundabout'
