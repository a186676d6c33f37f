"""Process-wide device context shared by the drop-in classes (created on first use; raises without a GPU)."""
from ..runtime import Context

_ctx = None


def context() -> Context:
    global _ctx
    if _ctx is None or not getattr(_ctx, '_ctx', None):      # none yet, or its owner closed it: the drop-in classes get a fresh one
        _ctx = Context(0)
    return _ctx


def set_context(ctx: Context):
    global _ctx
    _ctx = ctx
