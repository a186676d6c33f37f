"""MI355X-native (gfx950) batched implementation of the per-timestep hot path of
SaeedRahmani/MPC_for_AV_at_Intersection: the bicycle-model LTV-MPC step and the motion-primitive A* expansion.

Layers
  csrc/ + libmpcx.so   hand-written HIP kernels behind the C ABI of include/mpcx.h
  runtime.py           device runtime (torch tensors as buffers, ctypes calls)
  lib/                 the reference's Python call surface (MPC, MotionPrimitiveSearch, ...) on top of it
"""
from ._lib import LIB_PATH  # noqa: F401

__all__ = ['LIB_PATH']
