"""Motion primitives (reference: main/lib/motion_primitive.py:9-45). The reference ships 9 pickles per vehicle
model; here the same arrays live in data/motion_primitives.npz (plain numpy, no pickle)."""
import os
from dataclasses import dataclass, field
from typing import Dict

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'data', 'motion_primitives.npz')


@dataclass
class MotionPrimitive:
    name: str
    forward_speed: float
    steering_angle: float
    n_seconds: float
    total_length: float = 0.
    points: np.ndarray = field(default_factory=lambda: np.array([]))


def load_motion_primitives(version="prius") -> Dict[str, MotionPrimitive]:
    if version not in ("prius", "bicycle_model"):
        raise Exception("Motion primitives version not recognized!")
    z = np.load(_DATA)
    names = sorted({k.split('/')[1] for k in z.files if k.startswith(version + '/')})
    if not names:
        raise Exception("No motion primitives found.")
    out = {}
    for n in names:
        fs, sa, ns, tl = z['%s/%s/meta' % (version, n)]
        out[n] = MotionPrimitive(name=n, forward_speed=float(fs), steering_angle=float(sa), n_seconds=float(ns),
                                 total_length=float(tl), points=z['%s/%s/points' % (version, n)].copy())
    return out
