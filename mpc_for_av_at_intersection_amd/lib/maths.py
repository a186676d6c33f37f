"""Angle wrap to [-pi, pi) (reference: main/lib/maths.py:4-10). Scalar host helper; the device twin is
`normalize_angle` in csrc/mpcx_expand.hip."""
import math


def normalize_angle(theta: float) -> float:
    theta = theta % math.tau
    return theta - math.tau if theta >= math.pi else theta
