"""Plant / world model used by the scenario loop (reference: main/lib/simulation.py:11-88, main/bicycle/main.py:28-41).
Scalar host code: one Euler step of the kinematic bicycle per control period. The batched device version of the
same step is mpcx_plant_step_batch."""
import math
from dataclasses import dataclass
from typing import List, Optional

import numpy as np


@dataclass
class State:
    x: float = 0.0
    y: float = 0.0
    yaw: float = 0.0
    v: float = 0.0


class Simulation:
    MAX_STEER = np.deg2rad(45.0)
    MAX_SPEED = 30.0 / 3.6
    MIN_SPEED = -5.

    def __init__(self, car_dimensions, sample_time: float, initial_state: State):
        self._L = car_dimensions.distance_back_to_front_wheel
        self._dt = sample_time
        self._x, self._y, self._th, self._v = initial_state.x, initial_state.y, initial_state.yaw, initial_state.v

    def step(self, a: float, delta: float) -> State:
        delta = max(min(delta, Simulation.MAX_STEER), -Simulation.MAX_STEER)
        v = self._v
        self._x += (v * np.cos(self._th)) * self._dt
        self._y += (v * np.sin(self._th)) * self._dt
        self._th += ((v / self._L) * np.tan(delta)) * self._dt
        self._v = max(min(v + a * self._dt, Simulation.MAX_SPEED), Simulation.MIN_SPEED)
        return State(x=self._x, y=self._y, yaw=self._th, v=self._v)


class History:
    def __init__(self, sample_time: float):
        self.x: List[float] = []; self.y: List[float] = []; self.yaw: List[float] = []; self.v: List[float] = []
        self.t: List[float] = []; self.delta: List[float] = []; self.a: List[float] = []
        self.xref_deviation: List[float] = []
        self._sample_time = sample_time

    def store(self, state: State, a: float, delta: float, xref_deviation: Optional[float] = None):
        self.x.append(state.x); self.y.append(state.y); self.yaw.append(state.yaw); self.v.append(state.v)
        self.t.append(self.get_current_time() + self._sample_time)
        self.delta.append(delta); self.a.append(a)
        self.xref_deviation.append(xref_deviation if xref_deviation is not None else np.nan)

    def get_current_time(self) -> float:
        return self.t[-1] if self.t else 0.


class HistorySimulation(Simulation):
    def __init__(self, car_dimensions, sample_time: float, initial_state: State):
        super().__init__(car_dimensions, sample_time, initial_state)
        self.history = History(sample_time=sample_time)
        self.history.store(initial_state, a=0., delta=0., xref_deviation=0.)

    def step(self, a: float, delta: float, xref_deviation: Optional[float] = None) -> State:
        st = super().step(a, delta)
        self.history.store(st, a=a, delta=delta, xref_deviation=xref_deviation)
        return st
