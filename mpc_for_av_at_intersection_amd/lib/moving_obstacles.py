"""Scripted traffic of the reference's scenarios (reference: main/lib/moving_obstacles.py:16-231, plant
main/bicycle/main.py:14-41): vehicles that follow a hard-wired steering rule at constant speed after a start delay.

Each class keeps the reference's constructor, `.step()`, `.get()` -> (x, y, v, yaw, a, steer), the `steering_angle`
/ `forward_velocity` properties and the `model.xc / yc / theta` attributes the scenario scripts read.  `get()` is the
6-tuple the conflict search consumes (mpcx_interaction_batch's obs6 rows); `tape(n)` returns n consecutive
get()/step() pairs as an (n, 6) array ready to be uploaded as a device-side traffic table.  Pure host scalar code --
one multiply-add per vehicle and step -- with numpy's libm so the poses match the reference bit for bit.
"""
from typing import Tuple

import numpy as np

from .car_dimensions import BicycleModelDimensions, CarDimensions  # noqa: F401


def calculate_steering_angle_for_radius(radius, L=2.86) -> float:
    """constant steering angle whose circle has the given radius (moving_obstacles.py:16-25)"""
    return np.arctan((1 / radius) * L)


class Bicycle:
    """velocity-driven kinematic bicycle (bicycle/main.py:14-41): pose integrates (v cos, v sin, v/L tan(delta)) * dt"""

    def __init__(self, car_dimensions, sample_time: float = 0.2):
        self.xc = 0
        self.yc = 0
        self.theta = 0
        self.sample_time = sample_time
        self.L = car_dimensions.distance_back_to_front_wheel

    def reset(self):
        self.xc = self.yc = self.theta = 0

    def step(self, v, delta):
        dx = v * np.cos(self.theta)
        dy = v * np.sin(self.theta)
        dth = (v / self.L) * np.tan(delta)
        self.xc += dx * self.sample_time
        self.yc += dy * self.sample_time
        self.theta += dth * self.sample_time


class _ScriptedVehicle:
    """shared mechanics: start delay, constant speed, get()/step(); subclasses provide `steering_angle`"""

    def _init_common(self, car_dimensions, speed, offset, model_dt, counter_dt):
        self.speed = speed
        self.model = Bicycle(car_dimensions=car_dimensions, sample_time=model_dt)
        self.offset = offset if (offset is not None and offset > 0) else None
        self.dt = counter_dt
        self.counter = 0

    @property
    def forward_velocity(self):
        waiting = self.offset is not None and not (self.counter > (self.offset / self.dt))
        return 0 if waiting else self.speed

    def step(self):
        delta = self.steering_angle
        self.model.step(self.forward_velocity, delta)
        self.counter += 1

    def get(self) -> Tuple[float, float, float, float, float, float]:
        return self.model.xc, self.model.yc, self.forward_velocity, self.model.theta, 0.0, self.steering_angle

    def tape(self, n_steps: int) -> np.ndarray:
        out = np.empty((n_steps, 6))
        for k in range(n_steps):
            out[k] = self.get()
            self.step()
        return out

    def _place_on_main_road(self, direction):
        self.direction = 1 if direction >= 0 else -1
        if self.direction == 1:
            self.model.xc, self.model.yc, self.model.theta, self.x_turn = -30, -3, 0, -10
        else:
            self.model.xc, self.model.yc, self.model.theta, self.x_turn = 30, 3, np.pi, 12


class MovingObstacleTIntersection(_ScriptedVehicle):
    """cross traffic of the (T-)intersection scenarios (moving_obstacles.py:165-231): enters on the main road from the
    left (direction >= 0, lane y = -3) or right (lane y = +3); a turning vehicle steers -0.38 rad (short right turn)
    or +0.19 rad (long left turn) from x_turn on until its heading has swept a quarter turn"""

    def __init__(self, car_dimensions, direction: int, turning: bool, speed: float, offset=None, dt=10e-3):
        self.turning = turning
        self._init_common(car_dimensions, speed, offset, dt, dt)
        self._place_on_main_road(direction)

    @property
    def steering_angle(self) -> float:
        if self.turning is not True:
            return 0.
        m = self.model
        if self.direction == 1:
            return -0.38 if (m.xc >= self.x_turn and m.theta > (-np.pi / 2)) else 0.
        return 0.19 if (m.xc <= self.x_turn and m.theta < (3 * np.pi / 2)) else 0.


class MovingObstacleArterial(_ScriptedVehicle):
    """vehicle driving straight up (+y) from (x_init, y_init) (moving_obstacles.py:128-163)"""

    def __init__(self, car_dimensions, x_init: float, y_init: float, speed: float, offset=None, dt=10e-3):
        self._init_common(car_dimensions, speed, offset, dt, dt)
        self.model.xc, self.model.yc, self.model.theta = x_init, y_init, np.pi / 2

    @property
    def steering_angle(self) -> float:
        return 0.0


class MovingObstacleRoundabout(_ScriptedVehicle):
    """roundabout traffic (moving_obstacles.py:28-126): position-triggered arcs of radius 5 around the island; the
    start-delay counter always uses 0.2 s whatever `dt` drives the plant (:45), and reading `steering_angle` may SNAP the
    heading to -pi / 0 once the vehicle has come around (:81-83, :103-105) -- get() and step() both read it, as in the
    reference"""

    def __init__(self, car_dimensions, direction: int, turning: bool, speed: float, offset=None, dt=10e-3, start_pos=2, end_pos=4):
        self.turning = turning
        self._init_common(car_dimensions, speed, offset, dt, 0.2)
        self.start_pos, self.end_pos = start_pos, end_pos
        self._place_on_main_road(direction)

    @property
    def steering_angle(self) -> float:
        if self.turning is not True:
            return 0.0
        m = self.model
        arc = calculate_steering_angle_for_radius(5)
        delta = 0.0
        if self.direction == 1:
            if -7 <= m.xc <= -4 and m.yc < 0:
                delta = -arc
                print(m.xc, m.yc)
            if -3 < m.xc:
                delta = arc
            if m.yc > 0 and -5 <= m.xc <= -3:
                delta = -arc
            if m.xc <= -3 and m.yc > 0:
                m.theta = -np.pi
                delta = 0
        else:
            if 4 <= m.xc <= 7 and m.yc > 0:
                delta = -arc
                print(m.xc, m.yc)
            if m.xc < 3:
                delta = arc
            if m.yc < 0 and 3 <= m.xc <= 5:
                delta = -arc
            if 3 <= m.xc and m.yc < 0:
                m.theta = 0
                delta = 0
        return delta
