"""Constant-acceleration / constant-steer rollout of another vehicle (reference:
main/lib/moving_obstacles_prediction.py:10-47), GPU-backed through mpcx_predict_obstacles_batch."""
import numpy as np

from ._session import context


class MovingObstaclesPrediction:
    def __init__(self, x, y, v, yaw, a, steering_angle, sample_time: float, car_dimensions):
        self.x, self.y, self.v, self.yaw, self.a, self.steering_angle = x, y, v, yaw, a, steering_angle
        self.L = car_dimensions.distance_back_to_front_wheel
        self.sample_time = sample_time

    def state_prediction(self, time_horizon):
        t = np.arange(0, time_horizon, self.sample_time)
        n = len(t)
        ctx = context()
        six = ctx.f64([[self.x, self.y, self.v, self.yaw, self.a, self.steering_angle]])
        out = ctx.predict_obstacles(six, n, self.sample_time, self.L).cpu().numpy()[0]
        if n:                                   # the reference's object ends up at the last predicted state
            self.x, self.y, self.yaw = (float(v) for v in out[-1])
            for _ in range(n):
                self.v += self.a * self.sample_time
        return out[:, 0].copy(), out[:, 1].copy(), out[:, 2].copy(), np.arange(n) * self.sample_time
