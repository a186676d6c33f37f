"""Vehicle geometry constants (reference: main/lib/car_dimensions.py:52-107). Two collision discs of radius
width/sqrt(2) centred on the longitudinal axis; anchor point = rear axle."""
from typing import Tuple

import numpy as np


class CarDimensions:
    wheelbase: float = 0.0
    box: Tuple[float, float] = (0.0, 0.0)      # (width, length)

    def __init__(self, skip_back_circle_collision_checking: bool = False):
        self.skip_back_circle_collision_checking = skip_back_circle_collision_checking

    @property
    def distance_back_to_front_wheel(self) -> float:
        return self.wheelbase

    @property
    def bounding_box_size(self) -> Tuple[float, float]:
        return self.box

    @property
    def center_point_offset(self) -> Tuple[float, float]:
        return self.distance_back_to_front_wheel / 2, 0.0

    @property
    def radius(self) -> float:
        return self.bounding_box_size[0] / (2 ** .5)

    @property
    def circle_centers(self) -> np.ndarray:
        width, length = self.bounding_box_size
        half_span = length / 2 - width / 2
        mid_x, mid_y = self.center_point_offset
        discs = [[mid_x + half_span, mid_y]]
        if not self.skip_back_circle_collision_checking:
            discs.append([mid_x - half_span, mid_y])
        return np.array(discs)


class BicycleModelDimensions(CarDimensions):
    wheelbase = 2.86

    @property
    def bounding_box_size(self):
        return 2.0, self.wheelbase + 0.64


class PriusDimensions(CarDimensions):
    def __init__(self, scaling_factor: float = 1., skip_back_circle_collision_checking=False):
        super().__init__(skip_back_circle_collision_checking)
        self._k = scaling_factor

    @property
    def distance_back_to_front_wheel(self):
        return 4 * self._k

    @property
    def bounding_box_size(self):
        return 2.04 * self._k, 4.84 * self._k
