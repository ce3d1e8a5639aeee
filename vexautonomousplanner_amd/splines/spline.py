"""Spline base class with the reference's interface (splines/spline.py:7-105).

Only the contract lives here.  The generic helpers of the reference base class (heading, curvature,
polyline arc length) are expressed through the abstract evaluators, which the concrete class runs
on the device."""
from abc import ABC, abstractmethod

import numpy as np


class Spline(ABC):
    def __init__(self):
        self.x_points = None
        self.y_points = None
        self._length_cache = None

    @abstractmethod
    def fit(self, x, y) -> bool:
        ...

    @abstractmethod
    def get_point(self, t: float) -> np.ndarray:
        ...

    @abstractmethod
    def get_derivative(self, t: float) -> np.ndarray:
        ...

    @abstractmethod
    def get_second_derivative(self, t: float) -> np.ndarray:
        ...

    def get_heading(self, t: float) -> float:
        """Tangent angle at t (splines/spline.py:48-59)."""
        d = self.get_derivative(t)
        return float(np.arctan2(d[1], d[0]))

    def get_curvature(self, t: float) -> float:
        """Signed curvature at t, 0 where the speed vanishes (splines/spline.py:61-80)."""
        d1 = self.get_derivative(t)
        d2 = self.get_second_derivative(t)
        speed2 = d1[0] * d1[0] + d1[1] * d1[1]
        if speed2 < 1e-10:
            return 0.0
        return float((d1[0] * d2[1] - d1[1] * d2[0]) / speed2 ** 1.5)

    def get_arc_length(self, t_start: float, t_end: float, num_points: int = 100) -> float:
        """Polyline length between two parameters (splines/spline.py:82-105)."""
        ts = np.linspace(t_start, t_end, num_points)
        pts = np.array([self.get_point(t) for t in ts])
        return float(np.sum(np.linalg.norm(np.diff(pts, axis=0), axis=1)))
