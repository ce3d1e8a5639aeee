"""motion_profile_generator with the reference's call surface
(motion_profiling_v2/motion_profile_generator.py:14-646), the heavy parts on the device.

  Constraints                -> same dataclass, same helper methods                      MPG:14-67
  forward_backward_pass      -> vap_route_forward_backward: sampling, node / action-point limits,
                                boundary_map, forward and backward sweeps                 MPG:70-316
  generate_motion_profile    -> vap_route_motion_profile: the above plus the time-domain resample
                                with turn / wait insertion                                MPG:389-628
"""
import math
from dataclasses import dataclass
from typing import List, Tuple

from . import one_dim_mp_generator


@dataclass
class Constraints:
    max_vel: float
    max_acc: float
    max_dec: float
    friction_coef: float
    max_jerk: float
    track_width: float

    def max_speed_at_curvature(self, curvature: float) -> float:
        if abs(curvature) < 1e-6:
            return self.max_vel
        w = 2 * self.max_vel / self.track_width
        return min((w * self.max_vel) / (abs(curvature) * self.max_vel + w), self.max_vel)

    def set_max_vel(self, max_vel):
        self.max_vel = max_vel

    def set_max_acc(self, max_acc):
        self.max_acc = max_acc

    def limit_velocity_by_ang_accel(self, dkappads: float, max_angular_accel: float) -> float:
        if abs(dkappads) < 1e-9:
            return self.max_vel
        v2 = max_angular_accel / abs(dkappads)
        if v2 < 0:
            return 0.0
        return min(math.sqrt(v2), self.max_vel)

    def max_accels_at_turn(self, angular_accel: float):
        left = self.max_acc + angular_accel * self.track_width / 2
        right = self.max_acc - angular_accel * self.track_width / 2
        return left if abs(left) < abs(right) else right

    def get_wheel_speeds(self, linear_vel: float, angular_vel: float) -> Tuple[float, float]:
        half = angular_vel * self.track_width / 2
        return linear_vel - half, linear_vel + half


def forward_backward_pass(spline_manager, constraints: Constraints, delta_dist: float,
                          start_vel: float = 0.01, end_vel: float = 0.01) -> List[float]:
    """MPG:70-316: distance-grid sampling + forward/backward acceleration-limited pass, on the GPU."""
    if spline_manager.lookup_table is None:
        spline_manager.build_lookup_table()
    out, _ = spline_manager._dev().forward_backward(constraints, delta_dist, start_vel, end_vel)
    return [float(v) for v in out["velocity"]]


def get_wheel_trajectory(linear_vels: List[float], angular_vels: List[float],
                         track_width: float) -> Tuple[List[float], List[float]]:
    """MPG:631-646."""
    left = [v - w * track_width / 2 for v, w in zip(linear_vels, angular_vels)]
    right = [v + w * track_width / 2 for v, w in zip(linear_vels, angular_vels)]
    return left, right


def motion_profile_angle(angle, constraints: Constraints, dt: float = 0.01):
    """MPG:319-346: heading / angular-velocity samples of an in-place turn."""
    arc = abs(angle) * constraints.track_width / 2
    vels = one_dim_mp_generator.generate_trapezoidal_profile(constraints.max_vel, constraints.max_acc, arc, dt)
    headings, travelled = [], 0
    sign = -1 if angle > 0 else 1
    for v in vels:
        headings.append(travelled / (constraints.track_width / 2) * sign)
        travelled += v * dt
    ang = [0] + [(headings[i] - headings[i - 1]) / dt for i in range(1, len(headings))]
    return headings, ang


def generate_motion_profile(spline_manager, constraints: Constraints, dt: float = 0.01, dd: float = 0.005):
    """MPG:389-628 on the device.  Returns the reference's 9-tuple (times, positions, linear_vels,
    accelerations, headings, angular_vels, nodes_map, actions_map, coords) as Python lists."""
    spline_manager.rebuild_tables()
    rows, nodes_map, actions_map = spline_manager._dev().motion_profile(constraints, dt, dd)
    cols = [[float(v) for v in rows[:, i]] for i in range(6)]
    coords = [rows[i, 6:8].copy() for i in range(len(rows))]
    return (cols[0], cols[1], cols[2], cols[3], cols[4], cols[5], [int(v) for v in nodes_map],
            [int(v) for v in actions_map], coords)
