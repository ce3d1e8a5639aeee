"""motion_profile_generator with the reference's call surface
(motion_profiling_v2/motion_profile_generator.py:14-646), the heavy parts on the device.

  Constraints                -> same dataclass, same helper methods                      MPG:14-67
  forward_backward_pass      -> K3+K4 sampling and K5 velocity pass (vap_sample, vap_velocity_pass)  MPG:70-316
  generate_motion_profile    -> rebuild_tables + forward_backward_pass on the device, then the
                                time-domain resample                MPG:389-628
Plain nodes only for now: per-node stop / velocity / acceleration limits and action points
(MPG:100-163) are SURVEY §8(f) rank 2.
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


def _plain_route(spline_manager) -> bool:
    for n in spline_manager.nodes:
        if n.stop or n.max_velocity > 0 or n.max_acceleration > 0:
            return False
    return len(spline_manager.action_points) == 0


def forward_backward_pass(spline_manager, constraints: Constraints, delta_dist: float,
                          start_vel: float = 0.01, end_vel: float = 0.01) -> List[float]:
    """MPG:70-316: distance-grid sampling + forward/backward acceleration-limited pass, on the GPU."""
    if not _plain_route(spline_manager):
        raise NotImplementedError("per-node / action-point limits (MPG:100-163) are SURVEY §8(f) rank 2: "
                                  "not on the device path yet")
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
    """MPG:389-628.  The distance-domain part runs on the device; the time-domain resample is the
    next row of the scope table."""
    raise NotImplementedError("time-domain resample (MPG:413-628): SURVEY §8(f) rank 1, next to be built")
