"""1-D trapezoid / triangle velocity profile for in-place turns
(motion_profiling_v2/one_dim_mp_generator.py:4-69).  Host code: SURVEY §8(f) rank 3 — a few hundred
samples per turn, reached only for nodes with turn != 0."""
import numpy as np


def generate_trapezoidal_profile(max_velocity, max_acceleration, total_distance, time_step=0.01):
    t_acc = max_velocity / max_acceleration
    d_acc = 0.5 * max_acceleration * t_acc ** 2
    if 2 * d_acc > total_distance:  # never reaches max_velocity: triangle
        t_acc = np.sqrt(total_distance / max_acceleration)
        max_velocity = max_acceleration * t_acc
        total_time = 2 * t_acc
    else:
        total_time = 2 * t_acc + (total_distance - 2 * d_acc) / max_velocity
    times = np.arange(0, total_time + time_step, time_step)
    velocity = np.zeros_like(times)
    for i, t in enumerate(times):
        if t <= t_acc:
            velocity[i] = max_acceleration * t
        elif t <= total_time - t_acc:
            velocity[i] = max_velocity
        else:
            velocity[i] = max_velocity - max_acceleration * (t - (total_time - t_acc))
    return velocity
