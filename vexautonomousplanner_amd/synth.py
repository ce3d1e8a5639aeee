"""Synthetic waypoint batches for parity tests and bench.py (SURVEY.md §8(d) "Synthetic inputs").

The reference has no generator of its own (it takes clicks on a field image, gui/path.py:356-390);
this one produces smooth random-walk paths with no coincident waypoints, in feet, and default robot
constraints from utilities/config_manager.py:27-47 / gui/gui_manager.py:92 / gui/path.py:314-321.
"""
import numpy as np

# max_vel, max_acc, max_dec, friction_coef, max_jerk, track_width  (ft, s)
DEFAULT_CONSTRAINTS = (4.0, 8.0, 8.0, 0.8, 16.0, 12.5 / 12.0)
START_VEL = 0.01  # motion_profile_generator.py:74
END_VEL = 0.01  # motion_profile_generator.py:75
DEFAULT_DD = 0.005  # motion_profile_generator.py:390
DEFAULT_DT = 0.01  # motion_profile_generator.py:390


def make_waypoints(batch: int, n_waypoints: int, seed: int, dtype=np.float32) -> np.ndarray:
    """(batch, n_waypoints, 2) waypoints in feet.

    Start at (-5,-5) ft; step length ~U(0.3,1.0) ft; heading random walk with
    d_psi ~ N(0, 0.6 rad).  Generated in fp64 and rounded once to ``dtype`` so that an fp32 run and
    the fp64 reference see bit-identical waypoint values.
    """
    rng = np.random.default_rng(seed)
    psi0 = rng.uniform(0.0, 2.0 * np.pi, size=(batch, 1))
    dpsi = rng.normal(0.0, 0.6, size=(batch, n_waypoints - 1))
    step = rng.uniform(0.3, 1.0, size=(batch, n_waypoints - 1))
    psi = psi0 + np.cumsum(dpsi, axis=1)
    pts = np.empty((batch, n_waypoints, 2), dtype=np.float64)
    pts[:, 0, :] = -5.0
    pts[:, 1:, 0] = -5.0 + np.cumsum(step * np.cos(psi), axis=1)
    pts[:, 1:, 1] = -5.0 + np.cumsum(step * np.sin(psi), axis=1)
    return np.ascontiguousarray(pts.astype(dtype))


def make_waypoints_block(paths_per_block: int, n_waypoints: int, seed: int, block: int, dtype=np.float32) -> np.ndarray:
    """Block `block` of the seeded GLOBAL batch a multi-GPU run works on: the global batch is the concatenation of
    blocks of ``paths_per_block`` paths, block 0 being ``make_waypoints(paths_per_block, W, seed)`` (so a 1-GPU run is
    the batch it always was) and block r > 0 drawn from the stream ``(seed, r)``.  Rank r of a weak-scaling run
    generates block r alone — nobody builds the other ranks' paths."""
    if block == 0:
        return make_waypoints(paths_per_block, n_waypoints, seed, dtype)
    return make_waypoints(paths_per_block, n_waypoints, [int(seed), int(block)], dtype)
