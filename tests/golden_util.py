"""Helpers shared by the parity tests: loading tests/golden/*.npz (vectors produced by the real
reference, see oracle/gen_golden.py) and turning them into oracle / product inputs."""
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def names(prefix=None):
    out = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "*.npz")))
    if prefix is not None:
        pre = (prefix,) if isinstance(prefix, str) else tuple(prefix)
        out = [n for n in out if n.startswith(pre)]
    return out


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def node_dict(g):
    return dict(is_reverse=g["node_is_reverse_node"], turn=g["node_turn"], stop=g["node_stop"],
                wait_time=g["node_wait_time"], max_velocity=g["node_max_velocity"],
                max_acceleration=g["node_max_acceleration"], tangent=g["node_tangent"],
                magnitudes=np.nan_to_num(g["node_magnitudes"]))


def action_dict(g):
    if "ap_t" not in g.files:
        return None
    return dict(t=g["ap_t"], stop=g["ap_stop"], wait_time=g["ap_wait_time"],
                max_velocity=g["ap_max_velocity"], max_acceleration=g["ap_max_acceleration"])


def ref_segments(g):
    ns = int(g["n_splines"])
    seg = np.concatenate([g[f"spline{i}_segments"] for i in range(ns)])
    sl = np.concatenate([g[f"spline{i}_segment_lengths"] for i in range(ns)])
    pl = np.array([float(g[f"spline{i}_param_last"]) for i in range(ns)])
    return seg, sl, pl
