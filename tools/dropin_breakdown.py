#!/usr/bin/env python3
"""Developer tool (GPU box): where a config-1 call through the drop-in classes spends its wall time
(build_path; rebuild_tables; the device call of generate_motion_profile; the conversion to Python lists)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "dropin"))
import numpy as np

from motion_profiling_v2 import motion_profile_generator as mpg
from splines.spline_manager import QuinticHermiteSplineManager
from vexautonomousplanner_amd.nodes import Node
from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints

wp = make_waypoints(1, 8, 1)[0].astype(np.float64)
nodes = [Node() for _ in wp]
acc = {}
N = 30
for i in range(N + 3):
    t = [time.perf_counter()]
    sm = QuinticHermiteSplineManager()
    sm.build_path(wp, nodes, [])
    t.append(time.perf_counter())
    cons = mpg.Constraints(*DEFAULT_CONSTRAINTS)
    sm.rebuild_tables()
    t.append(time.perf_counter())
    rows, nodes_map, actions_map = sm._dev().motion_profile(cons, 0.01, 0.005)
    t.append(time.perf_counter())
    cols = [[float(v) for v in rows[:, k]] for k in range(6)]
    coords = [rows[k, 6:8].copy() for k in range(len(rows))]
    t.append(time.perf_counter())
    if i >= 3:
        for name, a, b in zip(("build_path", "rebuild_tables", "device motion_profile", "lists"), t[:-1], t[1:]):
            acc[name] = acc.get(name, 0.0) + (b - a) * 1e3 / N
print({k: round(v, 3) for k, v in acc.items()}, "rows", len(rows))
