#!/usr/bin/env python3
"""Developer tool (GPU box): wall time of config 1 (one 8-waypoint path) through the batched route calls with B = 1
(profile_routes -> apply_node_limits -> time_profile -> insert_waits), step by step, as DeviceRoute.motion_profile
issues them; velocity kernel auto vs seq_literal; row capacity small vs the drop-in's bound."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from vexautonomousplanner_amd.batch import BatchedTrajectoryGenerator
from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints

wp_np = make_waypoints(1, 8, 1).astype(np.float64)
W = 8
z = np.zeros((1, W))
zb = np.zeros((1, W), dtype=bool)
for kern, cap in (("auto", 1024), ("auto", 14096), ("seq_literal", 14096), ("seq_fast", 14096)):
    gen = BatchedTrajectoryGenerator(0, "f64", velocity_kernel=kern)
    acc = {}
    N = 30
    for i in range(N + 3):
        t = [time.perf_counter()]
        wp = torch.tensor(wp_np, dtype=torch.float64, device="cuda:0")
        res = gen.profile_routes(wp, node_reverse=zb, node_turn=z, constraints=DEFAULT_CONSTRAINTS, dd=0.005, capacity=1300,
                                 want=("curvature", "velocity"))
        t.append(time.perf_counter())
        tp = gen.time_profile(res, DEFAULT_CONSTRAINTS, dt=0.01, capacity_rows=cap, node_reverse=zb)
        t.append(time.perf_counter())
        out = gen.insert_waits(res, tp, node_wait_time=z, dt=0.01, node_turn=z, node_reverse=zb, constraints=DEFAULT_CONSTRAINTS)
        t.append(time.perf_counter())
        head = torch.cat([out["counts"][0], res["flags"][:1]]).cpu().numpy()
        t.append(time.perf_counter())
        rows = out["rows"][0, :int(head[0])].cpu().numpy()
        nm = out["nodes_map"][0, :int(head[1])].cpu().numpy()
        am = out["actions_map"][0, :int(head[2])].cpu().numpy()
        t.append(time.perf_counter())
        if i >= 3:
            for name, a, b in zip(("profile_routes", "time_profile", "insert_waits", "head (sync)", "copy out"), t[:-1], t[1:]):
                acc[name] = acc.get(name, 0.0) + (b - a) * 1e3 / N
    print(kern, cap, {k: round(v, 3) for k, v in acc.items()}, "total", round(sum(acc.values()), 3), "rows", rows.shape, flush=True)
