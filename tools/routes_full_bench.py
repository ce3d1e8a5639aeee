#!/usr/bin/env python3
"""Developer tool (GPU box): the whole batched route pipeline on config-3-shaped routes with reverse / turn nodes, node
limits, waits — profile_routes -> apply_node_limits -> time_profile -> insert_waits — wall time per stage (run it under
rocprofv3 --kernel-trace --stats for the kernels).   python tools/routes_full_bench.py [paths] [waypoints] [samples]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from vexautonomousplanner_amd.batch import BatchedTrajectoryGenerator
from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
W = int(sys.argv[2]) if len(sys.argv) > 2 else 32
S = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
rng = np.random.default_rng(1)
gen = BatchedTrajectoryGenerator(0, "f32")
wp = torch.tensor(make_waypoints(B, W, 3), dtype=gen.tdtype, device=gen.device)
rev = rng.random((B, W)) < 0.05
turn = np.where(rng.random((B, W)) < 0.05, 90.0, 0.0)
rev[:, -1] = False
turn[:, -1] = turn[:, 0] = 0.0
mv = np.where(rng.random((B, W)) < 0.3, rng.uniform(1.5, 3.5, (B, W)), 0.0)
stop = rng.random((B, W)) < 0.1
stop[:, 0] = stop[:, -1] = False
wait = np.where(rng.random((B, W)) < 0.1, 0.25, 0.0)


def stage(name, fn, n=5):
    out = fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(n):
        out = fn()
    ev[1].record()
    torch.cuda.synchronize()
    print(f"{name}: {ev[0].elapsed_time(ev[1]) / n:.3f} ms", flush=True)
    return out


res = stage("profile_routes", lambda: gen.profile_routes(wp, node_reverse=rev, node_turn=turn, constraints=DEFAULT_CONSTRAINTS, samples=S))
stage("apply_node_limits", lambda: gen.apply_node_limits(res, DEFAULT_CONSTRAINTS, node_max_velocity=mv, node_stop=stop))
tp = stage("time_profile", lambda: gen.time_profile(res, DEFAULT_CONSTRAINTS, capacity_rows=4096, node_reverse=rev))
out = stage("insert_waits", lambda: gen.insert_waits(res, tp, node_wait_time=wait, node_turn=turn, node_reverse=rev, constraints=DEFAULT_CONSTRAINTS))
print("rows", int(out["counts"][:, 0].sum().item()), "flags_or", int(res["flags"].max().item()))
