#!/usr/bin/env python3
"""Developer tool (GPU box): the velocity kernel's group size (paths per workgroup) on the bench's workloads —
    python tools/lanes_group_bench.py c4 c5        (stage times per group size, default mode)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from vexautonomousplanner_amd.batch import BatchedTrajectoryGenerator
from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints

for name in sys.argv[1:] or ["c4", "c5"]:
    wl = bench.WORKLOADS[name]
    wp = torch.tensor(make_waypoints(wl["paths"], wl["W"], wl["seed"]), device="cuda:0")
    for kern in ("auto", "lanes16", "lanes32", "lanes64"):
        gen = BatchedTrajectoryGenerator(0, "f32", velocity_kernel=kern, time_domain_residual=False)
        out = None
        for _ in range(5):
            out = gen.profile(wp, DEFAULT_CONSTRAINTS, samples=wl["S"], out=out)
        torch.cuda.synchronize()
        gen.ctx.set_timing(True)
        acc = {}
        for _ in range(10):
            out = gen.profile(wp, DEFAULT_CONSTRAINTS, samples=wl["S"], out=out)
            for k, v in gen.timing().items():
                acc[k] = acc.get(k, 0.0) + v / 10
        print(f"{name} {kern:8s} velocity {acc['velocity']:.4f} ms  total {acc['total']:.4f} ms", flush=True)
        del gen, out
