#!/usr/bin/env python3
"""Developer tool (GPU box): stage times of the batched route path (profile_routes: reverse / turn splits) next to the
plain path on the same waypoints.   python tools/routes_bench.py [paths] [waypoints] [samples] [split probability]"""
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
psplit = float(sys.argv[4]) if len(sys.argv) > 4 else 0.1
rng = np.random.default_rng(1)
gen = BatchedTrajectoryGenerator(0, "f32")
wp = torch.tensor(make_waypoints(B, W, 3), dtype=gen.tdtype, device=gen.device)
rev = rng.random((B, W)) < psplit / 2
turn = np.where(rng.random((B, W)) < psplit / 2, 90.0, 0.0)
rev[:, -1] = False
turn[:, -1] = turn[:, 0] = 0.0


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    gen.ctx.set_timing(True)
    acc = {}
    for _ in range(n):
        fn()
        for k, v in gen.timing().items():
            acc[k] = acc.get(k, 0.0) + v / n
    gen.ctx.set_timing(False)
    return {k: round(v, 4) for k, v in acc.items()}


out = {}
print("plain ", timed(lambda: out.update(p=gen.profile(wp, DEFAULT_CONSTRAINTS, samples=S))))
print("routes", timed(lambda: out.update(r=gen.profile_routes(wp, node_reverse=rev, node_turn=turn, constraints=DEFAULT_CONSTRAINTS, samples=S))),
      "splines per route: mean %.2f max %d" % (float(out["r"]["spline_counts"].float().mean()), int(out["r"]["spline_counts"].max())))
none = np.zeros((B, W), dtype=bool)
print("routes without splits", timed(lambda: out.update(r0=gen.profile_routes(wp, node_reverse=none, node_turn=np.zeros((B, W)), constraints=DEFAULT_CONSTRAINTS, samples=S))))
same = all(torch.equal(out["p"][k], out["r0"][k]) for k in ("x", "y", "heading", "curvature", "velocity"))
print("routes without splits == plain path, bit for bit:", same)
