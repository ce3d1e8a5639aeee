#!/usr/bin/env python3
"""Developer tool (GPU box): stage times of the batched pipeline for several (paths, samples) shapes with the
same number of sample points — shows how the velocity kernel's block shape / residency affects throughput."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vexautonomousplanner_amd.batch import BatchedTrajectoryGenerator
from vexautonomousplanner_amd.synth import make_waypoints, DEFAULT_CONSTRAINTS

shapes = [(16384, 2500), (8192, 5000), (4096, 10000), (2048, 20000)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for B, S in shapes:
    gen = BatchedTrajectoryGenerator(0, "f32")
    wp = torch.tensor(make_waypoints(B, 32, 3), device="cuda:0", dtype=torch.float32)
    out = None
    for _ in range(3):
        out = gen.profile(wp, DEFAULT_CONSTRAINTS, samples=S, out=out)
    gen.ctx.set_timing(True)
    acc = {}
    for _ in range(5):
        out = gen.profile(wp, DEFAULT_CONSTRAINTS, samples=S, out=out)
        for k, v in gen.timing().items():
            acc[k] = acc.get(k, 0.0) + v / 5
    gen.ctx.set_timing(False)
    pts = B * S
    print(f"{B:6d} x {S:6d}: " + " ".join(f"{k} {v:.3f}" for k, v in acc.items()) +
          f" | velocity {pts / acc['velocity'] / 1e6:.1f} Gpt/s-ms^-1".replace("Gpt/s-ms^-1", "pts/ns x1e-3"), flush=True)
