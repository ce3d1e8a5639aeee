#!/usr/bin/env python3
"""Developer tool (GPU box): cost of the batched time-domain resample (vap_time_profile) after a config-3 batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vexautonomousplanner_amd.batch import BatchedTrajectoryGenerator
from vexautonomousplanner_amd.synth import make_waypoints, DEFAULT_CONSTRAINTS

B, W, S = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (4096, 32, 10000)))
gen = BatchedTrajectoryGenerator(0, "f32")
wp = torch.tensor(make_waypoints(B, W, 3), device="cuda:0", dtype=torch.float32)
res = gen.profile(wp, DEFAULT_CONSTRAINTS, samples=S)
tp = gen.time_profile(res, DEFAULT_CONSTRAINTS, capacity_rows=2048)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(5):
    tp = gen.time_profile(res, DEFAULT_CONSTRAINTS, capacity_rows=2048, out=tp)
ev[1].record()
torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / 5
rows = int(tp["counts"][:, 0].sum().item())
print(f"{B} x {W} x {S}: time-domain resample {ms:.3f} ms, {rows} rows ({rows / B:.0f} per path), "
      f"{rows / ms / 1e3:.1f} M rows/s, flags_or {int(res['flags'].max().item())}")
