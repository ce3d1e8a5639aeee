#!/usr/bin/env python3
"""Developer tool (GPU box): randomized bit-identity sweep of the lane-per-path velocity kernel against the kernel it
must equal.

Random batch shapes (B, W, S on the fixed-S grid or the reference's dd grid), random robots, start / end velocities:
  * velocity rows of lanes16 / lanes32 / lanes64 == the sequential sweep's (seq_fast), fp32 and fp64 rows;
  * the same under random node limits (max_velocity, max_acceleration, stop: the VCAP / ACC instantiations).
tests/test_gpu_lanes.py holds the curated cases; this looks for rare ones (ragged rows, partial groups, rows with and
without interior tiles).
  python tools/fuzz_lanes.py [seconds]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from vexautonomousplanner_amd.batch import BatchedTrajectoryGenerator
from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(20261004)
ROWS = ("x", "y", "heading", "curvature", "velocity")
gens = {}


def gen(dt, kernel):
    key = (dt, kernel)
    if key not in gens:
        gens[key] = BatchedTrajectoryGenerator(0, dt, velocity_kernel=kernel)
    return gens[key]


fails = n_cases = 0
t0 = time.time()
while time.time() - t0 < budget:
    W = int(rng.choice([2, 3, 4, 5, 8, 13, 32, 57, 113, 300]))
    S = int(rng.choice([2, 3, 7, 63, 64, 65, 127, 128, 129, 255, 1000, 1024, 1025, 4096, 4097, 10000, 20481]))
    B = int(rng.choice([1, 2, 3, 15, 16, 17, 31, 33, 63, 65, 130])) if S <= 1100 else int(rng.integers(1, 20))
    seed = int(rng.integers(0, 1 << 30))
    cons = list(DEFAULT_CONSTRAINTS)
    if rng.random() < 0.5:
        cons[0] = float(rng.uniform(1.0, 8.0))
        cons[1] = float(rng.uniform(2.0, 16.0))
        cons[2] = float(rng.uniform(2.0, 16.0))
        cons[5] = float(rng.uniform(0.5, 2.0))
    sv, ev = (0.01, 0.01) if rng.random() < 0.7 else (float(rng.uniform(0.0, 2.0)), float(rng.uniform(0.0, 2.0)))
    kw = dict(samples=S)
    if S <= 4097 and rng.random() < 0.3:
        kw = dict(dd=float(rng.choice([0.0005, 0.002, 0.005, 0.02])), capacity=int(rng.choice([512, 4096, 16384])))
    wp64 = make_waypoints(B, W, seed).astype(np.float64)
    what = f"B={B} W={W} {kw} sv={sv:.3f} ev={ev:.3f} seed={seed} cons={[round(c, 3) for c in cons]}"
    for dt in ("f32", "f64"):
        wp = torch.tensor(wp64, device="cuda:0", dtype=torch.float32 if dt == "f32" else torch.float64)
        ref = {k: v.clone() for k, v in gen(dt, "seq_fast").profile(wp, cons, start_vel=sv, end_vel=ev, **kw).items()}
        for kern in ("lanes16", "lanes32", "lanes64"):
            got = gen(dt, kern).profile(wp, cons, start_vel=sv, end_vel=ev, **kw)
            # NaN rows (flagged degenerate paths) compare as bit patterns
            if not torch.equal(got["velocity"].view(torch.int32 if dt == "f32" else torch.int64),
                               ref["velocity"].view(torch.int32 if dt == "f32" else torch.int64)):
                fails += 1
                print(f"FAIL {dt} {kern} velocity differs from seq_fast: {what}", flush=True)
        if W >= 3 and n_cases % 2 == 0:
            # node limits: the VCAP (max_velocity / stop) and ACC (max_acceleration) instantiations
            lim = np.random.default_rng(seed)
            mv = np.where(lim.random((B, W)) < 0.3, lim.uniform(0.5, cons[0], (B, W)), 0.0)
            stop = lim.random((B, W)) < 0.15
            stop[:, 0] = stop[:, -1] = False
            ma = np.where(lim.random((B, W)) < 0.4, lim.uniform(1.0, 20.0, (B, W)), 0.0) if n_cases % 4 == 0 else None
            rows = {}
            for kern in ("seq_fast", "lanes16", "lanes32", "lanes64"):
                g = gen(dt, kern)
                r = g.profile(wp, cons, start_vel=sv, end_vel=ev, **kw)
                g.apply_node_limits(r, cons, node_max_velocity=mv, node_stop=stop, node_max_acceleration=ma, start_vel=sv, end_vel=ev)
                rows[kern] = r["velocity"].clone().view(torch.int32 if dt == "f32" else torch.int64)
            for kern in ("lanes16", "lanes32", "lanes64"):
                if not torch.equal(rows[kern], rows["seq_fast"]):
                    fails += 1
                    print(f"FAIL {dt} {kern} velocity under node limits ({'acc' if ma is not None else 'vcap'}) differs from seq_fast: {what}", flush=True)
    torch.cuda.synchronize()
    n_cases += 1
print(f"{n_cases} cases in {time.time() - t0:.0f} s, {fails} failures")
sys.exit(1 if fails else 0)
