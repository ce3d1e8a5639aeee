#!/usr/bin/env python3
"""Developer tool (GPU box): randomised check of the batched time-domain resample (vap_time_profile, fp64)
against the oracle's time loop: random plain paths, robots (max_dec != max_acc), time and distance steps.
  python tools/fuzz_time_profile.py [seconds] [seed]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from oracle import oracle
from vexautonomousplanner_amd.batch import BatchedTrajectoryGenerator
from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
gen = BatchedTrajectoryGenerator(0, "f64")
n = fails = 0
worst = 0.0
t0 = time.time()
while time.time() - t0 < budget:
    W = int(rng.integers(2, 10))
    B = int(rng.integers(1, 7))
    wp = make_waypoints(B, W, int(rng.integers(0, 1 << 30))).astype(np.float64)
    cons = list(DEFAULT_CONSTRAINTS)
    if rng.random() < 0.7:
        cons[0] = float(rng.uniform(1.5, 7.0))
        cons[1] = float(rng.uniform(2.0, 14.0))
        cons[2] = float(rng.uniform(2.0, 14.0))
        cons[5] = float(rng.uniform(0.6, 1.6))
    dt = float(rng.choice([0.01, 0.02, 0.005]))
    dd = float(rng.choice([0.005, 0.003, 0.011]))
    res = gen.profile(torch.tensor(wp, device="cuda:0"), cons, dd=dd, capacity=int(64 / dd))
    tp = gen.time_profile(res, cons, dt=dt, capacity_rows=8192)
    torch.cuda.synchronize()
    counts = tp["counts"].cpu().numpy()
    rows_g = tp["rows"].cpu().numpy()
    nm_g = tp["nodes_map"].cpu().numpy()
    for b in range(B):
        rows, nmap, _ = oracle.OraclePath(wp[b]).generate_motion_profile(cons, dt=dt, dd=dd)
        n += 1
        T = rows.shape[0]
        ok = int(counts[b, 0]) == T and [int(v) for v in nm_g[b, :counts[b, 1]]] == [int(v) for v in nmap]
        e = np.inf
        if ok:
            e = np.max(np.abs(rows_g[b, :T] - rows) / np.maximum(np.abs(rows), 1.0)) if T else 0.0
        worst = max(worst, e)
        if not (ok and e <= 1e-6):
            fails += 1
            print(f"MISMATCH err {e:.2e} rows {int(counts[b, 0])} vs {T} W={W} cons={[round(c, 3) for c in cons]} dt={dt} dd={dd}", flush=True)
print(f"{n} paths in {time.time() - t0:.0f} s, {fails} mismatches; worst {worst:.2e}")
sys.exit(1 if fails else 0)
