#!/usr/bin/env python3
"""Developer tool (GPU box): k_time_integrate_quad (four lanes per path, VAP_TIME_KERNEL_QUAD) and k_time_fused (the same with the geometry in
the workgroup, VAP_TIME_KERNEL_FUSED) against the lane-per-path
kernel (VAP_TIME_KERNEL_LANE) — random batches through both; rows, counts and maps must be the same bits; then the
config-3 timing of each.
  python tools/ab_time_quad.py [cases]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from vexautonomousplanner_amd.batch import BatchedTrajectoryGenerator
from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(11)
bad = n = 0


def same(a, b):
    ca, cb = a["counts"].cpu().numpy(), b["counts"].cpu().numpy()
    if not np.array_equal(ca, cb):
        return False
    ra, rb = a["rows"].cpu().numpy(), b["rows"].cpu().numpy()
    na, nb = a["nodes_map"].cpu().numpy(), b["nodes_map"].cpu().numpy()
    for p in range(ca.shape[0]):
        if ra[p, :ca[p, 0]].tobytes() != rb[p, :ca[p, 0]].tobytes() or not np.array_equal(na[p, :ca[p, 1]], nb[p, :ca[p, 1]]):
            return False
    return True


for dtype, resid in (("f32", True), ("f32", False), ("f64", False)):
    gen = BatchedTrajectoryGenerator(0, dtype, time_domain_residual=resid)
    td = torch.float64 if dtype == "f64" else torch.float32
    for case in range(cases):
        B = int(rng.choice([1, 3, 5, 16, 17, 63, 200]))
        W = int(rng.integers(2, 12))
        wp = torch.tensor(make_waypoints(B, W, int(rng.integers(0, 1 << 30))), device="cuda:0", dtype=td)
        cons = list(DEFAULT_CONSTRAINTS)
        cons[0] = float(rng.uniform(1.5, 7.0)); cons[1] = float(rng.uniform(2.0, 14.0)); cons[2] = float(rng.uniform(2.0, 14.0))
        dt = float(rng.choice([0.01, 0.02, 0.005])); dd = float(rng.choice([0.005, 0.003, 0.011]))
        res = gen.profile(wp, cons, dd=dd, capacity=int(64 / dd))
        cap = 300 if case % 8 == 0 else 8192     # (300: truncated paths)
        out = {}
        for k in ("lane", "quad", "fused"):
            gen.set_time_kernel(k)
            out[k] = gen.time_profile(res, cons, dt=dt, capacity_rows=cap)
        n += 1
        if not (same(out["lane"], out["quad"]) and same(out["lane"], out["fused"])):
            bad += 1
            print("DIFFER", dtype, resid, case, B, W, dt, dd, flush=True)
    wp = torch.tensor(make_waypoints(4096, 32, 3), device="cuda:0", dtype=td)
    res = gen.profile(wp, DEFAULT_CONSTRAINTS, samples=10000)
    out = {}
    for k in ("lane", "quad", "fused", "lane", "quad", "fused"):
        gen.set_time_kernel(k)
        tp = gen.time_profile(res, DEFAULT_CONSTRAINTS, capacity_rows=2048)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(5):
            tp = gen.time_profile(res, DEFAULT_CONSTRAINTS, capacity_rows=2048, out=tp)
        ev[1].record()
        torch.cuda.synchronize()
        out[k] = tp
        print(f"config 3 {dtype} residual={resid} {k}: {ev[0].elapsed_time(ev[1]) / 5:.3f} ms", flush=True)
    n += 1
    if not (same(out["lane"], out["quad"]) and same(out["lane"], out["fused"])):
        bad += 1
        print("DIFFER config 3", dtype, resid, flush=True)
print(f"{n} batches, {bad} differ")
sys.exit(1 if bad else 0)
