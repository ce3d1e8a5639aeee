#!/usr/bin/env python3
"""Developer tool (GPU box): randomized bit-identity sweep of the long-row velocity kernel (look-back between
super-chunks inside one launch, k_velocity_chase) against the sequential sweep and against its earlier super-round form.

Random long rows (more samples than the register-resident kernel takes: 20 481 ... 400 000), on the fixed-S grid or the
reference's dd grid (ragged rows: paths end in different super-chunks), sparse and dense grids (W = 2 ... 3 waypoints give
runs of samples on one table entry: the sign-aware backward step), random robots and start / end velocities, the three
arithmetic modes.  Every `relax` row must equal `seq_fast`'s bit for bit; every tenth case also runs `relax_rounds`.
tests/test_gpu_parity.py holds the curated cases; this looks for rare ones (a record read torn, a finality chain
accepted too early would show as a differing row or a NOCONVERGE flag).
  python tools/fuzz_long_rows.py [seconds]
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
rng = np.random.default_rng(20261005)
gens = {}


def gen(dt, kernel):
    key = (dt, kernel)
    if key not in gens:
        if dt == "f32r32":
            gens[key] = BatchedTrajectoryGenerator(0, "f32", recurrence="f32", velocity_kernel=kernel)
        else:
            gens[key] = BatchedTrajectoryGenerator(0, dt, velocity_kernel=kernel)
    return gens[key]


def bits(t):
    return t.view(torch.int32 if t.dtype == torch.float32 else torch.int64)


fails = n_cases = n_dense = n_ragged = 0
t0 = time.time()
while time.time() - t0 < budget:
    W = int(rng.choice([2, 3, 4, 8, 32, 100, 256]))
    S = int(rng.choice([20481, 20992, 24577, 30001, 45000, 65536, 100003, 400000]))
    B = int(rng.integers(1, 7)) if S <= 65536 else int(rng.integers(1, 3))
    seed = int(rng.integers(0, 1 << 30))
    cons = list(DEFAULT_CONSTRAINTS)
    if rng.random() < 0.5:
        cons[0] = float(rng.uniform(1.0, 8.0))
        cons[1] = float(rng.uniform(2.0, 16.0))
        cons[2] = float(rng.uniform(2.0, 16.0))
        cons[5] = float(rng.uniform(0.5, 2.0))
    sv, ev = (0.01, 0.01) if rng.random() < 0.7 else (float(rng.uniform(0.0, 2.0)), float(rng.uniform(0.0, 2.0)))
    wp64 = make_waypoints(B, W, seed).astype(np.float64)
    kw = dict(samples=S)
    if rng.random() < 0.3:
        # the reference's own grid: rows of different lengths, all beyond the register-resident kernel
        scale = rng.uniform(0.6, 1.0, size=(B, 1, 1))
        wp64 = wp64 * scale
        seg = np.linalg.norm(np.diff(wp64, axis=1), axis=2).sum(axis=1)          # chord length <= arc length
        dd = float(seg.min() / 21000.0)
        cap = int(1.6 * seg.max() / dd) + 4096
        if cap <= 450000:
            kw = dict(dd=dd, capacity=cap)
            n_ragged += 1
    what = f"B={B} W={W} {kw} sv={sv:.3f} ev={ev:.3f} seed={seed} cons={[round(c, 3) for c in cons]}"
    for dt in ("f32", "f64", "f32r32"):
        wp = torch.tensor(wp64, device="cuda:0", dtype=torch.float64 if dt == "f64" else torch.float32)
        ref = {k: v.clone() for k, v in gen(dt, "seq_fast").profile(wp, cons, start_vel=sv, end_vel=ev, **kw).items()}
        kinds = ("relax", "relax_rounds") if n_cases % 10 == 0 else ("relax",)
        for kern in kinds:
            got = gen(dt, kern).profile(wp, cons, start_vel=sv, end_vel=ev, **kw)
            if int(got["flags"].abs().sum().item()) != int(ref["flags"].abs().sum().item()):
                fails += 1
                print(f"FAIL {dt} {kern} flags {got['flags'].tolist()} vs {ref['flags'].tolist()}: {what}", flush=True)
            if not torch.equal(bits(got["velocity"]), bits(ref["velocity"])):
                fails += 1
                print(f"FAIL {dt} {kern} velocity differs from seq_fast: {what}", flush=True)
        if dt == "f64":
            k = ref["curvature"]
            if float((k[:, 1:] == k[:, :-1]).double().mean().item()) > 0.3:
                n_dense += 1
    torch.cuda.synchronize()
    n_cases += 1
    if n_cases % 50 == 0:
        print(f"... {n_cases} cases, {fails} failures, {time.time() - t0:.0f} s", flush=True)
print(f"{n_cases} cases ({n_dense} on dense grids, {n_ragged} ragged) in {time.time() - t0:.0f} s, {fails} failures")
sys.exit(1 if fails else 0)
