#!/usr/bin/env python3
"""Developer tool (GPU box): randomised check of the batched path for routes WITH reverse / turn nodes
(profile_routes -> apply_node_limits -> time_profile(node_reverse) -> insert_waits(node_turn, ...)) against the oracle's
generate_motion_profile: random robots, splits, tangent overrides, limits, stops, waits, action points; every route of a
batch is different.  fp64.    python tools/fuzz_batch_split_routes.py [seconds] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import oracle
from vexautonomousplanner_amd.batch import BatchedTrajectoryGenerator
from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 9)
gen = BatchedTrajectoryGenerator(0, "f64")
t0 = time.time()
n = fails = skipped = 0
worst_v = worst_r = 0.0
while time.time() - t0 < budget:
    B = int(rng.integers(1, 9))
    W = int(rng.choice([3, 4, 5, 6, 8, 13]))
    cons = list(DEFAULT_CONSTRAINTS)
    if rng.random() < 0.5:
        cons[0], cons[1], cons[2], cons[5] = (float(rng.uniform(1.5, 7.0)), float(rng.uniform(2.0, 14.0)),
                                              float(rng.uniform(2.0, 14.0)), float(rng.uniform(0.6, 1.6)))
    wp = make_waypoints(B, W, int(rng.integers(0, 1 << 30))).astype(np.float64)
    rev = rng.random((B, W)) < 0.25
    turn = np.where(rng.random((B, W)) < 0.25, rng.choice([-170.0, -120.0, -45.0, 30.0, 90.0, 175.0, 360.0], size=(B, W)), 0.0)
    rev[:, -1] = False
    turn[:, -1] = 0.0
    turn[:, 0] = 0.0          # (a turn at node 0 is the reference's IndexError: covered by test_bad_routes_are_flagged)
    tan = np.full((B, W, 2), np.nan)
    mag = np.zeros((B, W, 2))
    for b in range(B):
        for k in range(W):
            if rng.random() < 0.15:
                a = rng.uniform(0, 2 * np.pi)
                tan[b, k] = (np.cos(a), np.sin(a))
                mag[b, k] = rng.uniform(0.3, 1.0, size=2)
    mv = np.where(rng.random((B, W)) < 0.3, rng.uniform(1.0, 5.0, (B, W)), 0.0)
    ma = np.where(rng.random((B, W)) < 0.2, rng.uniform(2.0, 20.0, (B, W)), 0.0)
    stop = rng.random((B, W)) < 0.2
    stop[:, 0] = stop[:, -1] = False
    wait = np.where(rng.random((B, W)) < 0.25, rng.uniform(0.02, 0.5, (B, W)), 0.0)
    wait[:, -1] = 0
    aps = []
    for b in range(B):
        k = int(rng.integers(0, 4))
        ts = np.sort(rng.uniform(0.1, W - 1.1, size=k))
        if k and rng.random() < 0.15:
            ts[0] = float(rng.integers(1, W - 1))                       # exactly on a node
        aps.append([{"t": float(t), "max_velocity": float(rng.uniform(1.0, 4.0)) if rng.random() < 0.4 else 0.0,
                     "max_acceleration": float(rng.uniform(2.0, 20.0)) if rng.random() < 0.3 else 0.0,
                     "stop": bool(rng.random() < 0.3), "wait_time": float(rng.uniform(0.02, 0.3)) if rng.random() < 0.4 else 0.0}
                    for t in np.sort(ts)])
    res = gen.profile_routes(torch.tensor(wp, device="cuda:0", dtype=torch.float64), node_reverse=rev, node_turn=turn,
                             node_tangent=tan, node_magnitudes=mag, constraints=cons, dd=0.005, capacity=16384)
    gen.apply_node_limits(res, cons, node_max_velocity=mv, node_stop=stop, node_max_acceleration=ma, action_points=aps)
    tp = gen.time_profile(res, cons, dt=0.01, capacity_rows=8192, node_reverse=rev)
    out = gen.insert_waits(res, tp, node_wait_time=wait, action_points=aps, dt=0.01, node_turn=turn, node_reverse=rev,
                           constraints=cons)
    torch.cuda.synchronize()
    flags = res["flags"].cpu().numpy()
    vel = res["velocity"].cpu().numpy()
    rows_all = out["rows"].cpu().numpy()
    counts = out["counts"].cpu().numpy()
    nmap_all, amap_all = out["nodes_map"].cpu().numpy(), out["actions_map"].cpu().numpy()
    for b in range(B):
        nodes = dict(is_reverse=rev[b].astype(float), turn=turn[b], stop=stop[b].astype(float), wait_time=wait[b], max_velocity=mv[b],
                     max_acceleration=ma[b], tangent=tan[b], magnitudes=mag[b])
        al = aps[b]
        actions = dict(t=np.array([a["t"] for a in al]), stop=np.array([float(a["stop"]) for a in al]),
                       wait_time=np.array([a["wait_time"] for a in al]), max_velocity=np.array([a["max_velocity"] for a in al]),
                       max_acceleration=np.array([a["max_acceleration"] for a in al])) if al else None
        try:
            op = oracle.OraclePath(wp[b], nodes=nodes, actions=actions)
            op.rebuild_tables()
            v_ref = op.forward_backward(cons, dd=0.005)["velocity"]
            r_ref, n_ref, a_ref = op.generate_motion_profile(cons, dt=0.01, dd=0.005)
        except ValueError:
            skipped += 1
            continue
        n += 1
        N = len(v_ref)
        ev = np.max(np.abs(vel[b, :N] - v_ref) / v_ref) if (int(res["meta"][b, 3]) == N and flags[b] == 0) else np.inf
        T, nn, na = (int(x) for x in counts[b])
        ok = T == r_ref.shape[0] and list(nmap_all[b, :nn]) == [int(x) for x in n_ref] and list(amap_all[b, :na]) == [int(x) for x in a_ref]
        er = np.max(np.abs(rows_all[b, :T] - r_ref) / np.maximum(np.abs(r_ref), 1.0)) if ok else np.inf
        worst_v, worst_r = max(worst_v, ev), max(worst_r, er)
        if not (ev <= 1e-7 and er <= 1e-6):      # (fp64; the angular term is ill-conditioned at max_acceleration ~ 25)
            fails += 1
            if fails <= 12:
                print(f"MISMATCH velocity {ev:.2e} rows {er:.2e} T {T} vs {r_ref.shape[0]} flags {flags[b]} maps {ok} | W={W} cons={[round(c, 3) for c in cons]} "
                      f"rev={rev[b].astype(int).tolist()} turn={turn[b].tolist()} tan={np.round(tan[b], 3).tolist()} mag={np.round(mag[b], 3).tolist()} "
                      f"mv={mv[b].round(2).tolist()} ma={ma[b].round(2).tolist()} stop={stop[b].astype(int).tolist()} wait={wait[b].round(3).tolist()} "
                      f"aps={al} wp={wp[b].round(4).tolist()}", flush=True)
print(f"{n} routes in {time.time() - t0:.0f} s ({skipped} the oracle refused), {fails} mismatches; worst velocity {worst_v:.2e}, worst rows {worst_r:.2e}")
sys.exit(1 if fails else 0)
