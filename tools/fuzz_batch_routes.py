#!/usr/bin/env python3
"""Developer tool (GPU box): randomised check of the batched route path (profile -> apply_node_limits -> time_profile ->
insert_waits) against the oracle's generate_motion_profile on routes with random limits, stops, waits and action points
(no reverse / turn nodes), fp64.   python tools/fuzz_batch_routes.py [seconds] [seed]"""
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
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
gen = BatchedTrajectoryGenerator(0, "f64")
t0 = time.time()
n = fails = 0
worst_v = worst_r = 0.0
while time.time() - t0 < budget:
    B = int(rng.integers(1, 9))
    W = int(rng.choice([3, 4, 5, 8, 13]))
    cons = list(DEFAULT_CONSTRAINTS)
    if rng.random() < 0.5:
        cons[0], cons[1], cons[2], cons[5] = (float(rng.uniform(1.5, 7.0)), float(rng.uniform(2.0, 14.0)),
                                              float(rng.uniform(2.0, 14.0)), float(rng.uniform(0.6, 1.6)))
    wp = make_waypoints(B, W, int(rng.integers(0, 1 << 30))).astype(np.float64)
    mv = np.where(rng.random((B, W)) < 0.3, rng.uniform(1.0, 5.0, (B, W)), 0.0)
    ma = np.where(rng.random((B, W)) < 0.3, rng.uniform(2.0, 30.0, (B, W)), 0.0)
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
                     "max_acceleration": float(rng.uniform(2.0, 20.0)) if rng.random() < 0.4 else 0.0,
                     "stop": bool(rng.random() < 0.3), "wait_time": float(rng.uniform(0.02, 0.3)) if rng.random() < 0.4 else 0.0}
                    for t in np.sort(ts)])
    res = gen.profile(torch.tensor(wp, device="cuda:0", dtype=torch.float64), cons, dd=0.005, capacity=16384)
    gen.apply_node_limits(res, cons, node_max_velocity=mv, node_stop=stop, node_max_acceleration=ma, action_points=aps)
    tp = gen.time_profile(res, cons, dt=0.01, capacity_rows=8192)
    out = gen.insert_waits(res, tp, node_wait_time=wait, action_points=aps, dt=0.01)
    torch.cuda.synchronize()
    vel = res["velocity"].cpu().numpy()
    rows_all = out["rows"].cpu().numpy()
    counts = out["counts"].cpu().numpy()
    nmap_all, amap_all = out["nodes_map"].cpu().numpy(), out["actions_map"].cpu().numpy()
    for b in range(B):
        nodes = dict(is_reverse=np.zeros(W), turn=np.zeros(W), stop=stop[b].astype(float), wait_time=wait[b], max_velocity=mv[b],
                     max_acceleration=ma[b], tangent=np.full((W, 2), np.nan), magnitudes=np.zeros((W, 2)))
        al = aps[b]
        actions = dict(t=np.array([a["t"] for a in al]), stop=np.array([float(a["stop"]) for a in al]),
                       wait_time=np.array([a["wait_time"] for a in al]), max_velocity=np.array([a["max_velocity"] for a in al]),
                       max_acceleration=np.array([a["max_acceleration"] for a in al])) if al else None
        op = oracle.OraclePath(wp[b], nodes=nodes, actions=actions)
        op.rebuild_tables()
        v_ref = op.forward_backward(cons, dd=0.005)["velocity"]
        try:
            r_ref, n_ref, a_ref = op.generate_motion_profile(cons, dt=0.01, dd=0.005)
        except ValueError:
            continue
        n += 1
        N = len(v_ref)
        ev = np.max(np.abs(vel[b, :N] - v_ref) / v_ref) if int(res["meta"][b, 3]) == N else np.inf
        T, nn, na = (int(x) for x in counts[b])
        ok = T == r_ref.shape[0] and list(nmap_all[b, :nn]) == [int(x) for x in n_ref] and list(amap_all[b, :na]) == [int(x) for x in a_ref]
        er = np.max(np.abs(rows_all[b, :T] - r_ref) / np.maximum(np.abs(r_ref), 1.0)) if ok else np.inf
        worst_v, worst_r = max(worst_v, ev), max(worst_r, er)
        if not (ev <= 1e-7 and er <= 1e-6):      # (fp64; the angular term is ill-conditioned at max_acceleration ~ 25)
            fails += 1
            print(f"MISMATCH velocity {ev:.2e} rows {er:.2e} T {T} vs {r_ref.shape[0]} | W={W} cons={[round(c, 3) for c in cons]} mv={mv[b].round(2).tolist()} "
                  f"ma={ma[b].round(2).tolist()} stop={stop[b].astype(int).tolist()} wait={wait[b].round(3).tolist()} aps={al} seed_wp={wp[b].round(4).tolist()}", flush=True)
print(f"{n} routes in {time.time() - t0:.0f} s, {fails} mismatches; worst velocity {worst_v:.2e}, worst rows {worst_r:.2e}")
sys.exit(1 if fails else 0)
