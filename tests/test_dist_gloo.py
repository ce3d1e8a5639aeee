"""World-size-2 run of the multi-GPU layer on CPU (gloo): the collectives' shapes, the shard
arithmetic and "sharded result == unsharded result" — the per-rank compute is stood in for by the
CPU oracle, since no GPU exists here."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world_size, port, n_paths, W, S, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    from oracle import oracle
    from vexautonomousplanner_amd import dist as vd
    from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints
    dev = torch.device("cpu")
    cons = vd.broadcast_constraints(DEFAULT_CONSTRAINTS if rank == 0 else None, dev)
    assert cons == [float(v) for v in DEFAULT_CONSTRAINTS]
    full = torch.tensor(make_waypoints(n_paths, W, 7).astype(np.float64)) if rank == 0 else None
    mine = vd.scatter_waypoints(full, n_paths, W, torch.float64, dev)
    lo, hi = vd.shard_bounds(n_paths, rank, world_size)
    assert mine.shape == (hi - lo, W, 2)
    r = oracle.profile_batch(mine.numpy(), S, cons)
    meta = torch.zeros((hi - lo, 4), dtype=torch.float64)
    meta[:, 1] = torch.tensor(r["total_length"])
    meta[:, 2] = meta[:, 1] / (S - 1.5)
    meta[:, 3] = S
    vel = torch.tensor(r["velocity"])
    summ = vd.all_gather_rows(vd.path_summaries(meta, vel), n_paths)
    assert summ.shape == (n_paths, 3)
    rows = vd.gather_rows_to_root(vel, n_paths)
    if rank == 0:
        np.save(os.path.join(out_dir, "vel.npy"), rows.numpy())
    np.save(os.path.join(out_dir, f"summ{rank}.npy"), summ.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_paths", [6, 7])
def test_sharded_equals_unsharded_world2(tmp_path, n_paths):
    sys.path.insert(0, ROOT)
    from oracle import oracle
    from vexautonomousplanner_amd import dist as vd
    from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints
    W, S = 8, 200
    port = 29000 + (os.getpid() % 2000) + n_paths
    mp.spawn(_worker, args=(2, port, n_paths, W, S, str(tmp_path)), nprocs=2, join=True)
    ref = oracle.profile_batch(make_waypoints(n_paths, W, 7).astype(np.float64), S, DEFAULT_CONSTRAINTS)
    vel = np.load(tmp_path / "vel.npy")
    assert np.array_equal(vel, ref["velocity"])          # paths are independent: bit-identical
    s0, s1 = np.load(tmp_path / "summ0.npy"), np.load(tmp_path / "summ1.npy")
    assert np.array_equal(s0, s1)
    np.testing.assert_array_equal(s0[:, 0], ref["total_length"])
    assert np.all(s0[:, 1] == S) and np.all(s0[:, 2] > 0)
    # shard arithmetic: contiguous, disjoint, covering
    cover = []
    for r in range(3):
        lo, hi = vd.shard_bounds(10, r, 3)
        cover += list(range(lo, hi))
    assert cover == list(range(10))
