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


def _worker8(rank, world_size, port, n_paths, W, S, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    from oracle import oracle
    from vexautonomousplanner_amd import dist as vd
    from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints
    rec = vd.comm_record(torch.device("cpu"))
    assert rec["world_size"] == world_size and rec["ranks_reporting"] == world_size and rec["backend"] == "gloo"
    cons = vd.broadcast_constraints(DEFAULT_CONSTRAINTS if rank == 0 else None, torch.device("cpu"))
    full = torch.tensor(make_waypoints(n_paths, W, 7).astype(np.float64)) if rank == 0 else None
    mine = vd.scatter_waypoints(full, n_paths, W, torch.float64, torch.device("cpu"))
    lo, hi = vd.shard_bounds(n_paths, rank, world_size)
    assert mine.shape == (hi - lo, W, 2)
    r = oracle.profile_batch(mine.numpy(), S, cons, n_threads=1)
    meta = torch.zeros((hi - lo, 4), dtype=torch.float64)
    meta[:, 1] = torch.tensor(r["total_length"])
    meta[:, 2] = meta[:, 1] / (S - 1.5)
    meta[:, 3] = S
    vel = torch.tensor(r["velocity"])
    summ = vd.all_gather_rows(vd.path_summaries(meta, vel), n_paths)
    rows = vd.gather_rows_to_root(vel, n_paths)
    if rank == 0:
        np.save(os.path.join(out_dir, "vel.npy"), rows.numpy())
    np.save(os.path.join(out_dir, f"summ{rank}.npy"), summ.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_equals_unsharded_world8_uneven_shards(tmp_path):
    """Eight ranks (the node's GPU count) over gloo with 13 paths: shards of 2, 2, 2, 2, 2, 1, 1, 1 — broadcast,
    uneven point-to-point scatter, padded all-gather, gather to the root, and the communicator record bench.py prints
    (world size and an all-reduce of ones = 8)."""
    sys.path.insert(0, ROOT)
    from oracle import oracle
    from vexautonomousplanner_amd import dist as vd
    from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints
    n_paths, W, S, world = 13, 5, 120, 8
    assert [vd.shard_bounds(n_paths, r, world)[1] - vd.shard_bounds(n_paths, r, world)[0] for r in range(world)] == [2, 2, 2, 2, 2, 1, 1, 1]
    port = 31000 + (os.getpid() % 2000)
    mp.spawn(_worker8, args=(world, port, n_paths, W, S, str(tmp_path)), nprocs=world, join=True)
    ref = oracle.profile_batch(make_waypoints(n_paths, W, 7).astype(np.float64), S, DEFAULT_CONSTRAINTS)
    assert np.array_equal(np.load(tmp_path / "vel.npy"), ref["velocity"])
    s = [np.load(tmp_path / f"summ{r}.npy") for r in range(world)]
    assert all(np.array_equal(s[0], x) for x in s[1:])
    np.testing.assert_array_equal(s[0][:, 0], ref["total_length"])


def test_rank_blocks_of_the_global_batch():
    """bench.py: rank r generates block r of the seeded global batch on its own; block 0 is the 1-GPU batch."""
    sys.path.insert(0, ROOT)
    from vexautonomousplanner_amd.synth import make_waypoints, make_waypoints_block
    assert np.array_equal(make_waypoints_block(16, 8, 3, 0), make_waypoints(16, 8, 3))
    b1, b2 = make_waypoints_block(16, 8, 3, 1), make_waypoints_block(16, 8, 3, 2)
    assert b1.shape == (16, 8, 2) and not np.array_equal(b1, b2) and not np.array_equal(b1, make_waypoints(16, 8, 3))
    assert np.array_equal(b1, make_waypoints_block(16, 8, 3, 1))       # deterministic
    d = np.linalg.norm(np.diff(b1.astype(np.float64), axis=1), axis=2)
    assert d.min() >= 0.29 and d.max() <= 1.01                          # the same generator: steps of 0.3 .. 1.0 ft


def test_bench_self_launch_hands_a_failing_rank_through():
    """`python bench.py --gpus 2` with no launcher in the environment starts its own two ranks as a child
    torch.distributed.run.  Without a GPU the ranks cannot run (the product has no CPU path), so what this CPU test
    shows is the plumbing: the child is started, its failure comes back as a non-zero exit code, no JSON line."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--paths-per-gpu", "4", "--no-cpu-baseline", "--parity-paths", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    if torch.cuda.is_available():
        pytest.skip("a GPU box: tests/test_gpu_sharding.py::test_bench_starts_its_own_ranks covers the launch")
    assert p.returncode != 0
    assert "torch.distributed" in p.stderr or "ChildFailedError" in p.stderr or "rank" in p.stderr.lower()
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
