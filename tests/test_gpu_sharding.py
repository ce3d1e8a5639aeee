"""GPU-side evidence for the multi-GPU layer on a ONE-GPU box (paths are independent, the batch is cut
into contiguous blocks, one per rank, with no data-path collective):

  * single process: the HIP path on the shard_bounds blocks of one batch (2 even halves, 8 uneven shards) gives,
    concatenated, the unsharded run bit for bit — rows, meta and path_summaries;
  * two processes sharing cuda:0 (gloo moves host copies; what is under test is the shard arithmetic, the
    dataclass / sequence forms of the constraints, the uneven p2p scatter / gather and the padded all-gather of
    vexautonomousplanner_amd.dist wrapped around the real HIP compute): gathered rows == single-process rows.
No 1 -> 8 GPU curve exists yet (the driver's scaling run was skipped in round 1); this is what a one-GPU box can show.
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS = ("x", "y", "heading", "curvature", "velocity")


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("n_paths,W,S,world", [(64, 32, 10000, 2), (203, 8, 1024, 8), (9, 5, 257, 8)])
def test_shards_concatenate_to_the_unsharded_result(torch_mod, dtype, n_paths, W, S, world):
    torch = torch_mod
    from vexautonomousplanner_amd import dist as vd
    from vexautonomousplanner_amd.batch import BatchedTrajectoryGenerator
    from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints
    gen = BatchedTrajectoryGenerator(0, dtype)
    wp = torch.tensor(make_waypoints(n_paths, W, 17), dtype=gen.tdtype, device=gen.device)
    full = {k: v.clone() for k, v in gen.profile(wp, DEFAULT_CONSTRAINTS, samples=S).items()}
    full_summ = vd.path_summaries(full["meta"], full["velocity"])
    torch.cuda.synchronize()
    sizes = []
    parts = {k: [] for k in FIELDS + ("meta", "flags")}
    summ = []
    for r in range(world):
        lo, hi = vd.shard_bounds(n_paths, r, world)
        sizes.append(hi - lo)
        if hi == lo:
            continue
        out = gen.profile(wp[lo:hi].contiguous(), DEFAULT_CONSTRAINTS, samples=S)
        for k in parts:
            parts[k].append(out[k].clone())
        summ.append(vd.path_summaries(out["meta"], out["velocity"]))
    torch.cuda.synchronize()
    assert sum(sizes) == n_paths and max(sizes) - min(sizes) <= 1
    for k in parts:
        assert torch.equal(torch.cat(parts[k]), full[k]), k
    # (length and sample count are copies; the traversal time is a per-path fp64 sum whose association torch picks
    # by tensor shape, so shards agree with the unsharded batch to rounding, not to the bit)
    got = torch.cat(summ)
    assert torch.equal(got[:, :2], full_summ[:, :2])
    assert float(((got[:, 2] - full_summ[:, 2]).abs() / full_summ[:, 2]).max()) <= 1e-14
    assert int(full["flags"].abs().max().item()) == 0


def _rank_worker(rank, world_size, port, n_paths, W, S, dtype, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world_size)     # before the first GPU call
    from vexautonomousplanner_amd import dist as vd
    from vexautonomousplanner_amd.batch import BatchedTrajectoryGenerator
    from vexautonomousplanner_amd.motion_profiling_v2.motion_profile_generator import Constraints
    from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints
    dev = torch.device("cuda", 0)
    # rank 0 owns the constraints as the drop-in dataclass, the others receive them
    cons = vd.broadcast_constraints(Constraints(*DEFAULT_CONSTRAINTS) if rank == 0 else None, dev)
    assert cons == [float(v) for v in DEFAULT_CONSTRAINTS]
    gen = BatchedTrajectoryGenerator(0, dtype)
    full = torch.tensor(make_waypoints(n_paths, W, 23), dtype=gen.tdtype, device=dev) if rank == 0 else None
    mine = vd.scatter_waypoints(full, n_paths, W, gen.tdtype, dev)
    lo, hi = vd.shard_bounds(n_paths, rank, world_size)
    assert mine.shape == (hi - lo, W, 2)
    out = gen.profile(mine, cons, samples=S)
    summ = vd.all_gather_rows(vd.path_summaries(out["meta"], out["velocity"]), n_paths)
    rows = vd.gather_rows_to_root(out["velocity"], n_paths)
    torch.cuda.synchronize()
    np.save(os.path.join(out_dir, f"summ{rank}.npy"), summ.cpu().numpy())
    if rank == 0:
        np.save(os.path.join(out_dir, "vel.npy"), rows.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_two_ranks_on_one_gpu_equal_the_single_process_run(torch_mod, tmp_path, dtype):
    torch = torch_mod
    import torch.multiprocessing as mp
    from vexautonomousplanner_amd import dist as vd
    from vexautonomousplanner_amd.batch import BatchedTrajectoryGenerator
    from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints
    n_paths, W, S = 37, 8, 2000          # odd: the shards differ by one path
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_rank_worker, args=(2, port, n_paths, W, S, dtype, str(tmp_path)), nprocs=2, join=True)
    gen = BatchedTrajectoryGenerator(0, dtype)
    wp = torch.tensor(make_waypoints(n_paths, W, 23), dtype=gen.tdtype, device=gen.device)
    ref = gen.profile(wp, DEFAULT_CONSTRAINTS, samples=S)
    ref_summ = vd.path_summaries(ref["meta"], ref["velocity"]).cpu().numpy()
    vel = np.load(tmp_path / "vel.npy")
    assert np.array_equal(vel, ref["velocity"].cpu().numpy())
    s0, s1 = np.load(tmp_path / "summ0.npy"), np.load(tmp_path / "summ1.npy")
    assert np.array_equal(s0, s1) and np.array_equal(s0[:, :2], ref_summ[:, :2])
    np.testing.assert_allclose(s0[:, 2], ref_summ[:, 2], rtol=1e-14)


def test_bench_starts_its_own_ranks(torch_mod):
    """`python bench.py --gpus 2` without a launcher: the script starts the two ranks itself as a child
    torch.distributed.run (rehearsed here on one GPU: gloo, both ranks on cuda:0), prints rank 0's one JSON line and
    returns the launcher's exit code."""
    import json
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-device", "--steps", "3",
           "--warmup", "1", "--paths-per-gpu", "256", "--no-cpu-baseline", "--parity-paths", "0"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["global_paths"] == 512 and rec["value"] > 0
    # the communicator's own record: the process group saw two ranks and both took part in a collective
    assert rec["comm"]["world_size"] == 2 and rec["comm"]["ranks_reporting"] == 2 and rec["comm"]["backend"] == "gloo"
    # a rank that fails takes the exit code with it
    bad = subprocess.run(cmd[:-5] + ["--paths-per-gpu", "-5", "--no-cpu-baseline", "--parity-paths", "0"], capture_output=True,
                         text=True, timeout=600, env=env)
    assert bad.returncode != 0
    assert not [ln for ln in bad.stdout.splitlines() if ln.startswith("{")]


def test_bench_one_rank_over_rccl_under_the_launcher(torch_mod):
    """The driver's multi-GPU command with one rank: `python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1`.
    Under a launcher bench.py sets the process group up whatever the world size, so this takes the `nccl` backend —
    RCCL on ROCm — through every call the N-rank run makes (init with a device id, the constraints broadcast, the two
    barriers around the timed region, the max-over-ranks all-reduce, the padded summary all-gather, the sweep's
    reductions) on the one GPU this box has."""
    import json
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--workload", "c5", "--paths-per-gpu", "2048", "--tolerance-sweep", "--no-cpu-baseline", "--parity-paths", "0"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["config"]["global_paths"] == 2048 and rec["value"] > 0
    assert rec["config"]["flags_or"] == 0
    assert rec["comm"]["backend"].startswith("rccl") and rec["comm"]["world_size"] == 1 and rec["comm"]["ranks_reporting"] == 1
    assert rec["comm"]["rccl"] not in (None, "unknown"), rec["comm"]
