"""Multi-GPU layer: one process per GPU, paths sharded by contiguous blocks, RCCL over xGMI.

Paths are independent: there is NO collective on the data path.  RCCL
(`torch.distributed` backend "nccl" on ROCm) is used only to
  * broadcast the 6 constraint values (and optionally scatter waypoint shards) from rank 0, and
  * all-gather small per-path summaries (length, sample count, traversal time) for candidate
    ranking, or gather whole result rows to rank 0 when a caller really wants them there.
Everything here is backend-agnostic, so the CPU test-suite exercises it with gloo.
"""
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def _comm_device(t: torch.Tensor) -> torch.device:
    """gloo moves host memory; RCCL moves device memory."""
    return torch.device("cpu") if dist.get_backend() == "gloo" else t.device


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(n_paths: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of rank `rank`; the first n_paths % world_size ranks get one more."""
    base, extra = divmod(n_paths, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _six(constraints):
    """A Constraints-like object (the drop-in dataclass, MPG:14-21 field order) or a 6-sequence -> 6 floats."""
    if hasattr(constraints, "max_vel"):
        c = constraints
        return [float(c.max_vel), float(c.max_acc), float(c.max_dec), float(c.friction_coef), float(c.max_jerk),
                float(c.track_width)]
    vals = [float(v) for v in constraints]
    if len(vals) != 6:
        raise ValueError("constraints need 6 values: max_vel, max_acc, max_dec, friction_coef, max_jerk, track_width")
    return vals


def broadcast_constraints(constraints, device, src: int = 0):
    """Rank `src` owns the constraints; every rank returns the same 6 floats."""
    rank, ws = world()
    vals = _six(constraints) if (rank == src and constraints is not None) else [0.0] * 6
    t = torch.tensor(vals, dtype=torch.float64, device=device)
    if ws > 1:
        tc = t.to(_comm_device(t))
        dist.broadcast(tc, src=src)
        t = tc
    return [float(v) for v in t.cpu()]


def scatter_waypoints(waypoints: Optional[torch.Tensor], n_paths: int, n_waypoints: int, dtype, device,
                      src: int = 0) -> torch.Tensor:
    """Rank `src` holds the (n_paths, W, 2) batch; each rank receives its contiguous shard."""
    rank, ws = world()
    lo, hi = shard_bounds(n_paths, rank, ws)
    if ws == 1:
        return waypoints[lo:hi].to(device=device, dtype=dtype).contiguous()
    mine = torch.empty((hi - lo, n_waypoints, 2), dtype=dtype, device=device)
    if rank == src:
        pieces = []
        for r in range(ws):
            a, b = shard_bounds(n_paths, r, ws)
            pieces.append(waypoints[a:b].to(device=device, dtype=dtype).contiguous())
        # shards may differ by one path: send/recv pairs instead of dist.scatter's equal-size rule
        reqs = [dist.isend(pieces[r], dst=r) for r in range(ws) if r != src]
        mine.copy_(pieces[src])
        for q in reqs:
            q.wait()
    else:
        dist.recv(mine, src=src)
    return mine


def all_gather_rows(local: torch.Tensor, n_paths: int) -> torch.Tensor:
    """All ranks obtain the (n_paths, ...) concatenation of their per-path rows, in path order."""
    rank, ws = world()
    if ws == 1:
        return local
    sizes = [shard_bounds(n_paths, r, ws) for r in range(ws)]
    width = max(b - a for a, b in sizes)
    cdev = _comm_device(local)
    pad = torch.zeros((width,) + tuple(local.shape[1:]), dtype=local.dtype, device=cdev)
    pad[: local.shape[0]] = local.to(cdev)
    bufs = [torch.empty_like(pad) for _ in range(ws)]
    dist.all_gather(bufs, pad)
    return torch.cat([bufs[r][: sizes[r][1] - sizes[r][0]] for r in range(ws)]).to(local.device)


def gather_rows_to_root(local: torch.Tensor, n_paths: int, dst: int = 0) -> Optional[torch.Tensor]:
    """Whole result rows to rank `dst` over point-to-point links (7 peers, each on its own xGMI link)."""
    rank, ws = world()
    if ws == 1:
        return local
    if rank == dst:
        out = torch.empty((n_paths,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        reqs = []
        for r in range(ws):
            a, b = shard_bounds(n_paths, r, ws)
            if r == dst:
                out[a:b].copy_(local)
            else:
                reqs.append(dist.irecv(out[a:b], src=r))
        for q in reqs:
            q.wait()
        return out
    dist.send(local.contiguous(), dst=dst)
    return None


def path_summaries(meta: torch.Tensor, velocity: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(B,3) fp64 per path: arc length, sample count, traversal time = sum dd / v (0 if no velocities)."""
    out = torch.zeros((meta.shape[0], 3), dtype=torch.float64, device=meta.device)
    out[:, 0] = meta[:, 1]
    out[:, 1] = meta[:, 3]
    if velocity is not None:
        v = velocity.to(torch.float64)
        n = meta[:, 3].to(torch.int64)
        idx = torch.arange(v.shape[1], device=v.device)[None, :]
        valid = idx < (n[:, None] - 1)
        vm = torch.where(valid, 0.5 * (v + torch.roll(v, -1, dims=1)), torch.ones_like(v))
        out[:, 2] = torch.where(valid, meta[:, 2:3] / vm, torch.zeros_like(v)).sum(dim=1)
    return out


def comm_record(device) -> dict:
    """What the communicator itself says about the run, for bench.py's JSON line: backend, the world size the process
    group reports, the RCCL version in use, and `ranks_reporting` = an all-reduce (sum) of one 1 per rank over that
    backend — N on an N-rank run only if every rank really took part in a collective."""
    if not (dist.is_available() and dist.is_initialized()):
        return {"backend": None, "world_size": 1, "rccl": None, "ranks_reporting": 1}
    backend = dist.get_backend()
    one = torch.ones(1, dtype=torch.int64, device=torch.device("cpu") if backend == "gloo" else device)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    rccl = None
    if backend == "nccl":
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:       # (a torch build without the query: the record says so rather than failing the bench)
            rccl = "unknown"
    return {"backend": "rccl (torch.distributed 'nccl')" if backend == "nccl" else backend, "world_size": dist.get_world_size(),
            "rccl": rccl, "ranks_reporting": int(one.item())}
