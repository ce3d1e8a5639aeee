#!/usr/bin/env python3
"""bench.py — trajectory sample-points/s of the batched hot path on N MI355X (one rank per GPU).

    python bench.py [--gpus N --steps K --warmup W] [--workload c3|c5|c2|c4] [--dtype f32|f64]
        (N > 1 without a launcher: the script starts its own N ranks as a child torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path (fit -> arc-length LUT -> sampling -> forward/backward velocity
pass) over this rank's batch of synthetic paths, inputs already resident in HBM.  Paths are
independent, so the batch is sharded over ranks with NO data-path collective (weak scaling: every
rank processes the same number of paths); RCCL only broadcasts the constraints before and gathers
per-path summaries after the timed region.

Workloads (BASELINE.json configs):
  c3 (default): 4096 paths x 32 waypoints x 10 000 samples per GPU  — the config the metric
                "sample-points/s (batched paths)" is quoted on
  c5          : 131 072 paths x 8 waypoints x 1024 samples per GPU
  c2          : 1 path x 256 waypoints x 1 000 000 samples
  c4          : 8192 paths x 32 waypoints x 10 000 samples — one GPU's share of config 4's 65 536 paths
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    "c3": dict(paths=4096, W=32, S=10000, seed=3, name="c3: 4096 paths x 32 waypoints x 10000 samples per GPU"),
    "c5": dict(paths=131072, W=8, S=1024, seed=5, name="c5: 131072 paths x 8 waypoints x 1024 samples per GPU"),
    "c2": dict(paths=1, W=256, S=1000000, seed=2, name="c2: 1 path x 256 waypoints x 1e6 samples"),
    "c4": dict(paths=8192, W=32, S=10000, seed=4, name="c4: 65536 paths x 32 waypoints x 10000 samples over 8 GPUs, the share of one (8192 paths)"),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s


def cpu_baseline(wl, budget_s=12.0):
    """The fp64 C restatement (oracle/, kind "port") on this box's host cores, bounded sample: batches of
    paths from the same generator until about budget_s seconds of CPU work have been timed."""
    from oracle import oracle
    from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints
    # threads: the cores this process may run on, capped (a 1-GPU box shares its host)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 32))
    W, S = wl["W"], min(wl["S"], 100000)
    per_batch = max(cores, min(wl["paths"], 16 * cores))
    oracle.profile_batch(make_waypoints(cores, W, wl["seed"]).astype(np.float64), min(S, 2000), DEFAULT_CONSTRAINTS,
                         n_threads=cores, want=("velocity",))          # warm-up: library load, thread start
    done, elapsed, k = 0, 0.0, 0
    while elapsed < budget_s and k < 512:
        wp = make_waypoints(per_batch, W, wl["seed"] + 1000 * k).astype(np.float64)
        t0 = time.perf_counter()
        oracle.profile_batch(wp, S, DEFAULT_CONSTRAINTS, n_threads=cores, want=("velocity",))
        elapsed += time.perf_counter() - t0
        done += per_batch
        k += 1
    # the same restatement on ONE thread (single-threaded beside all cores), a few seconds of it
    done1, elapsed1 = 0, 0.0
    while elapsed1 < 2.5 and done1 < 4096:
        wp = make_waypoints(4, W, wl["seed"] + 777 + done1).astype(np.float64)
        t0 = time.perf_counter()
        oracle.profile_batch(wp, S, DEFAULT_CONSTRAINTS, n_threads=1, want=("velocity",))
        elapsed1 += time.perf_counter() - t0
        done1 += 4
    return {"value": done * S / elapsed, "unit": "sample-points/s", "cores": cores, "kind": "port",
            "sample": f"{done} paths x {W} waypoints x {S} samples of the same generator in {k} batches, "
                      f"{cores} threads, {elapsed:.1f}s of CPU work",
            "single_thread": {"value": done1 * S / elapsed1, "unit": "sample-points/s", "cores": 1,
                              "sample": f"{done1} paths, {elapsed1:.1f}s"}}


def parity_check(out, wp, S, constraints, n_paths, dtype):
    """Worst relative velocity error of the first n_paths paths against the oracle (the measures of
    tests/test_gpu_parity.py); `bound` is north_star's 1e-5 for fp32 rows, 1e-7 for fp64 rows."""
    from oracle import oracle
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    wp64 = wp[:n_paths].double().cpu().numpy()
    ref = oracle.profile_batch(wp64, S, constraints, n_threads=max(1, min(cores, 32)))
    bound = 1e-5 if dtype == "f32" else 1e-7
    res = {"checked_paths": int(n_paths), "bound": bound}
    v = out["velocity"][:n_paths].double().cpu().numpy()
    ev = np.max(np.abs(v - ref["velocity"]) / ref["velocity"], axis=1)
    res["frac_above_1e-5"] = float(np.mean(ev > 1e-5))
    res["frac_above_bound"] = float(np.mean(ev > bound))
    res["worst"] = float(ev.max())
    res["median_path_worst"] = float(np.median(ev))
    geo = {}
    for k, floor in (("curvature", 1e-2), ("x", 1.0), ("y", 1.0)):
        if k in out:
            g = out[k][:n_paths].double().cpu().numpy()
            geo[k] = float(np.max(np.abs(g - ref[k]) / np.maximum(np.abs(ref[k]), floor)))
    if "heading" in out:
        geo["heading"] = float(np.max(np.abs(out["heading"][:n_paths].double().cpu().numpy() - ref["heading"])) / np.pi)
    res["worst_geometry"] = geo
    return res


def tolerance_sweep(device_index, wp32, constraints, S):
    """Per-path worst relative velocity difference of the fp32-row modes against the fp64 run of the same batch
    (fp32-representable waypoints, so all modes see identical inputs).  Quantiles over the paths of this rank."""
    import torch
    from vexautonomousplanner_amd.batch import BatchedTrajectoryGenerator
    ref = BatchedTrajectoryGenerator(device_index, "f64").profile(wp32.double(), constraints, samples=S, want=("velocity",))["velocity"]
    out = {"reference": "this library's fp64 mode on the same batch", "measure": "max over a path's samples of |v - v64| / v64",
           "modes": {}}
    for name, rec in (("f32 rows, f64 recurrence (default)", "f64"), ("f32 rows, f32 recurrence", "f32")):
        v = BatchedTrajectoryGenerator(device_index, "f32", recurrence=rec).profile(wp32.float(), constraints, samples=S,
                                                                                     want=("velocity",))["velocity"]
        e = ((v.double() - ref).abs() / ref).amax(dim=1)
        q = torch.quantile(e, torch.tensor([0.5, 0.9, 0.99, 0.999], dtype=torch.float64, device=e.device))
        out["modes"][name] = {"paths": int(e.numel()), "median": float(q[0]), "p90": float(q[1]), "p99": float(q[2]),
                              "p999": float(q[3]), "worst": float(e.max()), "paths_above_1e-5": int((e > 1e-5).sum())}
        del v, e
    return out


def kernel_source_sha():
    """sha256 over the kernel sources the library is built from: a PMC profile is only quoted while it matches."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "vexautonomousplanner_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def committed_profile(workload, dtype, paths, recurrence):
    """The committed PMC profile of this round (profiles/r*_traffic.json, made by tools/profile_round.sh: separate
    FETCH_SIZE / WRITE_SIZE passes, gfx950 corrections applied) — only while it was taken on this workload and mode AND on
    these kernel sources (it records their hash); otherwise (None, reason): a stale figure is not quoted."""
    import glob
    import re

    def round_of(path):     # r02_traffic.json < r10_traffic.json: by number, not by text
        m = re.match(r"r(\d+)", os.path.basename(path))
        return int(m.group(1)) if m else -1
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), key=round_of)
    if not files:
        return None, "no PMC profile committed"
    # one file per profiled workload and round (r04_traffic.json = the headline config, r04_c4_traffic.json, ...): the
    # newest round's file for THIS workload and mode
    newest = round_of(files[-1])
    want = {"workload": workload, "dtype": dtype, "paths": paths, "recurrence": recurrence if dtype == "f32" else "f64"}
    for f in reversed([f for f in files if round_of(f) == newest]):
        prof = json.load(open(f))
        name = os.path.basename(f)
        have = prof.get("bench", {})
        if any(have.get(k) != v for k, v in want.items()):
            continue
        if prof.get("kernel_source_sha") != kernel_source_sha():
            return None, f"{name} is stale: the kernel sources changed since it was taken"
        return prof, name
    return None, f"round {newest}'s profiles were taken on other workloads / modes"


def profiled_traffic(stage, workload, dtype, paths, recurrence):
    """HBM bytes per launch of the stage's kernel from the committed profile, or (None, reason)."""
    prof, name = committed_profile(workload, dtype, paths, recurrence)
    if prof is None:
        return None, name
    kern = {"sample": "k_sample", "velocity": "k_velocity", "fit": "k_fit", "lut": "k_lut"}[stage]
    vals = [v["hbm_bytes"] for k, v in prof["kernels"].items() if kern in k]
    return (sum(vals), name) if vals else (None, f"{name} has no {kern} entry")


def profiled_pipeline_traffic(workload, dtype, paths, recurrence):
    """HBM bytes per step over all of the step's kernels (every vap:: kernel the profiled step launched, once each),
    from the same committed profile (None when stale)."""
    prof, _ = committed_profile(workload, dtype, paths, recurrence)
    if prof is None:
        return None
    vals = [v["hbm_bytes"] for k, v in prof["kernels"].items() if "vap::" in k]
    return sum(vals) if vals else None


def dropin_config1(calls=20):
    """BASELINE config 1 through the drop-in path: the GUI's own call sequence for one 8-waypoint path
    (gui/path.py:356-390 update_spline -> build_path; gui/path.py:301-354 generate_motion_profile_lists ->
    Constraints(...) + generate_motion_profile(spline_manager, constraints)) with the drop-in classes, wall time per call
    (host clock around the whole sequence, Python lists out, as the GUI receives them).  The reference's Python takes
    267 ms for generate_motion_profile on this path (one Xeon 2.1 GHz core)."""
    sys.path.insert(0, os.path.join(ROOT, "dropin"))
    try:
        from motion_profiling_v2 import motion_profile_generator as mpg
        from splines.spline_manager import QuinticHermiteSplineManager
        from vexautonomousplanner_amd.nodes import Node
        from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints
        from vexautonomousplanner_amd._device_path import DeviceRoute
        wp = make_waypoints(1, 8, 1)[0].astype(np.float64)        # config 1: seed 1, golden c1_w8
        nodes = [Node() for _ in wp]

        def leg(batch_kernels):
            DeviceRoute.use_batch_kernels = batch_kernels
            t_build, t_prof, rows = [], [], 0
            try:
                for i in range(calls + 2):
                    t0 = time.perf_counter()
                    sm = QuinticHermiteSplineManager()
                    ok = sm.build_path(wp, nodes, [])
                    t1 = time.perf_counter()
                    res = mpg.generate_motion_profile(sm, mpg.Constraints(*DEFAULT_CONSTRAINTS))
                    t2 = time.perf_counter()
                    assert ok and len(res) == 9
                    rows = len(res[0])
                    if i >= 2:                                    # two untimed calls: library load, first allocations
                        t_build.append(t1 - t0)
                        t_prof.append(t2 - t1)
            finally:
                DeviceRoute.use_batch_kernels = False
            per_call = np.add(t_build, t_prof) * 1e3
            return {"build_path_ms": float(np.mean(t_build)) * 1e3, "generate_motion_profile_ms": float(np.mean(t_prof)) * 1e3,
                    "ms_per_call": float(per_call.mean()), "min_ms": float(per_call.min()),
                    "median_ms": float(np.median(per_call)), "max_ms": float(per_call.max()), "time_rows": rows}

        out = {"workload": "c1: one 8-waypoint path, default constraints, dd = 0.005, dt = 0.01", "calls": calls,
               "layer": "one-lane vap_route_* kernels, the reference's statement order (the drop-in's default)"}
        out.update(leg(False))
        # the opt-in layer (DeviceRoute.use_batch_kernels = True): the batch kernels with B = 1
        out["batch_kernels"] = leg(True)
        out["reference_python_ms"] = {"build_path": 4.9, "generate_motion_profile": 267.0,
                                      "measured": "build container, 1 Xeon 2.1 GHz core (BASELINE.md section 2)"}
        return out
    finally:
        sys.path.remove(os.path.join(ROOT, "dropin"))


def launch_ranks(n):
    """Run this script under torch.distributed.run with n ranks on this node (rendezvous on 127.0.0.1, a free
    port); rank 0's JSON line goes to this process's stdout, the return value is the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default="f32", choices=("f32", "f64"))
    ap.add_argument("--recurrence", default="f64", choices=("f64", "f32"),
                    help="dtype f32 only: arithmetic of the velocity recurrence behind the fp32 rows (f64 = the default "
                         "mode of the library, the one that holds 1e-5 against the reference on every path)")
    ap.add_argument("--parity-paths", type=int, default=128,
                    help="paths of rank 0's batch checked against the CPU oracle after the timed region (0 = skip)")
    ap.add_argument("--paths-per-gpu", type=int, default=None)
    ap.add_argument("--spinup-ms", type=float, default=80.0,
                    help="untimed device spin-up before the warm-up steps: run the step for this long so the clocks settle")
    ap.add_argument("--in-flight", type=int, default=1,
                    help="after the timed region (N = 1): also time the steps with this many batches in flight on as many "
                         "HIP streams, reported as `pipelined` (1 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-mode", action="store_true",
                    help="skip the after-the-fact measurement of the other recurrence arithmetic (fp32 rows, N = 1)")
    ap.add_argument("--time-domain", action="store_true",
                    help="also time the batched time-domain resample (vap_time_profile) after the timed region (implies "
                         "--time-domain-residual: the step then also writes the fp32 residual row that call integrates)")
    ap.add_argument("--time-domain-residual", action="store_true",
                    help="VAP_OPT_TIME_DOMAIN_RESIDUAL on in the timed step (the library's default; 4 B/pt of extra writes that only "
                         "a following time-domain call reads).  The bench's hot path is the distance domain — five output rows "
                         "per sample-point — so its default is OFF, and `config.time_domain_residual` says which was timed")
    ap.add_argument("--no-dropin-c1", action="store_true",
                    help="skip the config-1 leg after the timed region (the GUI's call sequence for one 8-waypoint path "
                         "through the drop-in classes, wall time per call)")
    ap.add_argument("--tolerance-sweep", action="store_true",
                    help="BASELINE config 5: after the timed region run the same batch in the other precision modes and "
                         "report the per-path distribution of |fp32 - fp64| (relative, velocity) on the device")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="process-group backend; gloo + --share-device rehearses the multi-rank path on a 1-GPU box")
    ap.add_argument("--share-device", action="store_true", help="(rehearsal) every rank uses cuda:0")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: start the N ranks (one per GPU) as a CHILD torch.distributed.run and
        # hand its output and exit code through.  Nothing in this process has touched the GPU yet (no torch import).
        sys.exit(launch_ranks(args.gpus))

    import torch
    import torch.distributed as dist

    from vexautonomousplanner_amd.batch import BatchedTrajectoryGenerator
    from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints_block

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world        # under a launcher the launcher's world size is the truth
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # under a launcher (WORLD_SIZE in the environment) the process group is set up even for one rank, so that
    # `python -m torch.distributed.run --nproc-per-node 1 bench.py` takes RCCL through the same calls as N ranks do
    distributed = world > 1 or ("WORLD_SIZE" in os.environ and "MASTER_PORT" in os.environ)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    wl = dict(WORKLOADS[args.workload])
    if args.paths_per_gpu:
        wl["paths"] = args.paths_per_gpu
    B, W, S = wl["paths"], wl["W"], wl["S"]
    tdt = torch.float32 if args.dtype == "f32" else torch.float64

    from vexautonomousplanner_amd import dist as vdist
    # rank 0 owns the constraints; everyone else receives them over RCCL (setup, untimed)
    constraints = vdist.broadcast_constraints(DEFAULT_CONSTRAINTS if rank == 0 else None, dev)
    # rank r works on block r — paths [r*B, (r+1)*B) — of the seeded global batch, and generates that block ALONE
    # (synth.make_waypoints_block: block 0 is the 1-GPU batch; no rank builds another rank's paths, no waypoint scatter)
    wp = torch.tensor(make_waypoints_block(B, W, wl["seed"], rank, dtype=np.float32 if args.dtype == "f32" else np.float64),
                      dtype=tdt, device=dev)
    # what the communicator itself reports (RCCL saw N ranks: world size and an all-reduce of ones), before the timed region
    comm = vdist.comm_record(dev)

    residual = bool(args.time_domain_residual or args.time_domain)
    gen = BatchedTrajectoryGenerator(local_rank, args.dtype, recurrence=args.recurrence, time_domain_residual=residual)
    out = None

    def step():
        nonlocal out
        out = gen.profile(wp, constraints=constraints, samples=S, out=out)

    # Untimed: the clocks of an idle MI355X take tens of milliseconds of load to settle (measured: the same 20 timed steps
    # read 1.144 ms/step after 3 warm-up steps and 1.062 after 30) — so the device is kept busy with the step itself for
    # --spinup-ms before the W warm-up steps the contract asks for.  Nothing of this is inside the timed region.
    if args.spinup_ms > 0:
        t_spin = time.perf_counter()
        while (time.perf_counter() - t_spin) * 1e3 < args.spinup_ms:
            step()
            torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        step()

    def barrier():
        torch.cuda.synchronize(dev)
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        te = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())

    # per-kernel durations, HIP events on the library's stream (averaged over the same K steps)
    gen.ctx.set_timing(True)
    acc = {}
    for _ in range(args.steps):
        step()
        for k, v in gen.timing().items():
            acc[k] = acc.get(k, 0.0) + v / args.steps
    gen.ctx.set_timing(False)

    flags = int(out["flags"].abs().max().item())
    # after the timed region, outside `value`: the time-domain resample of the same batch (SURVEY §8(f)-1,
    # what generate_motion_profile does with the velocity rows), on request
    time_domain = None
    if args.time_domain and world == 1 and wl["S"] <= 100000:
        cap_rows = 4096
        tp = gen.time_profile(out, constraints, capacity_rows=cap_rows)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            tp = gen.time_profile(out, constraints, capacity_rows=cap_rows, out=tp)
        e1.record()
        torch.cuda.synchronize(dev)
        n_rows = int(tp["counts"][:, 0].sum().item())
        ms = e0.elapsed_time(e1) / 3
        time_domain = {"ms": ms, "rows": n_rows, "rows_per_s": n_rows / (ms * 1e-3), "dt_s": 0.01,
                       "truncated_paths": int((tp["counts"][:, 0] >= cap_rows).sum().item())}
    # after the timed region: all-gather per-path summaries (length, samples, traversal time) over RCCL —
    # 24 B/path, what a caller ranking candidate trajectories needs from the other GPUs
    summ = vdist.all_gather_rows(vdist.path_summaries(out["meta"], out["velocity"]), B * world)
    total_len = float(summ[:, 0].sum().item())
    best_time = float(summ[:, 2].min().item())

    # after the timed region, rank 0, N = 1: BASELINE config 1 through the drop-in path, for the record
    dropin_c1 = None
    if rank == 0 and world == 1 and not args.no_dropin_c1:
        dropin_c1 = dropin_config1()

    # parity of the mode just timed: the first paths of rank 0's batch against the oracle (CPU, after the timed region)
    parity = None
    if rank == 0 and args.parity_paths > 0:
        parity = parity_check(out, wp, S, constraints, min(args.parity_paths, B), args.dtype)

    # after the timed region, N = 1, fp32 rows only: the OTHER recurrence arithmetic on the same batch, for the record —
    # never `value`.  (default mode: fp64 recurrence, holds 1e-5 everywhere; "f32": round 1's, faster, leaves the bound)
    other_mode = None
    if world == 1 and args.dtype == "f32" and not args.no_other_mode:
        orec = "f32" if args.recurrence == "f64" else "f64"
        ogen = BatchedTrajectoryGenerator(local_rank, "f32", recurrence=orec, time_domain_residual=residual)
        oout = None
        for _ in range(2):
            oout = ogen.profile(wp, constraints=constraints, samples=S, out=oout)
        torch.cuda.synchronize(dev)
        k = max(3, min(args.steps, 10))
        t1 = time.perf_counter()
        for _ in range(k):
            oout = ogen.profile(wp, constraints=constraints, samples=S, out=oout)
        torch.cuda.synchronize(dev)
        oms = (time.perf_counter() - t1) / k * 1e3
        other_mode = {"recurrence": "f64 behind fp32 rows" if orec == "f64" else "f32", "ms_per_step": oms,
                      "value": B * S / (oms * 1e-3), "steps": k}
        if args.parity_paths > 0:
            other_mode["parity"] = parity_check(oout, wp, S, constraints, min(args.parity_paths, B), "f32")
        del ogen, oout

    # after the timed region, N = 1, for the record — never `value`: the same K steps with TWO batches in flight (two
    # contexts, two HIP streams, the steps taken in turn): the sampling kernel of one batch fills issue slots that the
    # chain-bound velocity kernel of the other leaves idle.  A step of `value` is one batch, start to end.
    pipelined = None
    if world == 1 and args.in_flight > 1 and B * S * 40 * args.in_flight < 64e9:
        n = args.in_flight
        gens = [gen] + [BatchedTrajectoryGenerator(local_rank, args.dtype, recurrence=args.recurrence, time_domain_residual=residual)
                        for _ in range(n - 1)]
        streams = [torch.cuda.Stream(dev) for _ in range(n)]
        outs = [out] + [None] * (n - 1)

        def turn(k):
            for i in range(k):
                q = i % n
                with torch.cuda.stream(streams[q]):
                    outs[q] = gens[q].profile(wp, constraints=constraints, samples=S, out=outs[q])
        torch.cuda.synchronize(dev)
        turn(2 * n + args.warmup)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        turn(args.steps)
        torch.cuda.synchronize(dev)
        pms = (time.perf_counter() - t1) / args.steps * 1e3
        pipelined = {"batches_in_flight": n, "ms_per_step": pms, "value": B * S / (pms * 1e-3), "steps": args.steps}
        out = outs[0]
        del gens, outs, streams

    # BASELINE config 5, "fp64 vs fp32 tolerance sweep": this rank's whole batch in every mode, compared on the device
    sweep = None
    if args.tolerance_sweep:
        sweep = tolerance_sweep(local_rank, wp, constraints, S)
        if distributed:
            # worst over ranks, counts summed: three small all-reduces after the timed region
            for mode in sweep["modes"].values():
                t = torch.tensor([mode["worst"]], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                mode["worst"] = float(t.item())
                c = torch.tensor([mode["paths_above_1e-5"], mode["paths"]], dtype=torch.float64, device=dev)
                dist.all_reduce(c)
                mode["paths_above_1e-5"], mode["paths"] = int(c[0].item()), int(c[1].item())

    if rank == 0:
        points = B * S * world
        esz = 4 if args.dtype == "f32" else 8
        bytes_per_point = 5 * esz + (12 * esz * (W - 1)) / S   # 20 B + 48*G/S (fp32)
        stage_bytes = {"sample": 4 * esz + (12 * esz * (W - 1)) / S, "velocity": esz,
                       "fit": (12 * esz * (W - 1)) / S, "lut": 0.0}
        dom = max(("fit", "lut", "sample", "velocity"), key=lambda k: acc.get(k, 0.0))
        dom_ms = acc[dom]
        achieved = stage_bytes[dom] * B * S / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        traffic, traffic_note = profiled_traffic(dom, args.workload, args.dtype, B, args.recurrence)
        line = {
            "metric": "trajectory sample-points/sec (batched paths)",
            "value": points / elapsed * args.steps,
            "unit": "sample-points/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "device_spinup_ms": args.spinup_ms,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": wl["name"], "paths_per_gpu": B, "waypoints": W, "samples": S,
                       "grid": "dd_p = L_p / (S - 1.5), the reference's running sum from 0 plus its appended end sample",
                       "recurrence": ("f64 behind fp32 rows" if args.recurrence == "f64" else "f32") if args.dtype == "f32" else "f64",
                       "time_domain_residual": residual,
                       "global_paths": B * world, "parallelism": f"paths sharded x{world}, no data-path collective",
                       "flags_or": flags, "sum_path_length_ft": total_len,
                       "fastest_traversal_s": best_time},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note,
                         "kernel_ms": dom_ms, "algorithmic_bytes_per_point": stage_bytes[dom]},
            "comm": comm,
            "kernel_source_sha": kernel_source_sha(),
            "pipeline": {"bytes_per_point": bytes_per_point,
                         "traffic": profiled_pipeline_traffic(args.workload, args.dtype, B, args.recurrence),
                         "achieved_GBs": bytes_per_point * B * S / (acc["total"] * 1e-3) / 1e9,
                         "frac_of_hbm_peak": bytes_per_point * B * S / (acc["total"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "stage_ms": {k: round(v, 4) for k, v in acc.items()}},
        }
        # the counted bytes as a rate: how hard the memory system is actually being driven (the algorithmic fraction above
        # says how much of that is bytes the caller asked for)
        if traffic:
            line["roofline"]["traffic_GBs"] = traffic / (dom_ms * 1e-3) / 1e9
            line["roofline"]["traffic_frac_of_peak"] = line["roofline"]["traffic_GBs"] / HBM_PEAK_GBS
        if line["pipeline"]["traffic"]:
            line["pipeline"]["traffic_GBs"] = line["pipeline"]["traffic"] / (acc["total"] * 1e-3) / 1e9
            line["pipeline"]["traffic_frac_of_peak"] = line["pipeline"]["traffic_GBs"] / HBM_PEAK_GBS
        if parity is not None:
            line["parity"] = parity
        if other_mode is not None:
            line["other_mode"] = other_mode
        if pipelined is not None:
            line["pipelined"] = pipelined
        if sweep is not None:
            line["tolerance_sweep"] = sweep
        if time_domain is not None:
            line["time_domain"] = time_domain
        if dropin_c1 is not None:
            line["dropin_c1"] = dropin_c1
            line["dropin_c1_ms"] = dropin_c1["median_ms"]   # (mean, min and max are in the object: a full Python GC pass — ~40 ms with torch imported — lands in one call of a few dozen)
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(wl)
            # context, not measured here: the reference's own Python cannot travel to this box
            line["cpu_baseline"]["reference_python"] = {"value": 5.3e3, "unit": "sample-points/s", "cores": 1,
                                                        "measured": "build container, 1 Xeon 2.1 GHz core, config 1 (BASELINE.md)"}
        print(json.dumps(line))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
