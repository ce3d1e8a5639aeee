"""GPU parity at scale: the rate of paths outside the bound on random sweeps, and every BASELINE config at its
per-GPU size.

The curated cases of test_gpu_parity.py say "these paths hold the bound"; the tests here say how many random ones
do (north_star: <= 1e-5 relative against the reference, point for point):

  * config-3 shape (32 waypoints x 10 000 samples), 2048 random paths against the oracle, per mode:
      "f32"    fp32 rows, fp64 recurrence (the default, the mode bench.py quotes `value` on): EVERY path inside
      "f32r32" all-fp32 recurrence: the rate is recorded and bounded (1.4 % of paths when this was written:
               the reference's recurrence amplifies rounding errors in curves tighter than 2/track_width)
      "f64"    1e-7 (rounding-level differences of the inputs, amplified by the same mechanism; median 1e-12)
  * a seeded slice of tools/fuzz_parity.py (random shapes, robots, grids, start / end velocities)
  * config 3, config 4's per-GPU share, config 5's per-GPU share: flags, bounds on every velocity, traversal times,
    and a random 64-path subset against the oracle.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def make_gen(dtype, **kw):
    from vexautonomousplanner_amd.batch import BatchedTrajectoryGenerator
    if dtype == "f32r32":
        return BatchedTrajectoryGenerator(0, "f32", recurrence="f32", **kw)
    return BatchedTrajectoryGenerator(0, dtype, **kw)


def per_path_errors(got, ref):
    """Worst sample of every path, the measures of test_gpu_parity.py."""
    e = {"velocity": np.max(np.abs(got["velocity"] - ref["velocity"]) / np.abs(ref["velocity"]), axis=1),
         "curvature": np.max(np.abs(got["curvature"] - ref["curvature"]) / np.maximum(np.abs(ref["curvature"]), 1e-2), axis=1),
         "heading": np.max(np.abs(got["heading"] - ref["heading"]), axis=1) / np.pi,
         "x": np.max(np.abs(got["x"] - ref["x"]) / np.maximum(np.abs(ref["x"]), 1.0), axis=1),
         "y": np.max(np.abs(got["y"] - ref["y"]) / np.maximum(np.abs(ref["y"]), 1.0), axis=1)}
    return e


def run(torch, gen, wp64, **kw):
    wp = torch.tensor(wp64, dtype=gen.tdtype, device=gen.device)
    r = gen.profile(wp, **kw)
    torch.cuda.synchronize()
    return {k: (v.cpu().numpy().astype(np.float64) if k != "flags" else v.cpu().numpy()) for k, v in r.items()}


SWEEP_PATHS = 2048


@pytest.fixture(scope="module")
def config3_sweep():
    """2048 random config-3-shaped paths (the generator of SURVEY 8(d), a seed no fixture uses) and the oracle's rows."""
    from oracle import oracle
    from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints
    wp = make_waypoints(SWEEP_PATHS, 32, 12345).astype(np.float32).astype(np.float64)   # what an fp32 caller hands over
    ref = oracle.profile_batch(wp, 10000, DEFAULT_CONSTRAINTS, n_threads=16)
    return wp, ref


@pytest.mark.parametrize("dtype", ["f32", "f32r32", "f64"])
def test_config3_sweep_rate_of_paths_outside_the_bound(torch_mod, config3_sweep, dtype):
    wp, ref = config3_sweep
    gen = make_gen(dtype)
    got = run(torch_mod, gen, wp, samples=10000)
    assert not got["flags"].any()
    e = per_path_errors(got, ref)
    worst = {k: float(v.max()) for k, v in e.items()}
    bound = 1e-7 if dtype == "f64" else 1e-5
    frac = float(np.mean(e["velocity"] > bound))
    print(f"config-3 sweep {dtype}: {SWEEP_PATHS} paths, velocity worst {worst['velocity']:.2e} median {np.median(e['velocity']):.2e} "
          f"frac above {bound:g}: {frac:.4f}; curvature {worst['curvature']:.2e} heading {worst['heading']:.2e} x {worst['x']:.2e} y {worst['y']:.2e}")
    geo = 1e-9 if dtype == "f64" else 1e-5
    assert worst["curvature"] <= geo and worst["heading"] <= geo and worst["x"] <= geo and worst["y"] <= geo
    if dtype == "f32r32":
        # the all-fp32 recurrence: recorded, and bounded so that it cannot get worse unnoticed — and from below, so
        # that the sweep is known to still bite (a sweep on which even this mode passes everywhere would prove nothing
        # about the default mode)
        assert 0.005 <= frac <= 0.03 and worst["velocity"] <= 5e-4
    else:
        assert frac == 0.0, f"{int(frac * SWEEP_PATHS)} paths above {bound:g} (worst {worst['velocity']:.2e})"
        if dtype == "f32":
            assert worst["velocity"] <= 2e-6    # far inside: output rounding plus the amplified input differences


FUZZ_CASES = 160


@pytest.mark.parametrize("dtype,tol", [("f32", 1e-5), ("f64", 1e-7)])
def test_fuzz_slice_random_shapes_robots_grids(torch_mod, dtype, tol):
    """A seeded slice of tools/fuzz_parity.py: random W, S, batch, robot, start / end velocity, fixed-S grid or the
    reference's own dd grid (ragged rows) — every case inside the bound, sample counts equal."""
    from oracle import oracle
    from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints
    torch = torch_mod
    rng = np.random.default_rng(20261004)
    gen = make_gen(dtype)
    worst = {}
    fails = []
    for case in range(FUZZ_CASES):
        W = int(rng.choice([2, 3, 4, 5, 8, 13, 32, 57, 113]))
        S = int(rng.choice([2, 3, 7, 64, 255, 256, 257, 1000, 1024, 1025, 4096, 4097, 10000, 20481]))
        B = int(rng.integers(1, 9)) if S <= 10000 else 1
        seed = int(rng.integers(0, 1 << 30))
        cons = list(DEFAULT_CONSTRAINTS)
        if rng.random() < 0.5:
            cons[0] = float(rng.uniform(1.0, 8.0))
            cons[1] = float(rng.uniform(2.0, 16.0))
            cons[2] = float(rng.uniform(2.0, 16.0))
            cons[5] = float(rng.uniform(0.5, 2.0))
        wp = make_waypoints(B, W, seed).astype(np.float32).astype(np.float64)
        sv, ev = 0.01, 0.01
        if rng.random() < 0.3:
            sv, ev = float(rng.uniform(0.01, 2.0)), float(rng.uniform(0.01, 2.0))
        use_dd = S <= 4097 and rng.random() < 0.3
        if use_dd:
            dd = float(rng.uniform(0.002, 0.02))
            per = []
            for b in range(B):
                op = oracle.OraclePath(wp[b])
                op.rebuild_tables()
                per.append(op.forward_backward(cons, dd=dd, start_vel=sv, end_vel=ev))
            cap = max(len(p["velocity"]) for p in per) + 3
            ref = {k: np.zeros((B, cap)) for k in ("x", "y", "heading", "curvature", "velocity")}
            for b, pth in enumerate(per):
                for k in ref:
                    ref[k][b, :len(pth[k])] = pth[k]
            pad = ref["velocity"] == 0
            ref["velocity"][pad] = 1.0
            got = run(torch, gen, wp, constraints=cons, dd=dd, capacity=cap, start_vel=sv, end_vel=ev)
            n_ref = np.array([len(p["velocity"]) for p in per])
            assert np.array_equal(got["meta"][:, 3].astype(int), n_ref), (case, W, dd, seed)
            assert np.all(got["velocity"][pad] == 0)
            got["velocity"][pad] = 1.0
        else:
            ref = oracle.profile_batch(wp, S, cons, start_vel=sv, end_vel=ev, n_threads=8)
            got = run(torch, gen, wp, constraints=cons, samples=S, start_vel=sv, end_vel=ev)
        assert not got["flags"].any(), (case, got["flags"])
        e = {k: float(v.max()) for k, v in per_path_errors(got, ref).items()}
        for k, v in e.items():
            worst[k] = max(worst.get(k, 0.0), v)
        geo = 1e-9 if dtype == "f64" else 1e-5
        if not (e["velocity"] <= tol and all(e[k] <= geo for k in ("curvature", "heading", "x", "y"))):
            fails.append((case, B, W, S, "dd" if use_dd else "fixed", seed, cons, e))
    print(f"fuzz slice {dtype}: {FUZZ_CASES} cases, worst " + " ".join(f"{k} {v:.2e}" for k, v in worst.items()))
    assert not fails, fails


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_large_w_worst_case_vs_reference_fixture(torch_mod, dtype):
    """big_w2048_p2: the worst case of the large-W randomized sweep (tools/fuzz_parity.py big, case 115, path 2: 2048
    waypoints, 415 173 samples on the reference's own grid) as the REAL reference computed it — the whole velocity row.
    On such paths the 1000-entry arc-length table is sparse (2 segments per entry), curvature reaches the amplifying
    regime and last-bit differences are multiplied up to 1e9 (the oracle itself is 7.9e-8 from the reference here,
    tests/test_oracle_golden.py), so BOTH row types are held to north_star's 1e-5; geometry to the usual bounds."""
    import golden_util as gu
    torch = torch_mod
    g = gu.load("big_w2048_p2")
    N = int(g["n_samples"])
    gen = make_gen(dtype)
    got = run(torch, gen, g["waypoints"][None], constraints=g["constraints"], dd=float(g["dd"]), capacity=N + 2)
    assert not got["flags"].any() and int(got["meta"][0, 3]) == N
    ref_v = g["velocity_full"]
    ev = np.abs(got["velocity"][0, :N] - ref_v) / ref_v
    gi = g["grid_idx"]
    geo = 1e-9 if dtype == "f64" else 1e-5
    ek = np.max(np.abs(got["curvature"][0][gi] - g["grid_curvature"]) / np.maximum(np.abs(g["grid_curvature"]), 1e-2))
    eh = np.max(np.abs(got["heading"][0][gi] - g["grid_heading"])) / np.pi
    ex = np.max(np.abs(got["x"][0][gi] - g["grid_x"]) / np.maximum(np.abs(g["grid_x"]), 1.0))
    print(f"big_w2048_p2/{dtype} vs the reference: velocity worst {ev.max():.2e} at sample {int(ev.argmax())} "
          f"({int((ev > 1e-7).sum())} above 1e-7), curvature {ek:.2e} heading {eh:.2e} x {ex:.2e}")
    assert ev.max() <= 1e-5 and ek <= geo and eh <= geo and ex <= geo


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("W", [113, 513, 2048])
def test_large_w_slice_vs_oracle(torch_mod, dtype, W):
    """The gated slice of `tools/fuzz_parity.py big`: paths of 113 / 513 / 2048 waypoints (past the LDS-resident
    coefficient limit, up to the maximum), fixed-S grids and the reference's own dd grid, against the oracle.
    Velocity bound 1e-5 for both row types (see test_large_w_worst_case_vs_reference_fixture: the oracle is itself
    ~1e-7 from the reference on such paths); geometry 1e-5 / 1e-9."""
    from oracle import oracle
    from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints
    torch = torch_mod
    rng = np.random.default_rng(9000 + W)
    gen = make_gen(dtype)
    worst = {}
    for case in range(4):
        B = int(rng.integers(1, 4))
        seed = int(rng.integers(0, 1 << 30))
        wp = make_waypoints(B, W, seed).astype(np.float32).astype(np.float64)
        cons = list(DEFAULT_CONSTRAINTS)
        if case % 2:
            cons[0], cons[1], cons[5] = float(rng.uniform(1.0, 8.0)), float(rng.uniform(2.0, 16.0)), float(rng.uniform(0.5, 2.0))
        if case < 2:        # the reference's own grid, ragged rows
            dd = float(rng.uniform(0.003, 0.02))
            per = []
            for b in range(B):
                op = oracle.OraclePath(wp[b])
                op.rebuild_tables()
                per.append(op.forward_backward(cons, dd=dd))
            cap = max(len(p["velocity"]) for p in per) + 3
            ref = {k: np.zeros((B, cap)) for k in ("x", "y", "heading", "curvature", "velocity")}
            for b, pth in enumerate(per):
                for k in ref:
                    ref[k][b, :len(pth[k])] = pth[k]
            pad = ref["velocity"] == 0
            ref["velocity"][pad] = 1.0
            got = run(torch, gen, wp, constraints=cons, dd=dd, capacity=cap)
            assert np.array_equal(got["meta"][:, 3].astype(int), np.array([len(p["velocity"]) for p in per]))
            assert np.all(got["velocity"][pad] == 0)
            got["velocity"][pad] = 1.0
        else:
            S = int(rng.choice([4097, 10000, 30000]))
            ref = oracle.profile_batch(wp, S, cons, n_threads=8)
            got = run(torch, gen, wp, constraints=cons, samples=S)
        assert not got["flags"].any()
        e = {k: float(v.max()) for k, v in per_path_errors(got, ref).items()}
        for k, v in e.items():
            worst[k] = max(worst.get(k, 0.0), v)
    print(f"large-W slice W={W}/{dtype}: worst " + " ".join(f"{k} {v:.2e}" for k, v in worst.items()))
    geo = 1e-9 if dtype == "f64" else 1e-5
    assert worst["velocity"] <= 1e-5 and all(worst[k] <= geo for k in ("curvature", "heading", "x", "y")), worst


CONFIGS = {
    "c3": dict(paths=4096, W=32, S=10000, seed=3),          # BASELINE config 3
    "c4_share": dict(paths=8192, W=32, S=10000, seed=4),    # config 4: 65 536 paths over 8 GPUs
    "c5_share": dict(paths=131072, W=8, S=1024, seed=5),    # config 5: 1 M paths over 8 GPUs
}


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("cfg", sorted(CONFIGS))
def test_config_at_per_gpu_size(torch_mod, cfg, dtype):
    """Every batched BASELINE config at the size one GPU sees: no flags, every velocity finite and inside
    [0.01, max_vel], positive traversal times, and 64 random paths of the batch against the oracle."""
    from oracle import oracle
    from vexautonomousplanner_amd import dist as vdist
    from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints
    torch = torch_mod
    c = CONFIGS[cfg]
    B, W, S = c["paths"], c["W"], c["S"]
    gen = make_gen(dtype)
    wp64 = make_waypoints(B, W, c["seed"]).astype(np.float32).astype(np.float64)
    wp = torch.tensor(wp64, dtype=gen.tdtype, device=gen.device)
    r = gen.profile(wp, DEFAULT_CONSTRAINTS, samples=S)
    torch.cuda.synchronize()
    assert int(r["flags"].abs().max().item()) == 0
    v = r["velocity"]
    assert bool(torch.isfinite(v).all().item())
    vmax = DEFAULT_CONSTRAINTS[0]
    assert float(v.min().item()) >= 0.01 * (1 - 1e-5) and float(v.max().item()) <= vmax * (1 + 1e-6)
    for k in ("x", "y", "heading", "curvature"):
        assert bool(torch.isfinite(r[k]).all().item()), k
    summ = vdist.path_summaries(r["meta"], v)
    assert float(summ[:, 2].min().item()) > 0.0 and bool(torch.isfinite(summ).all().item())
    assert bool((r["meta"][:, 3] == S).all().item())
    idx = np.sort(np.random.default_rng(c["seed"]).choice(B, size=64, replace=False))
    ref = oracle.profile_batch(wp64[idx], S, DEFAULT_CONSTRAINTS, n_threads=16)
    tidx = torch.tensor(idx, device=gen.device)
    got = {k: r[k][tidx].cpu().numpy().astype(np.float64) for k in ("x", "y", "heading", "curvature", "velocity")}
    e = {k: float(val.max()) for k, val in per_path_errors(got, ref).items()}
    print(f"{cfg}/{dtype}: {B} x {W} x {S}; 64-path subset worst " + " ".join(f"{k} {val:.2e}" for k, val in e.items()))
    tol_v, tol_g = (1e-7, 1e-9) if dtype == "f64" else (1e-5, 1e-5)
    assert e["velocity"] <= tol_v and all(e[k] <= tol_g for k in ("curvature", "heading", "x", "y")), e


@pytest.mark.parametrize("B,W", [(32768 + 37, 8), (32768 + 37, 3), (8192 + 5, 32), (8192 + 3, 64), (8192 + 1, 2)])
def test_many_paths_table_kernel_is_bit_identical(torch_mod, B, W):
    """Large batches build their arc-length tables several paths per workgroup (k_lut_many: the sequential sums of the
    group's paths in the lanes of one wavefront — 64 paths per workgroup from 32 768 paths of <= 9 waypoints on, 8 paths
    from 8192 paths of <= 64 waypoints on); smaller batches one path per workgroup (k_lut).  Same expressions, same
    order: every row of the big batch equals the row of the same path run in a small batch."""
    torch = torch_mod
    from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints
    S = 64
    gen = make_gen("f64")
    wp = torch.tensor(make_waypoints(B, W, 77), dtype=gen.tdtype, device=gen.device)
    big = {k: v.clone() for k, v in gen.profile(wp, DEFAULT_CONSTRAINTS, samples=S).items()}
    idx = torch.cat([torch.arange(0, 300), torch.arange(B - 300, B), torch.arange(B // 2, B // 2 + 300)]).to(gen.device)
    small = gen.profile(wp[idx].contiguous(), DEFAULT_CONSTRAINTS, samples=S)
    torch.cuda.synchronize()
    assert int(big["flags"].abs().max().item()) == 0
    for k in ("meta", "x", "y", "heading", "curvature", "velocity"):
        assert torch.equal(big[k][idx], small[k]), k


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_full_size_batch_is_deterministic_and_order_independent(torch_mod, dtype):
    """Size-independent properties at config 3's full size (4096 x 32 x 10 000): the same batch twice gives the same bits
    (the relaxation's fixed point does not depend on scheduling), and a permuted batch gives the permuted rows — a
    path's result does not depend on its neighbours or on its position in the batch."""
    torch = torch_mod
    from vexautonomousplanner_amd.batch import BatchedTrajectoryGenerator
    from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints
    gen = BatchedTrajectoryGenerator(0, dtype)
    B, W, S = 4096, 32, 10000
    wp = torch.tensor(make_waypoints(B, W, 3), dtype=gen.tdtype, device=gen.device)
    first = {k: v.clone() for k, v in gen.profile(wp, DEFAULT_CONSTRAINTS, samples=S, want=("curvature", "velocity")).items()
             if k in ("curvature", "velocity", "meta", "flags")}
    again = gen.profile(wp, DEFAULT_CONSTRAINTS, samples=S, want=("curvature", "velocity"))
    torch.cuda.synchronize()
    for k in first:
        assert torch.equal(first[k], again[k]), k
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(5)).to(gen.device)
    shuffled = gen.profile(wp[perm].contiguous(), DEFAULT_CONSTRAINTS, samples=S, want=("curvature", "velocity"))
    torch.cuda.synchronize()
    for k in first:
        assert torch.equal(first[k][perm], shuffled[k]), k
    assert int(first["flags"].abs().max().item()) == 0


def test_tolerance_sweep(torch_mod):
    """BASELINE config 5, "fp64 vs fp32 tolerance sweep" (bench.py --tolerance-sweep), as a gate, at the per-GPU share of
    the full config: 131 072 config-5-shaped paths (8 waypoints x 1024 samples) in the two fp32-row modes against this
    library's fp64 run of the same batch.  Default mode: no path above 1e-5, the worst far inside (the size matters: with
    the heading differences rounded to fp32 — tried in round 3 — 16 384 paths showed a worst of 4e-7 and this batch one of
    9.6e-6: the recurrence's amplification has a long tail).  All-fp32 recurrence: some paths are outside — the sweep
    keeps biting."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints
    torch = torch_mod
    B, W, S = 131072, 8, 1024
    wp = torch.tensor(make_waypoints(B, W, 5, dtype=np.float32), dtype=torch.float32, device="cuda:0")
    sweep = bench.tolerance_sweep(0, wp, list(DEFAULT_CONSTRAINTS), S)
    default = sweep["modes"]["f32 rows, f64 recurrence (default)"]
    f32rec = sweep["modes"]["f32 rows, f32 recurrence"]
    print(sweep)
    assert default["paths"] == B and f32rec["paths"] == B
    assert default["paths_above_1e-5"] == 0 and default["worst"] <= 5e-7
    assert f32rec["paths_above_1e-5"] > 0 and f32rec["worst"] > 1e-5
