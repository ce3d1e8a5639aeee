"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle and the golden vectors.

Tolerances (north_star: <= 1e-5 relative in fp32 vs the reference profile):
  velocity   : |dv| <= 1e-5 * |v_ref|                (pure relative; v >= 0.01 everywhere)
  curvature  : |dk| <= 1e-5 * max(|k_ref|, 1e-2)     (relative, floored at 0.01 rad/ft)
  heading    : |dh| <= 1e-5 * pi                     (absolute: headings pass through 0)
  x, y       : |dx| <= 1e-5 * max(|x_ref|, 1)        (feet)
fp64 runs are held to 1e-9 on the same measures.
"""
import numpy as np
import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def make_gen(dtype, **kw):
    """"f32" = fp32 rows with the fp64 recurrence behind them (the default mode), "f32r32" = the all-fp32
    recurrence (VAP_RECURRENCE_F32), "f64" = fp64 throughout."""
    from vexautonomousplanner_amd.batch import BatchedTrajectoryGenerator
    if dtype == "f32r32":
        return BatchedTrajectoryGenerator(0, "f32", recurrence="f32", **kw)
    return BatchedTrajectoryGenerator(0, dtype, **kw)


@pytest.fixture(scope="module")
def gens(torch_mod):
    return {k: make_gen(k) for k in ("f32", "f32r32", "f64")}


DTYPES_TOL = [("f32", 1e-5), ("f32r32", 1e-5), ("f64", 1e-9)]
DTYPES = ["f32", "f32r32", "f64"]


def check_fields(got, ref, tol, what=""):
    v, rv = got["velocity"], ref["velocity"]
    assert np.all(np.isfinite(v)), what
    e_v = np.max(np.abs(v - rv) / np.abs(rv))
    e_k = np.max(np.abs(got["curvature"] - ref["curvature"]) / np.maximum(np.abs(ref["curvature"]), 1e-2))
    e_h = np.max(np.abs(got["heading"] - ref["heading"])) / np.pi
    e_x = np.max(np.abs(got["x"] - ref["x"]) / np.maximum(np.abs(ref["x"]), 1.0))
    e_y = np.max(np.abs(got["y"] - ref["y"]) / np.maximum(np.abs(ref["y"]), 1.0))
    msg = f"{what}: v {e_v:.2e} k {e_k:.2e} h {e_h:.2e} x {e_x:.2e} y {e_y:.2e}"
    print(msg)
    assert e_v <= tol and e_k <= tol and e_h <= tol and e_x <= tol and e_y <= tol, msg
    return e_v, e_k, e_h, e_x


def run_gpu(torch, gen, wp64, **kw):
    wp = torch.tensor(wp64, dtype=gen.tdtype, device=gen.device)
    r = gen.profile(wp, **kw)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy().astype(np.float64) if k != "flags" else v.cpu().numpy() for k, v in r.items()}


# cons_*: robots with max_dec != max_acc / other limits (the reference overwrites max_dec with max_acc before
# the first forward step, MPG:110, 194-196)
GOLDEN_FIXED = [n for n in gu.names(("c3_", "c5_", "c2_w256_S20000", "cons_")) if n.startswith("cons_") is False or n.endswith("_S1024")]
GOLDEN_DD = [n for n in gu.names(("plain_", "c1_", "cons_")) if n.startswith("cons_") is False or n.endswith("_w8")]


@pytest.mark.parametrize("dtype,tol", DTYPES_TOL)
@pytest.mark.parametrize("name", GOLDEN_FIXED)
def test_fixed_grid_vs_reference_golden(torch_mod, gens, name, dtype, tol):
    g = gu.load(name)
    S = int(g["samples"])
    r = run_gpu(torch_mod, gens[dtype], g["waypoints"][None], samples=S, constraints=g["constraints"])
    assert r["flags"][0] == 0
    assert abs(r["meta"][0, 1] - float(g["total_length"])) <= 1e-14 * float(g["total_length"])
    assert r["meta"][0, 2] == pytest.approx(float(g["dd"]), rel=1e-14)
    gi = g["grid_idx"]
    got = {k: r[k][0][gi] for k in ("x", "y", "heading", "curvature", "velocity")}
    ref = {"x": g["grid_x"], "y": g["grid_y"], "heading": g["grid_heading"],
           "curvature": g["grid_curvature"], "velocity": g["grid_velocity"]}
    check_fields(got, ref, tol, f"{name}/{dtype}")


@pytest.mark.parametrize("dtype,tol", DTYPES_TOL)
@pytest.mark.parametrize("name", GOLDEN_DD)
def test_reference_grid_vs_reference_golden(torch_mod, gens, name, dtype, tol):
    g = gu.load(name)
    N = int(g["n_samples"])
    r = run_gpu(torch_mod, gens[dtype], g["waypoints"][None], dd=float(g["dd"]), capacity=N + 7,
                constraints=g["constraints"])
    assert r["flags"][0] == 0
    assert int(r["meta"][0, 3]) == N
    gi = g["grid_idx"]
    got = {k: r[k][0][:N][gi] for k in ("x", "y", "heading", "curvature", "velocity")}
    ref = {"x": g["grid_x"], "y": g["grid_y"], "heading": g["grid_heading"],
           "curvature": g["grid_curvature"], "velocity": g["grid_velocity"]}
    check_fields(got, ref, tol, f"{name}/{dtype}")
    # the rows are zero-filled past n_samples
    assert np.all(r["velocity"][0][N:] == 0)


@pytest.mark.parametrize("dtype,tol", [("f32", 1e-5), ("f64", 1e-9)])
def test_running_sum_grid_vs_reference_golden(torch_mod, gens, dtype, tol):
    """MPG:112-122: the grid is the running sum current_dist += dd.  On this 2000-waypoint, 654 050-sample
    path of the real reference the sum has drifted far enough from k*dd to select another table entry
    (fixture field runsum_flip_idx); sample count, table entries (curvature, heading) and points must
    follow the reference.  Velocity: fp64 rows at 1e-9, and the default mode's fp32 rows (fp64 recurrence behind
    them) at north_star's 1e-5 — curvature reaches |kappa| = 14 /ft here, far inside the amplifying regime
    (DESIGN.md section 3)."""
    g = gu.load("runsum_w2000")
    N = int(g["n_samples"])
    flips = g["runsum_flip_idx"]
    assert len(flips) >= 1
    r = run_gpu(torch_mod, gens[dtype], g["waypoints"][None], dd=float(g["dd"]), capacity=N + 2,
                constraints=g["constraints"])
    assert r["flags"][0] == 0
    assert int(r["meta"][0, 3]) == N
    gi = g["grid_idx"]
    assert np.all(np.isin(flips, gi))
    for k, ref in (("x", g["grid_x"]), ("y", g["grid_y"]), ("curvature", g["grid_curvature"])):
        got = r[k][0][gi]
        err = np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1.0 if k != "curvature" else 1e-2))
        assert err <= tol, (k, err)
    assert np.max(np.abs(r["heading"][0][gi] - g["grid_heading"])) <= tol * np.pi
    ev = np.max(np.abs(r["velocity"][0][gi] - g["grid_velocity"]) / g["grid_velocity"])
    print(f"runsum_w2000 {dtype} velocity max rel err {ev:.2e}")
    assert ev <= tol


def _initial_velocities(t, W, max_vel, end_vel, node_max_velocity, node_stop):
    """MPG:100-176 restricted to node max_velocity / stop: the `velocities` list forward_backward_pass starts
    from, given the parameter t of every sample (the loop samples, then the appended end sample)."""
    mv = node_max_velocity[0] if node_max_velocity[0] > 0 else max_vel
    out = []
    node, prev_t, t_end = 0, 0.0, float(W - 1)
    for tk in t[:-1]:
        out.append(mv)
        if (prev_t % 1) > (tk % 1) and tk < t_end:
            node += 1
            if node_stop[node]:
                out[-1] = 0.01
            mv = node_max_velocity[node] if node_max_velocity[node] > 0 else max_vel
        prev_t = tk
    out.append(end_vel)
    return np.array(out)


def _staged_velocity(torch, dtype, wp64, cons, vcap64, dd, cap, kernel):
    """vap_fit -> vap_build_lut -> vap_sample -> vap_velocity_pass(d_vcap) on a batch; returns velocity rows, n."""
    import ctypes as C
    from vexautonomousplanner_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda:0")
    td = torch.float64 if dtype == "f64" else torch.float32
    vd = _lib.VAP_F64 if dtype == "f64" else _lib.VAP_F32
    B, W = wp64.shape[:2]
    ctx = _lib.Context(0)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    ctx.set_option(_lib.OPT_VELOCITY_KERNEL, kernel)
    ctx.set_option(_lib.OPT_F32_RECURRENCE, _lib.RECURRENCE_F32 if dtype == "f32r32" else _lib.RECURRENCE_F64)
    # "f32": the velocity pass takes the fp64 rows vap_sample left on the context (d_dtheta = NULL)
    ctx_rows = dtype == "f32"
    c = _lib.make_constraints(cons)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    wp = torch.tensor(wp64, device=dev, dtype=td)
    seg = torch.empty((B, W - 1, 6, 2), dtype=torch.float64, device=dev)
    seglen = torch.empty((B, W - 1), dtype=torch.float64, device=dev)
    meta = torch.zeros((B, 4), dtype=torch.float64, device=dev)
    flags = torch.zeros((B,), dtype=torch.int32, device=dev)
    lut = torch.empty((B, _lib.LUT_SAMPLES), dtype=torch.float64, device=dev)
    o = {k: torch.empty((B, cap), dtype=td, device=dev) for k in ("x", "y", "heading", "curvature", "dtheta", "velocity")}
    # limit rows: the type of the recurrence they enter (vap_limit_rows_dtype: fp64 in the default "f32" mode)
    ltd = torch.float64 if L.vap_limit_rows_dtype(ctx.handle, vd) == _lib.VAP_F64 else torch.float32
    vc = torch.tensor(vcap64, device=dev, dtype=ltd) if vcap64 is not None else None
    _lib.check(L.vap_fit(ctx.handle, vd, B, W, p(wp), None, None, p(seg), p(seglen), p(meta), p(flags)), "vap_fit")
    _lib.check(L.vap_build_lut(ctx.handle, B, W, p(seg), p(lut), p(meta), p(flags)), "vap_build_lut")
    _lib.check(L.vap_sample(ctx.handle, vd, B, W, cap, dd, p(seg), p(lut), p(meta), p(o["x"]), p(o["y"]), p(o["heading"]),
                            p(o["curvature"]), p(o["dtheta"]), p(flags)), "vap_sample")
    _lib.check(L.vap_velocity_pass(ctx.handle, vd, B, cap, C.byref(c), 0.01, 0.01, p(meta), p(o["curvature"]),
                                   None if ctx_rows else p(o["dtheta"]), p(vc), p(o["velocity"]), p(flags)), "vap_velocity_pass")
    torch.cuda.synchronize()
    assert not flags.any().item()
    return o["velocity"].cpu().numpy().astype(np.float64), meta[:, 3].cpu().numpy().astype(int)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,W,S", [(6, 8, 1000), (3, 32, 10000), (5, 5, 257), (4, 2, 64)])
def test_initial_velocities_relaxation_equals_sequential_sweep(torch_mod, dtype, B, W, S):
    """d_vcap (per-sample initial velocities, MPG:121,127,153,172) in the register-resident relaxation kernel
    against the one-lane sequential sweep: bit for bit, as without them."""
    from vexautonomousplanner_amd import _lib
    from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints
    rng = np.random.default_rng(B * 1000 + W)
    wp = make_waypoints(B, W, 77).astype(np.float64)
    # piecewise-constant limits with a few "stops", like node / action-point limits mapped onto samples
    vcap = np.empty((B, S))
    for b in range(B):
        edges = np.sort(rng.integers(1, S - 1, size=5))
        vals = rng.uniform(0.8, 4.0, size=6)
        vcap[b] = vals[np.searchsorted(edges, np.arange(S), side="right")]
        vcap[b, rng.integers(1, S - 1, size=3)] = 0.01
    v_seq, _ = _staged_velocity(torch_mod, dtype, wp, DEFAULT_CONSTRAINTS, vcap, 0.0, S, _lib.VELOCITY_SEQ_FAST)
    v_rel, _ = _staged_velocity(torch_mod, dtype, wp, DEFAULT_CONSTRAINTS, vcap, 0.0, S, _lib.VELOCITY_RELAX)
    v_plain, _ = _staged_velocity(torch_mod, dtype, wp, DEFAULT_CONSTRAINTS, None, 0.0, S, _lib.VELOCITY_RELAX)
    assert np.array_equal(v_seq, v_rel)
    assert np.all(v_rel[:, 1:-1] <= vcap[:, 1:-1] * (1 + 1e-6))
    assert np.any(v_rel < v_plain * 0.99)       # the limits bind somewhere


@pytest.mark.parametrize("dtype,tol", DTYPES_TOL)
def test_initial_velocities_stop_node_vs_reference_golden(torch_mod, dtype, tol):
    """feat_stop (the real reference, a stop at node 3): the batched path with the reference's initial velocity
    list as d_vcap reproduces forward_backward_pass."""
    from vexautonomousplanner_amd import _lib
    g = gu.load("feat_stop")
    N = int(g["n_samples"])
    assert np.array_equal(g["grid_idx"], np.arange(N))
    W = len(g["waypoints"])
    vcap = _initial_velocities(g["grid_t"], W, float(g["constraints"][0]), float(g["end_vel"]), g["node_max_velocity"], g["node_stop"])
    assert np.sum(vcap == 0.01) >= 2      # the stop and the end velocity
    cap = N + 5
    row = np.zeros((1, cap))
    row[0, :N] = vcap
    for kernel in (_lib.VELOCITY_RELAX, _lib.VELOCITY_SEQ_FAST):
        v, n = _staged_velocity(torch_mod, dtype, g["waypoints"][None], g["constraints"], row, float(g["dd"]), cap, kernel)
        assert n[0] == N
        err = np.max(np.abs(v[0, :N] - g["grid_velocity"]) / g["grid_velocity"])
        print(f"feat_stop/{dtype} kernel {kernel}: velocity max rel err {err:.2e}")
        assert err <= tol


@pytest.mark.parametrize("dtype,tol", DTYPES_TOL)
def test_initial_velocities_node_limits_vs_oracle(torch_mod, dtype, tol):
    """Per-node max_velocity and stops on random 8-waypoint routes: oracle (node semantics of MPG:100-176) against
    the batched path fed with the initial-velocity rows."""
    from oracle import oracle
    from vexautonomousplanner_amd import _lib
    from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints
    rng = np.random.default_rng(404)
    B, W, dd = 6, 8, 0.005
    wp = make_waypoints(B, W, 31).astype(np.float64)
    refs, rows = [], []
    for b in range(B):
        mv = np.where(rng.random(W) < 0.4, rng.uniform(1.0, 3.5, W), 0.0)
        stop = (rng.random(W) < 0.25).astype(float)
        stop[0] = stop[-1] = 0
        nodes = dict(is_reverse=np.zeros(W), turn=np.zeros(W), stop=stop, wait_time=np.zeros(W), max_velocity=mv,
                     max_acceleration=np.zeros(W), tangent=np.full((W, 2), np.nan), magnitudes=np.zeros((W, 2)))
        op = oracle.OraclePath(wp[b], nodes=nodes)
        op.rebuild_tables()
        r = op.forward_backward(DEFAULT_CONSTRAINTS, dd=dd)
        refs.append(r["velocity"])
        rows.append(_initial_velocities(r["t"], W, DEFAULT_CONSTRAINTS[0], 0.01, mv, stop))
    cap = max(len(r) for r in refs) + 3
    vcap = np.zeros((B, cap))
    for b in range(B):
        vcap[b, :len(rows[b])] = rows[b]
    v, n = _staged_velocity(torch_mod, dtype, wp, DEFAULT_CONSTRAINTS, vcap, dd, cap, _lib.VELOCITY_RELAX)
    for b in range(B):
        N = len(refs[b])
        assert n[b] == N
        err = np.max(np.abs(v[b, :N] - refs[b]) / refs[b])
        assert err <= tol, (b, err)


@pytest.mark.parametrize("dtype,tol", DTYPES_TOL)
def test_apply_node_limits_stop_golden(torch_mod, gens, dtype, tol):
    """feat_stop (real reference): profile() then apply_node_limits(node_stop=...) on the reference's grid."""
    g = gu.load("feat_stop")
    N = int(g["n_samples"])
    gen = gens[dtype]
    wp = torch_mod.tensor(g["waypoints"][None], dtype=gen.tdtype, device=gen.device)
    r = gen.profile(wp, g["constraints"], dd=float(g["dd"]), capacity=N + 9)
    gen.apply_node_limits(r, g["constraints"], node_stop=g["node_stop"][None])
    torch_mod.cuda.synchronize()
    assert int(r["meta"][0, 3]) == N and int(r["flags"][0]) == 0
    v = r["velocity"][0, :N].cpu().numpy().astype(np.float64)
    err = np.max(np.abs(v - g["grid_velocity"]) / g["grid_velocity"])
    assert err <= tol, err
    # the stop takes effect at the first sample whose parameter has reached node 3
    k = int(r["node_sample"][0, 3])
    assert g["grid_t"][k] >= 3.0 > g["grid_t"][k - 1]
    assert float(r["vcap"][0, k]) == pytest.approx(0.01)


@pytest.mark.parametrize("dtype,tol", DTYPES_TOL)
@pytest.mark.parametrize("use_dd", [True, False])
def test_apply_node_limits_vs_oracle(torch_mod, gens, dtype, tol, use_dd):
    """Random routes with node max_velocity / stop and action points (max_velocity / stop): the batched path
    (profile + apply_node_limits) against the oracle's forward_backward with the same nodes (MPG:100-176),
    on the reference's grid and on the fixed-S grid."""
    from oracle import oracle
    from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints
    rng = np.random.default_rng(2024)
    B, W = 7, 8
    wp = make_waypoints(B, W, 52).astype(np.float64)
    gen = gens[dtype]
    mv = np.where(rng.random((B, W)) < 0.4, rng.uniform(1.0, 3.5, (B, W)), 0.0)
    stop = rng.random((B, W)) < 0.25
    stop[:, 0] = stop[:, -1] = False
    aps = []
    for b in range(B):
        n = int(rng.integers(0, 4))
        ts = np.sort(rng.uniform(0.2, W - 1.2, size=n))
        aps.append([{"t": float(t), "max_velocity": float(rng.uniform(1.0, 3.0)) if rng.random() < 0.6 else 0.0,
                     "stop": bool(rng.random() < 0.4)} for t in ts])
    aps[0] = [{"t": 2.0, "max_velocity": 1.5, "stop": False}]       # an action point exactly at a node's parameter
    S = 3000
    refs = []
    for b in range(B):
        nodes = dict(is_reverse=np.zeros(W), turn=np.zeros(W), stop=stop[b].astype(float), wait_time=np.zeros(W),
                     max_velocity=mv[b], max_acceleration=np.zeros(W), tangent=np.full((W, 2), np.nan), magnitudes=np.zeros((W, 2)))
        actions = None
        if aps[b]:
            actions = dict(t=np.array([a["t"] for a in aps[b]]), stop=np.array([float(a["stop"]) for a in aps[b]]),
                           wait_time=np.zeros(len(aps[b])), max_velocity=np.array([a["max_velocity"] for a in aps[b]]),
                           max_acceleration=np.zeros(len(aps[b])))
        op = oracle.OraclePath(wp[b], nodes=nodes, actions=actions)
        op.rebuild_tables()
        dd = 0.005 if use_dd else op.dd_for_samples(S)
        refs.append(op.forward_backward(DEFAULT_CONSTRAINTS, dd=dd)["velocity"])
    t = torch_mod.tensor(wp, dtype=gen.tdtype, device=gen.device)
    if use_dd:
        r = gen.profile(t, DEFAULT_CONSTRAINTS, dd=0.005, capacity=max(len(v) for v in refs) + 4)
    else:
        r = gen.profile(t, DEFAULT_CONSTRAINTS, samples=S)
    plain = r["velocity"].clone()
    gen.apply_node_limits(r, DEFAULT_CONSTRAINTS, node_max_velocity=mv, node_stop=stop, action_points=aps)
    torch_mod.cuda.synchronize()
    assert not r["flags"].any().item()
    got = r["velocity"].cpu().numpy().astype(np.float64)
    for b in range(B):
        N = len(refs[b])
        assert int(r["meta"][b, 3]) == N
        err = np.max(np.abs(got[b, :N] - refs[b]) / refs[b])
        assert err <= tol, (b, err)
    assert (r["velocity"] < plain * 0.9).any().item()


def _apply_golden_limits(torch_mod, gen, g):
    """profile() + apply_node_limits() for a golden route fixture (node / action-point limits, no splits)."""
    N = int(g["n_samples"])
    W = len(g["waypoints"])
    wp = torch_mod.tensor(g["waypoints"][None], dtype=gen.tdtype, device=gen.device)
    r = gen.profile(wp, g["constraints"], dd=float(g["dd"]), capacity=N + 9)
    aps = None
    if "ap_t" in g.files:
        aps = [[{"t": float(t), "max_velocity": float(mv), "max_acceleration": float(ma), "stop": bool(st)}
                for t, mv, ma, st in zip(g["ap_t"], g["ap_max_velocity"], g["ap_max_acceleration"], g["ap_stop"])]]
    gen.apply_node_limits(r, g["constraints"], node_max_velocity=g["node_max_velocity"][None], node_stop=g["node_stop"][None],
                          node_max_acceleration=g["node_max_acceleration"][None], action_points=aps)
    torch_mod.cuda.synchronize()
    assert int(r["meta"][0, 3]) == N and int(r["flags"][0]) == 0
    return r["velocity"][0, :N].cpu().numpy().astype(np.float64)


@pytest.mark.parametrize("dtype,tol", DTYPES_TOL)
@pytest.mark.parametrize("name", ["feat_limits", "feat_action", "feat_stop"])
def test_apply_node_limits_vs_reference_golden(torch_mod, gens, name, dtype, tol):
    """Golden routes of the real reference with node max_velocity / max_acceleration (feat_limits), action points
    with max_velocity / max_acceleration / stop (feat_action; its wait only concerns the time domain) and a stop
    node: forward_backward_pass through the batched path (boundary_map / max_accels quirks included)."""
    g = gu.load(name)
    assert not g["node_is_reverse_node"].any() and not g["node_turn"].any()
    v = _apply_golden_limits(torch_mod, gens[dtype], g)
    err = np.max(np.abs(v - g["grid_velocity"]) / g["grid_velocity"])
    print(f"{name}/{dtype}: velocity max rel err {err:.2e}")
    assert err <= tol


@pytest.mark.parametrize("dtype,tol", [("f32", 1e-5), ("f64", 1e-9)])
def test_apply_node_limits_with_accelerations_vs_oracle(torch_mod, gens, dtype, tol):
    """Random routes with every limit a node or action point can carry (max_velocity, max_acceleration, stop), incl.
    an action point on a node's sample (it replaces the node's boundary_map entry, MPG:162) — against the oracle.
    fp32 rows at north_star's 1e-5: the limit rows enter the fp64 recurrence in fp64 (vap_limit_rows_dtype)."""
    from oracle import oracle
    from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints
    rng = np.random.default_rng(77)
    B, W, dd = 8, 8, 0.005
    wp = make_waypoints(B, W, 91).astype(np.float64)
    gen = gens[dtype]
    mv = np.where(rng.random((B, W)) < 0.3, rng.uniform(1.0, 3.5, (B, W)), 0.0)
    ma = np.where(rng.random((B, W)) < 0.4, rng.uniform(2.0, 12.0, (B, W)), 0.0)
    stop = rng.random((B, W)) < 0.2
    stop[:, 0] = stop[:, -1] = False
    aps = []
    for b in range(B):
        ts = np.sort(rng.uniform(0.2, W - 1.2, size=int(rng.integers(0, 4))))
        aps.append([{"t": float(t), "max_velocity": float(rng.uniform(1.0, 3.0)) if rng.random() < 0.5 else 0.0,
                     "max_acceleration": float(rng.uniform(2.0, 10.0)) if rng.random() < 0.6 else 0.0,
                     "stop": bool(rng.random() < 0.3)} for t in ts])
    aps[0] = [{"t": 3.0, "max_velocity": 0.0, "max_acceleration": 3.0, "stop": False}]     # on node 3's sample
    ma[0, 3] = 9.0
    ma[1, 0], ma[1, 2], ma[2, 4] = 30.0, 40.0, 25.0   # far above max_acc: max_angular_accel/|k| (MPG:222) binds in curves
    # cases a randomised sweep (tools/fuzz_batch_routes.py) found: two action points inside one sample interval — the
    # reference looks at one pending action point per sample, so the second never takes effect, nor any after it —
    # and an action point just before a node, on the node's sample: the node is handled first whatever the parameters
    aps[3] = [{"t": 2.41490, "max_velocity": 1.2, "max_acceleration": 4.0, "stop": False},
              {"t": 2.41492, "max_velocity": 3.0, "max_acceleration": 0.0, "stop": True},
              {"t": 4.5, "max_velocity": 1.0, "max_acceleration": 3.0, "stop": True}]
    aps[4] = [{"t": 2.99995, "max_velocity": 2.0, "max_acceleration": 13.9, "stop": False}]
    stop[4, 3], mv[4, 3], ma[4, 3] = True, 3.5, 5.0
    refs = []
    for b in range(B):
        nodes = dict(is_reverse=np.zeros(W), turn=np.zeros(W), stop=stop[b].astype(float), wait_time=np.zeros(W),
                     max_velocity=mv[b], max_acceleration=ma[b], tangent=np.full((W, 2), np.nan), magnitudes=np.zeros((W, 2)))
        actions = None
        if aps[b]:
            actions = dict(t=np.array([a["t"] for a in aps[b]]), stop=np.array([float(a["stop"]) for a in aps[b]]),
                           wait_time=np.zeros(len(aps[b])), max_velocity=np.array([a["max_velocity"] for a in aps[b]]),
                           max_acceleration=np.array([a["max_acceleration"] for a in aps[b]]))
        op = oracle.OraclePath(wp[b], nodes=nodes, actions=actions)
        op.rebuild_tables()
        refs.append(op.forward_backward(DEFAULT_CONSTRAINTS, dd=dd)["velocity"])
    t = torch_mod.tensor(wp, dtype=gen.tdtype, device=gen.device)
    r = gen.profile(t, DEFAULT_CONSTRAINTS, dd=dd, capacity=max(len(v) for v in refs) + 4)
    gen.apply_node_limits(r, DEFAULT_CONSTRAINTS, node_max_velocity=mv, node_stop=stop, node_max_acceleration=ma, action_points=aps)
    torch_mod.cuda.synchronize()
    assert not r["flags"].any().item()
    got = r["velocity"].cpu().numpy().astype(np.float64)
    for b in range(B):
        N = len(refs[b])
        assert int(r["meta"][b, 3]) == N
        err = np.max(np.abs(got[b, :N] - refs[b]) / refs[b])
        assert err <= tol, (b, err)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("S", [200, 3000, 9000])
def test_acceleration_rows_relaxation_equals_sequential_sweep(torch_mod, dtype, S):
    """vap_velocity_pass_limits with per-sample max_acceleration rows: the relaxation kernel against the one-lane
    sequential sweep, bit for bit (fp64 rows beyond the relaxation kernel's reach take the sweep in both)."""
    import ctypes as C
    from vexautonomousplanner_amd import _lib
    from vexautonomousplanner_amd.batch import BatchedTrajectoryGenerator
    from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints
    torch = torch_mod
    rng = np.random.default_rng(S)
    B, W = 5, 8
    gen = make_gen(dtype)
    wp = torch.tensor(make_waypoints(B, W, 13), dtype=gen.tdtype, device=gen.device)
    r = gen.profile(wp, DEFAULT_CONSTRAINTS, samples=S)

    def steps(lo, hi):
        out = np.empty((B, S))
        for b in range(B):
            edges = np.sort(rng.integers(1, S - 1, size=4))
            out[b] = rng.uniform(lo, hi, size=5)[np.searchsorted(edges, np.arange(S), side="right")]
        return out
    rows = {"vcap": steps(1.0, 4.0), "af": steps(2.0, 12.0), "ab": steps(2.0, 12.0)}
    dec = rng.uniform(3.0, 10.0, size=B)
    ltd = torch.float64 if gen._L.vap_limit_rows_dtype(gen.ctx.handle, gen.vdtype) == _lib.VAP_F64 else torch.float32
    d = {k: torch.tensor(v, dtype=ltd, device=gen.device) for k, v in rows.items()}
    d_dec = torch.tensor(dec, dtype=ltd, device=gen.device)
    c = _lib.make_constraints(DEFAULT_CONSTRAINTS)
    p = lambda t: C.c_void_p(t.data_ptr())
    out = {}
    for kernel in (_lib.VELOCITY_SEQ_FAST, _lib.VELOCITY_AUTO, _lib.VELOCITY_SEQ_LITERAL):
        gen.ctx.set_option(_lib.OPT_VELOCITY_KERNEL, kernel)
        v = torch.empty((B, S), dtype=gen.tdtype, device=gen.device)
        _lib.check(gen._L.vap_velocity_pass_limits(gen.ctx.handle, gen.vdtype, B, S, C.byref(c), 0.01, 0.01, p(r["meta"]), p(r["curvature"]),
                                                   None, p(d["vcap"]), p(d["af"]), p(d["ab"]), p(d_dec), p(v), p(r["flags"])), "limits")
        torch.cuda.synchronize()
        out[kernel] = v.cpu().numpy().astype(np.float64)
    assert np.array_equal(out[_lib.VELOCITY_SEQ_FAST], out[_lib.VELOCITY_AUTO])
    lit = out[_lib.VELOCITY_SEQ_LITERAL]
    tol = 3e-5 if dtype == "f32r32" else (2e-7 if dtype == "f32" else 1e-9)
    assert np.max(np.abs(out[_lib.VELOCITY_AUTO] - lit) / lit) <= tol
    assert not r["flags"].any().item()


@pytest.mark.parametrize("dtype,tol", DTYPES_TOL)
@pytest.mark.parametrize("B,W,S,seed", [(64, 8, 1024, 5), (48, 32, 2000, 3), (3, 2, 300, 9), (5, 5, 257, 10)])
def test_batch_vs_oracle(torch_mod, gens, B, W, S, seed, dtype, tol):
    from oracle import oracle
    from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints
    wp = make_waypoints(B, W, seed).astype(np.float64)
    ref = oracle.profile_batch(wp, S, DEFAULT_CONSTRAINTS, n_threads=8)
    r = run_gpu(torch_mod, gens[dtype], wp, samples=S)
    assert np.all(r["flags"] == 0)
    np.testing.assert_allclose(r["meta"][:, 1], ref["total_length"], rtol=1e-14)
    check_fields(r, ref, tol, f"batch B={B} W={W} S={S}/{dtype}")


def test_truncation_flag_and_capacity(torch_mod, gens):
    g = gu.load("c1_w8")
    N = int(g["n_samples"])
    r = run_gpu(torch_mod, gens["f32"], g["waypoints"][None], dd=float(g["dd"]), capacity=N - 10)
    assert r["flags"][0] & 2
    assert int(r["meta"][0, 3]) == N - 10


def test_invalid_arguments(torch_mod, gens):
    from vexautonomousplanner_amd import _lib
    gen = gens["f32"]
    wp = torch_mod.zeros((1, 1, 2), dtype=gen.tdtype, device=gen.device)
    with pytest.raises(_lib.VapError):
        gen.profile(wp, samples=100)
    wp = torch_mod.zeros((1, 4, 2), dtype=gen.tdtype, device=gen.device)
    with pytest.raises(ValueError):
        gen.profile(wp)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,W,S,seed", [(8, 32, 10000, 3), (33, 8, 1024, 5), (5, 5, 257, 10), (4, 32, 4097, 12),
                                        (3, 2, 64, 9), (2, 16, 7001, 13)])
def test_relaxation_is_bit_identical_to_sequential_sweep(torch_mod, B, W, S, seed, dtype):
    """The speculative chunk relaxation (K5b) must reach exactly the fixed point the sequential
    sweep (K5a, same step arithmetic) computes: compare bit patterns, not tolerances."""
    from vexautonomousplanner_amd.batch import BatchedTrajectoryGenerator
    from vexautonomousplanner_amd.synth import make_waypoints
    wp = make_waypoints(B, W, seed).astype(np.float64)
    outs = {}
    for which in ("relax", "seq_fast", "seq_literal"):
        gen = make_gen(dtype, velocity_kernel=which)
        r = run_gpu(torch_mod, gen, wp, samples=S)
        assert np.all(r["flags"] == 0), which
        outs[which] = r["velocity"]
    assert np.array_equal(outs["relax"], outs["seq_fast"])
    tol = {"f32r32": 5e-6, "f32": 2e-7, "f64": 1e-9}[dtype]  # two fp32 roundings of the same recurrence
    assert np.max(np.abs(outs["seq_fast"] - outs["seq_literal"]) / outs["seq_literal"]) <= tol


@pytest.mark.parametrize("dtype,tol", DTYPES_TOL)
def test_relaxation_on_reference_grid_ragged_rows(torch_mod, dtype, tol):
    """dd-mode (ragged n_samples per path) through the relaxation kernel, against the oracle."""
    from oracle import oracle
    from vexautonomousplanner_amd.batch import BatchedTrajectoryGenerator
    from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints
    wp = make_waypoints(6, 8, 21).astype(np.float64)
    gen = make_gen(dtype, velocity_kernel="relax")
    r = run_gpu(torch_mod, gen, wp, dd=0.005, capacity=2048)
    for b in range(len(wp)):
        p = oracle.OraclePath(wp[b])
        p.rebuild_tables()
        ref = p.forward_backward(DEFAULT_CONSTRAINTS, 0.005)
        N = len(ref["velocity"])
        assert int(r["meta"][b, 3]) == N
        got = {k: r[k][b][:N] for k in ("x", "y", "heading", "curvature", "velocity")}
        check_fields(got, ref, tol, f"ragged path {b}/{dtype}")
        assert np.all(r["velocity"][b][N:] == 0)


@pytest.mark.parametrize("dtype,tol", DTYPES_TOL)
def test_config2_million_samples_vs_reference_golden(torch_mod, gens, dtype, tol):
    """BASELINE config 2: one 256-waypoint path at 1e6 samples.  Four samples share each table entry
    here, so the reference's divide-by-zero semantics (inf clamps the acceleration, NaN is skipped,
    -inf lifts the limit in the backward pass) decide the profile; compared point for point with the
    reference's own output on a strided + windowed subset, and through its velocity checksum."""
    g = gu.load("c2_w256_S1000000")
    S = int(g["samples"])
    r = run_gpu(torch_mod, gens[dtype], g["waypoints"][None], samples=S, constraints=g["constraints"])
    assert r["flags"][0] == 0
    gi = g["grid_idx"]
    got = {k: r[k][0][gi] for k in ("x", "y", "heading", "curvature", "velocity")}
    ref = {"x": g["grid_x"], "y": g["grid_y"], "heading": g["grid_heading"],
           "curvature": g["grid_curvature"], "velocity": g["grid_velocity"]}
    check_fields(got, ref, tol, f"c2 1e6/{dtype}")
    assert abs(np.sum(r["velocity"][0]) - float(g["velocity_sum"])) <= tol * float(g["velocity_sum"])
    # many consecutive samples must be exact repeats (shared table entry) - the regime this test is about
    k = r["curvature"][0]
    assert np.mean(k[1:] == k[:-1]) > 0.5


@pytest.mark.parametrize("dtype", DTYPES)
def test_dup_path_relaxation_matches_sequential(torch_mod, dtype):
    """A dense grid short enough for the relaxation kernel (samples sharing table entries): the
    DUP step forms of the relaxation and the sequential sweep must agree bit for bit."""
    from vexautonomousplanner_amd.batch import BatchedTrajectoryGenerator
    from vexautonomousplanner_amd.synth import make_waypoints
    wp = make_waypoints(3, 2, 31).astype(np.float64) * 1.0
    wp3 = make_waypoints(3, 3, 32).astype(np.float64)
    for w, S in ((wp3, 8000), (wp, 7000)):
        outs = {}
        for which in ("relax", "seq_fast", "seq_literal"):
            gen = make_gen(dtype, velocity_kernel=which)
            r = run_gpu(torch_mod, gen, w, samples=S)
            assert np.all(r["flags"] == 0)
            outs[which] = r["velocity"]
            kk = r["curvature"]
        assert np.array_equal(outs["relax"], outs["seq_fast"])
        tol = {"f32r32": 5e-6, "f32": 2e-7, "f64": 1e-9}[dtype]
        assert np.max(np.abs(outs["seq_fast"] - outs["seq_literal"]) / outs["seq_literal"]) <= tol


@pytest.mark.parametrize("dtype", DTYPES)
# (2, 3, 30001): a dense grid — most samples share a table entry with their neighbour, the sign-aware backward step
@pytest.mark.parametrize("B,W,S,seed", [(2, 64, 30001, 41), (1, 32, 20481, 42), (3, 16, 45000, 43), (2, 3, 30001, 32)])
def test_long_row_relaxation_is_bit_identical_to_sequential_sweep(torch_mod, B, W, S, seed, dtype):
    """Rows beyond the register-resident kernel go through the two-level relaxation (K5c): "relax" hands the
    super-chunk interfaces on by look-back inside one launch per direction, "relax_rounds" is the earlier form with one
    launch per super-round and a host check — both must equal the sequential sweep bit for bit."""
    from vexautonomousplanner_amd.batch import BatchedTrajectoryGenerator
    from vexautonomousplanner_amd.synth import make_waypoints
    wp = make_waypoints(B, W, seed).astype(np.float64)
    outs = {}
    for which in ("relax", "relax_rounds", "seq_fast"):
        gen = make_gen(dtype, velocity_kernel=which)
        r = run_gpu(torch_mod, gen, wp, samples=S)
        assert np.all(r["flags"] == 0), which
        outs[which] = r["velocity"]
        kk = r["curvature"]
    if W == 3:
        assert np.mean(kk[:, 1:] == kk[:, :-1]) > 0.5
    assert np.array_equal(outs["relax"], outs["seq_fast"])
    assert np.array_equal(outs["relax_rounds"], outs["seq_fast"])


@pytest.mark.parametrize("dtype", DTYPES)
def test_long_rows_more_super_chunks_than_the_chip_holds(torch_mod, dtype):
    """The look-back kernel waits only for workgroups with lower tickets, so a grid that is not resident all at once
    (here 48 paths x 59 or 24 super-chunks of one wavefront each; the chip holds 2048 such wavefronts, 1024 for the fp32
    kernel) drains in ticket order — and repeated calls on one context start from zeroed records every time."""
    from vexautonomousplanner_amd.synth import make_waypoints
    B, W, S = 48, 24, 60001
    wp = make_waypoints(B, W, 77).astype(np.float64)
    ref = run_gpu(torch_mod, make_gen(dtype, velocity_kernel="seq_fast"), wp, samples=S)
    gen = make_gen(dtype, velocity_kernel="relax")
    for rep in range(3):
        r = run_gpu(torch_mod, gen, wp, samples=S)
        assert np.all(r["flags"] == 0), rep
        assert np.array_equal(r["velocity"], ref["velocity"]), rep


@pytest.mark.parametrize("dtype", DTYPES)
def test_long_ragged_rows_look_back(torch_mod, dtype):
    """Ragged long rows (the reference's own grid, a fixed dd): paths end in different super-chunks, the row capacity
    leaves super-chunks with no sample at all, and the zero tail must be written."""
    from vexautonomousplanner_amd.synth import make_waypoints
    wp = make_waypoints(5, 12, 91).astype(np.float64)
    wp[1] *= 0.55
    wp[3] *= 0.8
    outs = {}
    for which in ("relax", "seq_fast"):
        gen = make_gen(dtype, velocity_kernel=which)
        r = run_gpu(torch_mod, gen, wp, dd=0.00015, capacity=70000)
        assert np.all(r["flags"] == 0), which
        outs[which] = r
    n = outs["relax"]["meta"][:, 3].astype(int)
    assert n.min() > 20480 and len(set(n.tolist())) > 1 and n.max() < 70000 - 2600
    assert np.array_equal(outs["relax"]["velocity"], outs["seq_fast"]["velocity"])
    for b in range(5):
        assert np.all(outs["relax"]["velocity"][b, n[b]:] == 0)


@pytest.mark.parametrize("B,W,S,seed", [(8, 32, 10000, 3), (5, 8, 2561, 5), (3, 16, 7001, 13), (4, 32, 4097, 12),
                                        (2, 3, 8000, 32), (6, 8, 1024, 77)])
def test_wave_per_path_relaxation_is_bit_identical(torch_mod, B, W, S, seed):
    """K5b' (one wave per path, sequential windows) against the block kernel and the sequential sweep."""
    from vexautonomousplanner_amd.batch import BatchedTrajectoryGenerator
    from vexautonomousplanner_amd.synth import make_waypoints
    wp = make_waypoints(B, W, seed).astype(np.float64)
    outs = {}
    for which in ("relax_wave", "relax_block", "seq_fast"):
        gen = make_gen("f32r32", velocity_kernel=which)
        r = run_gpu(torch_mod, gen, wp, samples=S)
        assert np.all(r["flags"] == 0), which
        outs[which] = r["velocity"]
    assert np.array_equal(outs["relax_wave"], outs["seq_fast"])
    assert np.array_equal(outs["relax_block"], outs["seq_fast"])


def test_wave_per_path_ragged_rows(torch_mod):
    from vexautonomousplanner_amd.batch import BatchedTrajectoryGenerator
    from vexautonomousplanner_amd.synth import make_waypoints
    wp = make_waypoints(6, 8, 21).astype(np.float64)
    outs = {}
    for which in ("relax_wave", "seq_fast"):
        gen = make_gen("f32r32", velocity_kernel=which)
        r = run_gpu(torch_mod, gen, wp, dd=0.002, capacity=4000)
        assert np.all(r["flags"] == 0)
        outs[which] = r["velocity"]
    assert np.array_equal(outs["relax_wave"], outs["seq_fast"])


def test_edge_sizes_and_degenerate_inputs(torch_mod, gens):
    """Minimum and maximum sizes, odd row lengths, coincident waypoints, non-finite input: defined
    outputs and flags, never a hang or a fault."""
    from oracle import oracle
    from vexautonomousplanner_amd import _lib
    from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints
    gen = gens["f32"]
    torch = torch_mod
    # smallest row: 2 samples (start and end sample only)
    wp = make_waypoints(3, 4, 50).astype(np.float64)
    r = run_gpu(torch, gen, wp, samples=2)
    assert np.all(r["flags"] == 0) and np.allclose(r["velocity"], 0.01)
    # odd row lengths exercise the unaligned store / stage paths
    for S in (3, 5, 63, 257, 1021, 4099):
        ref = oracle.profile_batch(wp, S, DEFAULT_CONSTRAINTS)
        r = run_gpu(torch, gen, wp, samples=S)
        assert np.all(r["flags"] == 0)
        check_fields(r, ref, 1e-5, f"odd S={S}")
    # widest supported path: 2048 waypoints (k_fit LDS limit); one more is refused, not truncated
    wide = make_waypoints(1, 2048, 51).astype(np.float64)
    ref = oracle.profile_batch(wide, 3000, DEFAULT_CONSTRAINTS)
    r = run_gpu(torch, gen, wide, samples=3000)
    assert np.all(r["flags"] == 0)
    check_fields(r, ref, 1e-5, "W=2048")
    with pytest.raises(_lib.VapError) as e:
        gen.profile(torch.zeros((1, 2049, 2), dtype=gen.tdtype, device=gen.device), samples=100)
    assert e.value.status == _lib.VAP_ERR_UNSUPPORTED
    # coincident waypoints: the reference produces NaNs and then raises; here the path is flagged
    bad = wp.copy()
    bad[1, 2] = bad[1, 1]
    r = run_gpu(torch, gen, bad, samples=300)
    assert r["flags"][1] & _lib.FLAG_DEGENERATE
    assert r["flags"][0] == 0 and r["flags"][2] == 0
    good = oracle.profile_batch(wp[[0, 2]], 300, DEFAULT_CONSTRAINTS)
    np.testing.assert_allclose(r["velocity"][[0, 2]], good["velocity"], rtol=1e-5)   # neighbours unaffected
    # non-finite waypoint: flagged, finite rows elsewhere, and the call returns
    nanp = wp.copy()
    nanp[2, 1, 0] = np.nan
    r = run_gpu(torch, gen, nanp, samples=300)
    assert r["flags"][2] & _lib.FLAG_DEGENERATE
    assert np.all(np.isfinite(r["velocity"][:2]))


@pytest.mark.parametrize("dtype,tol", DTYPES_TOL)
@pytest.mark.parametrize("cons", [(4.0, 12.0, 6.0, 0.8, 16.0, 12.5 / 12), (4.0, 6.0, 12.0, 0.8, 16.0, 12.5 / 12),
                                  (7.0, 6.8, 3.7, 0.8, 16.0, 0.70), (2.2, 15.0, 5.3, 0.8, 16.0, 1.14),
                                  (5.6, 11.9, 2.8, 0.8, 16.0, 1.18)])
def test_batch_vs_oracle_other_robots(torch_mod, gens, cons, dtype, tol):
    """max_dec != max_acc and other limits, every velocity kernel: the oracle (pinned on this by the cons_*
    vectors of the real reference) decelerates with max_acc, MPG:110, 194-196."""
    from oracle import oracle
    from vexautonomousplanner_amd.synth import make_waypoints
    for W, S, seed in ((8, 1000, 31), (32, 10000, 32), (2, 64, 33), (13, 30000, 34)):
        wp = make_waypoints(3, W, seed).astype(np.float64)
        ref = oracle.profile_batch(wp, S, cons, n_threads=8)
        for mode in ("auto", "seq_fast", "seq_literal") if S <= 10000 else ("auto",):
            gens[dtype].set_velocity_kernel(mode)
            try:
                r = run_gpu(torch_mod, gens[dtype], wp, samples=S, constraints=cons)
            finally:
                gens[dtype].set_velocity_kernel("auto")
            assert not r["flags"].any()
            # fp32: the finite-difference angular-acceleration term is formed from fp32 curvatures and heading
            # steps; its weight grows with max_acc, and at max_acc = 12-15 ft/s^2 the worst sample of a tight
            # curve sits right at 1e-5 (literal and fast form alike) where the default robot stays below 2.3e-6
            check_fields({k: r[k] for k in ("x", "y", "heading", "curvature", "velocity")}, ref,
                         3e-5 if dtype == "f32r32" else tol, f"{cons} W={W} S={S} {mode}/{dtype}")


# ---- batched time-domain resample (vap_time_profile; SURVEY §8(f)-1 at batch scale) -----------------
def _time_profile(dtype, wp, dt=0.01, cap=4096, cons=None):
    import torch
    from vexautonomousplanner_amd.batch import BatchedTrajectoryGenerator
    from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS
    cons = DEFAULT_CONSTRAINTS if cons is None else cons
    gen = make_gen(dtype)
    t = torch.tensor(wp, device="cuda:0", dtype=torch.float64 if dtype == "f64" else torch.float32)
    res = gen.profile(t, cons, dd=0.005, capacity=16384)
    tp = gen.time_profile(res, cons, dt=dt, capacity_rows=cap)
    torch.cuda.synchronize()
    return ({k: v.cpu().numpy() for k, v in tp.items()}, res["flags"].cpu().numpy())


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c1_w8"] + [n for n in gu.names("cons_") if n.endswith("_w8")])
def test_time_profile_batch_matches_reference_golden(name):
    """The reference's own 9-tuple (generate_motion_profile) for config 1 and for the robots with
    max_dec != max_acc (the time loop is where max_dec acts, MPG:572-573), through the batched kernels with
    the path placed between two others."""
    from vexautonomousplanner_amd.synth import make_waypoints
    g = gu.load(name)
    others = make_waypoints(2, 8, 77).astype(np.float64)
    wp = np.concatenate([others[:1], g["waypoints"][None], others[1:]], axis=0)
    tp, flags = _time_profile("f64", wp, cons=[float(v) for v in g["constraints"]])
    assert not flags.any()
    T = len(g["profile_times"])
    assert int(tp["counts"][1, 0]) == T
    rows = tp["rows"][1, :T]
    nm = g["profile_nodes_map"]
    assert int(tp["counts"][1, 1]) == len(nm)
    assert [int(v) for v in tp["nodes_map"][1, :len(nm)]] == [int(v) for v in nm]
    for col, key in ((0, "times"), (1, "positions"), (2, "linear_vels"), (3, "accelerations"), (4, "headings"),
                     (5, "angular_vels")):
        ref = g["profile_" + key]
        err = np.max(np.abs(rows[:, col] - ref) / np.maximum(np.abs(ref), 1.0))
        assert err <= 1e-8, (key, err)
    assert np.max(np.abs(rows[:, 6:8] - g["profile_coords"])) <= 1e-8


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["feat_limits", "feat_stop"])
def test_time_profile_of_limited_route_matches_reference_golden(name):
    """generate_motion_profile of the real reference for routes whose nodes carry max_velocity / max_acceleration
    (feat_limits) or a stop (feat_stop), no waits or splits: profile -> apply_node_limits -> time_profile in the
    batched path gives the reference's rows and nodes_map."""
    import torch
    from vexautonomousplanner_amd.batch import BatchedTrajectoryGenerator
    g = gu.load(name)
    assert not g["node_wait_time"].any() and not g["node_turn"].any() and not g["node_is_reverse_node"].any()
    cons = [float(v) for v in g["constraints"]]
    gen = BatchedTrajectoryGenerator(0, "f64")
    wp = torch.tensor(g["waypoints"][None], device="cuda:0", dtype=torch.float64)
    res = gen.profile(wp, cons, dd=0.005, capacity=16384)
    gen.apply_node_limits(res, cons, node_max_velocity=g["node_max_velocity"][None], node_stop=g["node_stop"][None],
                          node_max_acceleration=g["node_max_acceleration"][None])
    tp = gen.time_profile(res, cons, dt=0.01, capacity_rows=4096)
    torch.cuda.synchronize()
    assert not res["flags"].any().item()
    T = len(g["profile_times"])
    assert int(tp["counts"][0, 0]) == T
    rows = tp["rows"][0, :T].cpu().numpy()
    nm = g["profile_nodes_map"]
    assert [int(v) for v in tp["nodes_map"][0, :int(tp["counts"][0, 1])]] == [int(v) for v in nm]
    for col, key in ((0, "times"), (1, "positions"), (2, "linear_vels"), (3, "accelerations"), (4, "headings"),
                     (5, "angular_vels")):
        ref = g["profile_" + key]
        err = np.max(np.abs(rows[:, col] - ref) / np.maximum(np.abs(ref), 1.0))
        assert err <= 1e-8, (key, err)
    assert np.max(np.abs(rows[:, 6:8] - g["profile_coords"])) <= 1e-8


def _full_route_profile(torch, g_or_route, cons, dtype="f64"):
    """profile -> apply_node_limits -> time_profile -> insert_waits for one route given as a dict of arrays
    (waypoints, node_* , ap_*); returns rows, nodes_map, actions_map as numpy."""
    from vexautonomousplanner_amd.batch import BatchedTrajectoryGenerator
    r_ = g_or_route
    gen = make_gen(dtype)
    wp = torch.tensor(np.asarray(r_["waypoints"])[None], device="cuda:0", dtype=gen.tdtype)
    aps = None
    if "ap_t" in r_ and len(r_["ap_t"]):
        aps = [[{"t": float(t), "max_velocity": float(mv), "max_acceleration": float(ma), "stop": bool(st), "wait_time": float(w)}
                for t, mv, ma, st, w in zip(r_["ap_t"], r_["ap_max_velocity"], r_["ap_max_acceleration"], r_["ap_stop"], r_["ap_wait_time"])]]
    res = gen.profile(wp, cons, dd=0.005, capacity=16384)
    gen.apply_node_limits(res, cons, node_max_velocity=np.asarray(r_["node_max_velocity"])[None], node_stop=np.asarray(r_["node_stop"])[None],
                          node_max_acceleration=np.asarray(r_["node_max_acceleration"])[None], action_points=aps)
    tp = gen.time_profile(res, cons, dt=0.01, capacity_rows=4096)
    out = gen.insert_waits(res, tp, node_wait_time=np.asarray(r_["node_wait_time"])[None], action_points=aps, dt=0.01)
    torch.cuda.synchronize()
    assert not res["flags"].any().item()
    T, nn, na = (int(v) for v in out["counts"][0])
    return (out["rows"][0, :T].cpu().numpy(), [int(v) for v in out["nodes_map"][0, :nn]], [int(v) for v in out["actions_map"][0, :na]])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["feat_wait", "feat_action", "feat_limits", "feat_stop"])
def test_full_motion_profile_of_route_matches_reference_golden(name):
    """generate_motion_profile of the real reference — rows, nodes_map, actions_map — for routes with waits at
    nodes (feat_wait, incl. node 0) and action points with limits, a stop and a wait (feat_action), through the
    batched path: profile -> apply_node_limits -> time_profile -> insert_waits."""
    import torch
    g = gu.load(name)
    assert not g["node_turn"].any() and not g["node_is_reverse_node"].any()
    route = {k: g[k] for k in g.files if k.startswith(("node_", "ap_")) or k == "waypoints"}
    rows, nmap, amap = _full_route_profile(torch, route, [float(v) for v in g["constraints"]])
    T = len(g["profile_times"])
    assert rows.shape[0] == T
    assert nmap == [int(v) for v in g["profile_nodes_map"]]
    assert amap == [int(v) for v in g["profile_actions_map"]]
    for col, key in ((0, "times"), (1, "positions"), (2, "linear_vels"), (3, "accelerations"), (4, "headings"),
                     (5, "angular_vels")):
        ref = g["profile_" + key]
        err = np.max(np.abs(rows[:, col] - ref) / np.maximum(np.abs(ref), 1.0))
        assert err <= 1e-8, (key, err)
    assert np.max(np.abs(rows[:, 6:8] - g["profile_coords"])) <= 1e-8


@pytest.mark.gpu
def test_full_motion_profile_of_random_routes_matches_oracle():
    """Random routes with every attribute the batched path covers (limits, stops, waits, action points with waits)
    against the oracle's generate_motion_profile."""
    import torch
    from oracle import oracle
    from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints
    rng = np.random.default_rng(515)
    W = 8
    for it in range(6):
        wp = make_waypoints(1, W, 700 + it)[0].astype(np.float64)
        mv = np.where(rng.random(W) < 0.3, rng.uniform(1.0, 3.5, W), 0.0)
        ma = np.where(rng.random(W) < 0.3, rng.uniform(2.0, 12.0, W), 0.0)
        stop = (rng.random(W) < 0.2).astype(float)
        stop[0] = stop[-1] = 0
        wait = np.where(rng.random(W) < 0.3, rng.uniform(0.05, 0.4, W), 0.0)
        wait[-1] = 0
        n_ap = int(rng.integers(0, 4))
        ts = np.sort(rng.uniform(0.2, W - 1.2, size=n_ap))
        ap = dict(ap_t=ts, ap_max_velocity=np.where(rng.random(n_ap) < 0.4, rng.uniform(1.0, 3.0, n_ap), 0.0),
                  ap_max_acceleration=np.where(rng.random(n_ap) < 0.4, rng.uniform(2.0, 10.0, n_ap), 0.0),
                  ap_stop=(rng.random(n_ap) < 0.3).astype(float), ap_wait_time=np.where(rng.random(n_ap) < 0.5, rng.uniform(0.05, 0.3, n_ap), 0.0))
        nodes = dict(is_reverse=np.zeros(W), turn=np.zeros(W), stop=stop, wait_time=wait, max_velocity=mv, max_acceleration=ma,
                     tangent=np.full((W, 2), np.nan), magnitudes=np.zeros((W, 2)))
        actions = dict(t=ts, stop=ap["ap_stop"], wait_time=ap["ap_wait_time"], max_velocity=ap["ap_max_velocity"],
                       max_acceleration=ap["ap_max_acceleration"]) if n_ap else None
        ref_rows, ref_n, ref_a = oracle.OraclePath(wp, nodes=nodes, actions=actions).generate_motion_profile(DEFAULT_CONSTRAINTS, dt=0.01, dd=0.005)
        route = dict(waypoints=wp, node_max_velocity=mv, node_max_acceleration=ma, node_stop=stop, node_wait_time=wait, **ap)
        rows, nmap, amap = _full_route_profile(torch, route, DEFAULT_CONSTRAINTS)
        assert rows.shape[0] == ref_rows.shape[0], (it, rows.shape[0], ref_rows.shape[0])
        assert nmap == [int(v) for v in ref_n] and amap == [int(v) for v in ref_a], it
        err = np.max(np.abs(rows - ref_rows) / np.maximum(np.abs(ref_rows), 1.0))
        assert err <= 1e-7, (it, err)


@pytest.mark.gpu
@pytest.mark.parametrize("W", [5, 8, 32])
def test_time_profile_batch_matches_oracle(W):
    """fp64 batch vs the oracle time loop path by path: same row count, same node rows, values to 1e-7
    (position errors of the distance-domain velocities, <= 4e-11 each, add up over ~10^3 time steps)."""
    from oracle import oracle
    from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints
    wp = make_waypoints(12, W, 400 + W).astype(np.float64)
    tp, flags = _time_profile("f64", wp)
    assert not flags.any()
    for b in range(wp.shape[0]):
        rows, nmap, _ = oracle.OraclePath(wp[b]).generate_motion_profile(DEFAULT_CONSTRAINTS, dt=0.01, dd=0.005)
        T = rows.shape[0]
        assert int(tp["counts"][b, 0]) == T, (b, int(tp["counts"][b, 0]), T)
        assert [int(v) for v in tp["nodes_map"][b, :int(tp["counts"][b, 1])]] == [int(v) for v in nmap]
        got = tp["rows"][b, :T]
        err = np.max(np.abs(got - rows) / np.maximum(np.abs(rows), 1.0))
        assert err <= 1e-7, (b, err)


@pytest.mark.gpu
def test_time_profile_batch_fp32_and_truncation():
    """fp32 rows (the default mode): the velocity pass leaves its fp64 velocities on the context and the time-domain
    kernels integrate those (MPG:566-584), so the rows of generate_motion_profile hold north_star's 1e-5 — row count
    equal, every column, the table-quantised heading and angular velocity included.  A too-small capacity cuts the
    path there and flags it."""
    from oracle import oracle
    from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints
    wp = make_waypoints(8, 8, 913).astype(np.float32).astype(np.float64)     # what an fp32 caller hands over
    tp, flags = _time_profile("f32", wp)
    assert not flags.any()
    for b in range(wp.shape[0]):
        rows, _, _ = oracle.OraclePath(wp[b]).generate_motion_profile(DEFAULT_CONSTRAINTS, dt=0.01, dd=0.005)
        T = int(tp["counts"][b, 0])
        assert T == rows.shape[0]
        err = np.abs(tp["rows"][b, :T] - rows) / np.maximum(np.abs(rows), 1.0)
        assert err.max() <= 1e-5, (b, err.max(axis=0))
    tp, flags = _time_profile("f64", wp, cap=100)
    assert (tp["counts"][:, 0] == 100).all()
    assert (flags & 2).all()          # VAP_FLAG_TRUNCATED


@pytest.mark.gpu
@pytest.mark.parametrize("B", [6, 2100])      # the relaxation kernel's batch sizes, and the lane-per-path kernel's
def test_time_profile_integrates_the_row_as_the_caller_left_it(torch_mod, B):
    """The time domain integrates the velocity row it is handed, as it is AT THAT CALL (MPG:566-584 reads the list):
    behind fp32 rows the context only adds the sub-rounding residual of its fp64 recurrence.  A caller that edits the
    row in place between the velocity pass and vap_time_profile (here: scales it by 0.6) gets the edited profile — the
    same rows as for a copy of the edited row at an address the context has never seen — and not the cached one; and
    with VAP_OPT_TIME_DOMAIN_RESIDUAL off the call works on the plain fp32 row."""
    from vexautonomousplanner_amd.batch import BatchedTrajectoryGenerator
    from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints
    torch = torch_mod
    W, S, cap = 8, 700, 1024
    wp = torch.tensor(make_waypoints(B, W, 31), device="cuda:0", dtype=torch.float32)
    gen = BatchedTrajectoryGenerator(0, "f32")
    res = gen.profile(wp, DEFAULT_CONSTRAINTS, samples=S)
    tp0 = {k: v.clone() for k, v in gen.time_profile(res, DEFAULT_CONSTRAINTS, capacity_rows=cap).items()}
    res["velocity"].mul_(0.6)                                   # the caller's edit, in place
    tp1 = {k: v.clone() for k, v in gen.time_profile(res, DEFAULT_CONSTRAINTS, capacity_rows=cap).items()}
    moved = type(res)(res)                                      # the same rows at another address: no residual applies
    moved.generation = res.generation
    moved["velocity"] = res["velocity"].clone()
    tp2 = gen.time_profile(moved, DEFAULT_CONSTRAINTS, capacity_rows=cap)
    torch.cuda.synchronize()
    n0, n1, n2 = (t["counts"][:, 0].cpu().numpy() for t in (tp0, tp1, tp2))
    assert (n1 > n0 * 1.3).all(), "a slower row takes more time steps: the cached velocities were integrated instead"
    assert np.array_equal(n1, n2)
    for b in range(0, B, max(1, B // 6)):
        r1, r2 = tp1["rows"][b, :n1[b]].cpu().numpy(), tp2["rows"][b, :n2[b]].cpu().numpy()
        err = np.max(np.abs(r1[:, 1:3] - r2[:, 1:3]) / np.maximum(np.abs(r2[:, 1:3]), 1e-2))
        assert err <= 1e-5, (b, err)      # (position, velocity: the residual is below the row's rounding)
        assert np.max(np.abs(r1[:, 3] - r2[:, 3])) <= 1e-3   # acceleration = a velocity difference / dt: 1e-7 / 0.01 of noise
    plain = BatchedTrajectoryGenerator(0, "f32", time_domain_residual=False)
    res_p = plain.profile(wp, DEFAULT_CONSTRAINTS, samples=S)
    res_p["velocity"].mul_(0.6)
    tp3 = plain.time_profile(res_p, DEFAULT_CONSTRAINTS, capacity_rows=cap)
    torch.cuda.synchronize()
    assert torch.equal(res_p["velocity"], res["velocity"])      # the option does not touch the velocity rows
    assert torch.equal(tp3["counts"], tp2["counts"]) and torch.equal(tp3["rows"][0, :n2[0]], tp2["rows"][0, :n2[0]])


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,resid", [("f32", True), ("f32", False), ("f64", False)])
def test_time_profile_quad_kernel_is_bit_identical(torch_mod, dtype, resid):
    """VAP_OPT_TIME_KERNEL: four lanes per path (what AUTO takes up to 16384 paths), the same with the geometry in the
    workgroup (FUSED: AUTO's choice up to 16 paths per CU) and one lane per path walk the same recurrence (MPG:566-584) and
    fill the same geometry columns — same counts, maps and row bits, for batch sizes that leave quads and wavefronts partly
    empty, a capacity that truncates, and a batch above AUTO's switch."""
    from vexautonomousplanner_amd.batch import BatchedTrajectoryGenerator
    from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints
    torch = torch_mod
    td = torch.float64 if dtype == "f64" else torch.float32
    gen = BatchedTrajectoryGenerator(0, dtype, time_domain_residual=resid)
    for B, W, S, cap in ((1, 4, 900, 1024), (17, 8, 1500, 2048), (67, 6, 700, 60), (16500, 5, 300, 768)):
        wp_np = make_waypoints(B, W, 77 + B)
        if B == 17:
            wp_np[3] = wp_np[3, :1]          # a path of zero length: no rows, one nodes_map entry, in every kernel
        wp = torch.tensor(wp_np, device="cuda:0", dtype=td)
        res = gen.profile(wp, DEFAULT_CONSTRAINTS, samples=S)
        got = {}
        for k in ("lane", "quad", "fused", "auto"):
            gen.set_time_kernel(k)
            got[k] = {n: v.clone() for n, v in gen.time_profile(res, DEFAULT_CONSTRAINTS, capacity_rows=cap).items()}
        torch.cuda.synchronize()
        counts = got["lane"]["counts"].cpu().numpy()
        if B == 17:
            assert counts[3, 0] == 0 and counts[3, 1] == 1
            counts_live = np.delete(counts, 3, axis=0)
        else:
            counts_live = counts
        assert counts_live[:, 0].min() >= 60
        if cap == 60:
            assert (counts[:, 0] == 60).all()
        for k in ("quad", "fused", "auto"):
            assert torch.equal(got[k]["counts"], got["lane"]["counts"]), (B, k)
            rows_a, rows_b = got["lane"]["rows"].cpu().numpy(), got[k]["rows"].cpu().numpy()
            nm_a, nm_b = got["lane"]["nodes_map"].cpu().numpy(), got[k]["nodes_map"].cpu().numpy()
            for b in range(0, B, max(1, B // 300)):
                assert rows_a[b, :counts[b, 0]].tobytes() == rows_b[b, :counts[b, 0]].tobytes(), (B, k, b)
                assert np.array_equal(nm_a[b, :counts[b, 1]], nm_b[b, :counts[b, 1]]), (B, k, b)


@pytest.mark.gpu
def test_time_profile_argument_errors(torch_mod):
    """vap_time_profile through the C-ABI: status codes for a context without tables, a shape that does not
    match the last batch, and null / non-positive arguments."""
    import ctypes as C
    from vexautonomousplanner_amd import _lib
    from vexautonomousplanner_amd.batch import BatchedTrajectoryGenerator
    from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints
    torch = torch_mod
    L = _lib.lib()
    c = _lib.make_constraints(DEFAULT_CONSTRAINTS)
    B, W, S, cap = 4, 8, 512, 64
    vel = torch.zeros((B, S), dtype=torch.float32, device="cuda:0")
    meta = torch.zeros((B, 4), dtype=torch.float64, device="cuda:0")
    rows = torch.zeros((B, cap, 8), dtype=torch.float64, device="cuda:0")
    counts = torch.zeros((B, 2), dtype=torch.int32, device="cuda:0")
    nmap = torch.zeros((B, W), dtype=torch.int32, device="cuda:0")

    def call(ctx, b=B, w=W, dt=0.01, rows_p=rows, cap_rows=cap):
        return L.vap_time_profile(ctx.handle, _lib.VAP_F32, b, w, S, None, None, C.c_void_p(meta.data_ptr()),
                                  C.c_void_p(vel.data_ptr()), C.byref(c), dt, cap_rows,
                                  C.c_void_p(rows_p.data_ptr()) if rows_p is not None else None,
                                  C.c_void_p(counts.data_ptr()), C.c_void_p(nmap.data_ptr()), None)

    fresh = _lib.Context(0)
    assert call(fresh) == _lib.VAP_ERR_UNFITTED           # no tables in this context yet
    gen = BatchedTrajectoryGenerator(0, "f32")
    wp = torch.tensor(make_waypoints(B, W, 5), device="cuda:0", dtype=torch.float32)
    res = gen.profile(wp, DEFAULT_CONSTRAINTS, samples=S)
    assert call(gen.ctx, b=B + 1) == _lib.VAP_ERR_UNFITTED    # tables are for another batch shape
    assert call(gen.ctx, dt=0.0) == _lib.VAP_ERR_INVALID
    assert call(gen.ctx, cap_rows=0) == _lib.VAP_ERR_INVALID
    assert call(gen.ctx, rows_p=None) == _lib.VAP_ERR_INVALID
    tp = gen.time_profile(res, DEFAULT_CONSTRAINTS, capacity_rows=2048)
    torch.cuda.synchronize()
    assert int(tp["counts"][:, 0].min().item()) > 10


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", DTYPES)
def test_staged_api_equals_fused_call(torch_mod, dtype):
    """vap_fit -> vap_build_lut -> vap_sample -> vap_velocity_pass -> vap_time_profile with caller-owned
    buffers on a batch of 5 paths gives the fused vap_profile_batch (+ time_profile) results bit for bit."""
    import ctypes as C
    from vexautonomousplanner_amd import _lib
    from vexautonomousplanner_amd.batch import BatchedTrajectoryGenerator
    from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints
    torch = torch_mod
    L = _lib.lib()
    dev = torch.device("cuda:0")
    td = torch.float64 if dtype == "f64" else torch.float32
    vd = _lib.VAP_F64 if dtype == "f64" else _lib.VAP_F32
    B, W, S, cap_rows = 5, 8, 2000, 1024
    wp = torch.tensor(make_waypoints(B, W, 41), device=dev, dtype=td)
    gen = make_gen(dtype)
    fused = gen.profile(wp, DEFAULT_CONSTRAINTS, samples=S)
    fused_tp = gen.time_profile(fused, DEFAULT_CONSTRAINTS, capacity_rows=cap_rows)
    torch.cuda.synchronize()

    ctx = _lib.Context(0)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    ctx.set_option(_lib.OPT_F32_RECURRENCE, _lib.RECURRENCE_F32 if dtype == "f32r32" else _lib.RECURRENCE_F64)
    c = _lib.make_constraints(DEFAULT_CONSTRAINTS)
    p = lambda t: C.c_void_p(t.data_ptr())
    seg = torch.empty((B, W - 1, 6, 2), dtype=torch.float64, device=dev)
    seglen = torch.empty((B, W - 1), dtype=torch.float64, device=dev)
    meta = torch.zeros((B, 4), dtype=torch.float64, device=dev)
    flags = torch.zeros((B,), dtype=torch.int32, device=dev)
    lut = torch.empty((B, _lib.LUT_SAMPLES), dtype=torch.float64, device=dev)
    out = {k: torch.empty((B, S), dtype=td, device=dev) for k in ("x", "y", "heading", "curvature", "dtheta", "velocity")}
    _lib.check(L.vap_fit(ctx.handle, vd, B, W, p(wp), None, None, p(seg), p(seglen), p(meta), p(flags)), "vap_fit")
    _lib.check(L.vap_build_lut(ctx.handle, B, W, p(seg), p(lut), p(meta), p(flags)), "vap_build_lut")
    _lib.check(L.vap_sample(ctx.handle, vd, B, W, S, 0.0, p(seg), p(lut), p(meta), p(out["x"]), p(out["y"]), p(out["heading"]),
                            p(out["curvature"]), p(out["dtheta"]), p(flags)), "vap_sample")
    # "f32": the fp64 recurrence reads the fp64 rows vap_sample left on the context (d_dtheta = NULL)
    _lib.check(L.vap_velocity_pass(ctx.handle, vd, B, S, C.byref(c), 0.01, 0.01, p(meta), p(out["curvature"]),
                                   None if dtype == "f32" else p(out["dtheta"]), None, p(out["velocity"]), p(flags)), "vap_velocity_pass")
    rows = torch.zeros((B, cap_rows, 8), dtype=torch.float64, device=dev)
    counts = torch.zeros((B, 2), dtype=torch.int32, device=dev)
    nmap = torch.zeros((B, W), dtype=torch.int32, device=dev)
    _lib.check(L.vap_time_profile(ctx.handle, vd, B, W, S, p(seg), p(lut), p(meta), p(out["velocity"]), C.byref(c), 0.01, cap_rows,
                                  p(rows), p(counts), p(nmap), p(flags)), "vap_time_profile")
    # (the time-domain call above integrates the fp64 velocities the default-mode velocity pass left on the context for
    # exactly this fp32 row — so the all-fp32 pass below, which supersedes them, comes after it)
    torch.cuda.synchronize()
    if dtype == "f32":
        # with explicit fp32 rows the same call runs the all-fp32 recurrence: close, not identical
        v32 = torch.empty_like(out["velocity"])
        _lib.check(L.vap_velocity_pass(ctx.handle, vd, B, S, C.byref(c), 0.01, 0.01, p(meta), p(out["curvature"]), p(out["dtheta"]),
                                       None, p(v32), p(flags)), "vap_velocity_pass")
        torch.cuda.synchronize()
        assert float(((v32 - out["velocity"]).abs() / out["velocity"]).max()) < 1e-4
    assert not flags.any().item()
    for k in ("x", "y", "heading", "curvature", "velocity"):
        assert torch.equal(out[k], fused[k]), k
    assert torch.equal(meta, fused["meta"])
    assert torch.equal(counts, fused_tp["counts"])
    T = int(counts[:, 0].max().item())
    assert torch.equal(rows[:, :T], fused_tp["rows"][:, :T]) or all(
        torch.equal(rows[b, :int(counts[b, 0])], fused_tp["rows"][b, :int(counts[b, 0])]) for b in range(B))
    for b in range(B):
        assert torch.equal(nmap[b, :int(counts[b, 1])], fused_tp["nodes_map"][b, :int(counts[b, 1])])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("kw", [dict(samples=1024), dict(samples=700), dict(dd=0.02, capacity=1024)])
def test_two_paths_per_workgroup_sampling_is_bit_identical(torch_mod, dtype, kw):
    """Rows of one tile in batches of 8192 paths and more are sampled two paths per workgroup (their staging loads in
    flight together).  An odd batch of 8195 paths must give, row for row and bit for bit, what the same paths give in two
    halves (below 8192: one path per workgroup) — fixed grids, a grid shorter than the tile, ragged rows."""
    from vexautonomousplanner_amd.synth import make_waypoints
    B = 8195
    wp = make_waypoints(B, 8, 123).astype(np.float64)
    wp[1::7] *= 0.6
    gen = make_gen(dtype)
    whole = run_gpu(torch_mod, gen, wp, **kw)
    cut = 4100
    parts = [run_gpu(torch_mod, gen, wp[:cut], **kw), run_gpu(torch_mod, gen, wp[cut:], **kw)]
    assert np.all(whole["flags"] == 0)
    for k in ("x", "y", "heading", "curvature", "velocity", "meta"):
        both = np.concatenate([parts[0][k], parts[1][k]], axis=0)
        assert np.array_equal(whole[k], both, equal_nan=True), k


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_long_rows_on_two_streams_at_once(torch_mod, dtype):
    """Two look-back launches in flight on two HIP streams (two contexts), each with more super-chunks than the chip
    holds: a workgroup waits only for lower tickets of its OWN launch — workgroups that already run — so the two grids
    cannot starve each other whatever the dispatcher interleaves.  Rows equal the sequential sweep's bit for bit, no
    NOCONVERGE flag (a workgroup that gave up after its bounded wait would raise it)."""
    from vexautonomousplanner_amd.synth import make_waypoints
    B, W, S = 40, 16, 60001
    wps = [make_waypoints(B, W, 300 + i).astype(np.float64) for i in range(2)]
    refs = [run_gpu(torch_mod, make_gen(dtype, velocity_kernel="seq_fast"), w, samples=S)["velocity"] for w in wps]
    gens = [make_gen(dtype, velocity_kernel="relax") for _ in range(2)]
    streams = [torch_mod.cuda.Stream(device=gens[0].device) for _ in range(2)]
    tens = [torch_mod.tensor(w, dtype=g.tdtype, device=g.device) for w, g in zip(wps, gens)]
    torch_mod.cuda.synchronize()
    outs = [None, None]
    for rep in range(3):
        for i in range(2):
            with torch_mod.cuda.stream(streams[i]):
                outs[i] = gens[i].profile(tens[i], samples=S, out=outs[i])
        torch_mod.cuda.synchronize()
        for i in range(2):
            assert int(outs[i]["flags"].abs().sum().item()) == 0, (rep, i)
            assert np.array_equal(outs[i]["velocity"].cpu().numpy().astype(np.float64), refs[i]), (rep, i)
