"""The fused step of the default mode (vap_sample_lane.h inside k_velocity_lanes' forward producers): K3+K4's sampling
(SM:291-318, 340-346, 550-580, 204-215, MPG:112-176) with lane = sample, tables read through a 32-entry LDS window.
Every row it writes — x, y, heading, curvature, velocity, and through them the fp64 side rows — must equal the separate
sampling kernel followed by the velocity kernel BIT FOR BIT, for fixed-S and ragged (dd) grids, rows shorter and longer
than a tile, partial groups, grids denser than the arc-length table (windows that do not cover a tile), and many
waypoints (windows that cross several segments)."""
import numpy as np
import pytest

from test_gpu_parity import make_gen, run_gpu, torch_mod  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu
ROWS = ("x", "y", "heading", "curvature", "velocity")


def both(torch_mod, wp, **kw):
    fused = run_gpu(torch_mod, make_gen("f32", velocity_kernel="lanes", fused_sampling=True), wp, **kw)
    staged = run_gpu(torch_mod, make_gen("f32", velocity_kernel="lanes", fused_sampling=False), wp, **kw)
    return fused, staged


@pytest.mark.parametrize("B,W,S,seed", [(16, 32, 10000, 3), (33, 8, 1024, 5), (5, 5, 257, 10), (70, 8, 300, 11), (3, 2, 64, 9),
                                        (2, 16, 7001, 13), (4, 4, 2, 14), (3, 4, 3, 15), (1, 8, 1000, 17), (2, 64, 30001, 41),
                                        (3, 3, 8000, 32), (2, 2, 7000, 31), (2, 113, 4097, 44), (2, 300, 20000, 45)])
def test_fused_rows_equal_the_staged_kernels_bit_for_bit(torch_mod, B, W, S, seed):
    from vexautonomousplanner_amd.synth import make_waypoints
    wp = make_waypoints(B, W, seed).astype(np.float64)
    fused, staged = both(torch_mod, wp, samples=S)
    assert np.all(fused["flags"] == 0) and np.array_equal(fused["meta"], staged["meta"])
    for k in ROWS:
        assert np.array_equal(fused[k], staged[k]), k


@pytest.mark.parametrize("dd,cap", [(0.005, 2048), (0.0004, 16384), (0.02, 512)])
def test_fused_ragged_rows(torch_mod, dd, cap):
    """The reference's own grid: every path its own sample count, zeros past it; dd = 0.0004 puts several samples on one
    table entry (zero heading differences, windows that barely move)."""
    from vexautonomousplanner_amd.synth import make_waypoints
    wp = make_waypoints(21, 8, 21).astype(np.float64)[:, :5] if dd < 0.001 else make_waypoints(21, 8, 21).astype(np.float64)
    fused, staged = both(torch_mod, wp, dd=dd, capacity=cap)
    assert np.array_equal(fused["meta"], staged["meta"]) and np.array_equal(fused["flags"], staged["flags"])
    for k in ROWS:
        assert np.array_equal(fused[k], staged[k]), k


def test_fused_is_what_auto_runs_on_large_batches_and_follow_ups_work(torch_mod):
    """With VAP_OPT_FUSED_SAMPLING on, AUTO at 2048+ paths is the fused kernel; apply_node_limits (which reads the fp64 side rows the fused kernel left on
    the context) and time_profile (its fp64 velocities) give what they give after the staged kernels."""
    from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints
    torch = torch_mod
    B, W, S = 2100, 6, 600
    wp = make_waypoints(B, W, 91).astype(np.float32)
    mv = np.where(np.random.default_rng(1).random((B, W)) < 0.4, 2.0, 0.0)
    outs = []
    for fused in (True, False):
        gen = make_gen("f32", fused_sampling=fused)
        r = gen.profile(torch.tensor(wp, device=gen.device), DEFAULT_CONSTRAINTS, samples=S)
        rows0 = {k: r[k].clone() for k in ROWS}
        tp0 = {k: v.clone() for k, v in gen.time_profile(r, DEFAULT_CONSTRAINTS, capacity_rows=1024).items()}
        gen.apply_node_limits(r, DEFAULT_CONSTRAINTS, node_max_velocity=mv)
        torch.cuda.synchronize()
        outs.append((rows0, tp0, r["velocity"].clone()))
    for k in ROWS:
        assert torch.equal(outs[0][0][k], outs[1][0][k]), k
    assert torch.equal(outs[0][1]["counts"], outs[1][1]["counts"])
    T = int(outs[0][1]["counts"][:, 0].max().item())
    assert torch.equal(outs[0][1]["rows"][:, :8].nan_to_num(), outs[1][1]["rows"][:, :8].nan_to_num()) or all(
        torch.equal(outs[0][1]["rows"][b, :int(outs[0][1]["counts"][b, 0])], outs[1][1]["rows"][b, :int(outs[1][1]["counts"][b, 0])])
        for b in range(B))
    assert T > 10
    assert torch.equal(outs[0][2], outs[1][2])
