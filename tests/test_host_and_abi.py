"""CPU-only checks: the C-ABI library loads and exports exactly what include/vap.h declares, the
product refuses to run without a device (no CPU fallback), and the small host-side pieces of the
drop-in surface behave like the reference."""
import ctypes
import os
import re

import numpy as np
import pytest

import golden_util as gu
from vexautonomousplanner_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "vap.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vap_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    declared = header_functions()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/vap.h but not exported by libvap.so"
    assert sorted(_lib.EXPORTS) == declared
    assert L.vap_version() >= 100
    assert L.vap_status_string(_lib.VAP_ERR_NO_DEVICE).decode() == "no HIP device"


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    assert _lib.lib().vap_device_count() == 0
    with pytest.raises(_lib.VapError) as e:
        _lib.Context(0)
    assert e.value.status == _lib.VAP_ERR_NO_DEVICE
    from vexautonomousplanner_amd.batch import BatchedTrajectoryGenerator
    with pytest.raises(RuntimeError):
        BatchedTrajectoryGenerator(0)
    # the drop-in manager cannot fit a path either: it must raise, not quietly compute on the host
    from vexautonomousplanner_amd.nodes import Node
    from vexautonomousplanner_amd.splines.spline_manager import QuinticHermiteSplineManager
    wp = gu.load("c1_w8")["waypoints"]
    with pytest.raises(RuntimeError):
        QuinticHermiteSplineManager().build_path(wp, [Node() for _ in wp], [])


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", os.path.join(ROOT, "does_not_exist.so"))
    with pytest.raises(ImportError):
        _lib.lib()


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under the product package or dropin/ may import,
    load or link it."""
    pat = re.compile(r"(^|\W)(import\s+oracle|from\s+oracle|libvap_oracle|vap_oracle\.h|vapo_[a-z_]+)")
    for top in ("vexautonomousplanner_amd", "dropin", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                    text = open(os.path.join(dirpath, f), encoding="utf-8").read()
                    assert not pat.search(text), f"{top}/{f} references the oracle"


def test_dropin_import_paths():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "dropin"))
    try:
        for mod in ("splines", "motion_profiling_v2"):
            for k in [k for k in sys.modules if k.split(".")[0] == mod]:
                del sys.modules[k]
        from motion_profiling_v2 import motion_profile_generator
        from splines.spline_manager import PathLookupTable, QuinticHermiteSplineManager
        from splines.quintic_hermite_spline import QuinticHermiteSpline
        assert QuinticHermiteSplineManager.__module__.startswith("vexautonomousplanner_amd")
        assert hasattr(motion_profile_generator, "forward_backward_pass")
        assert hasattr(motion_profile_generator, "generate_motion_profile")
        assert PathLookupTable and QuinticHermiteSpline
    finally:
        sys.path.remove(os.path.join(ROOT, "dropin"))
        for mod in ("splines", "motion_profiling_v2"):
            for k in [k for k in sys.modules if k.split(".")[0] == mod]:
                del sys.modules[k]


def test_constraints_helpers_match_reference_formulas():
    from vexautonomousplanner_amd.motion_profiling_v2.motion_profile_generator import Constraints, get_wheel_trajectory
    c = Constraints(4.0, 8.0, 8.0, 0.8, 16.0, 12.5 / 12)
    assert c.max_speed_at_curvature(5e-7) == 4.0
    k = 1.3
    w = 2 * 4.0 / c.track_width
    assert c.max_speed_at_curvature(k) == min((w * 4.0) / (k * 4.0 + w), 4.0)
    assert c.max_accels_at_turn(0.0) == 8.0            # tie -> the "right" branch (MPG:56-59)
    assert c.max_accels_at_turn(2.0) == 8.0 - 2.0 * c.track_width / 2
    assert c.max_accels_at_turn(-2.0) == 8.0 - 2.0 * c.track_width / 2
    assert c.limit_velocity_by_ang_accel(0.0, 3.0) == 4.0
    l, r = c.get_wheel_speeds(2.0, 1.0)
    assert (l, r) == (2.0 - c.track_width / 2, 2.0 + c.track_width / 2)
    assert get_wheel_trajectory([2.0], [1.0], c.track_width) == ([l], [r])


def test_distance_grid_closed_form_equals_running_sum():
    """MPG:112-122: the reference's grid is the running sum current_dist += dd, rounding included.  The
    kernels take sample k from a closed form (csrc/vap_device.h build_grid_runs); vap_grid_distances
    exposes that form on the host and must agree with the plain sum bit for bit, count included."""
    L = _lib.lib()
    rng = np.random.default_rng(5)
    cases = [(0.005, 31.7), (0.005, 0.004), (0.005, 0.005), (2.0 ** -7, 9.0), (3 * 2.0 ** -9, 17.25), (0.7, 0.3)]
    for _ in range(120):
        mode = rng.integers(0, 4)
        if mode == 0:
            cases.append((float(rng.uniform(0.001, 0.02)), float(rng.uniform(0.5, 400.0))))
        elif mode == 1:     # dyadic spacings: the rounding ties
            cases.append((float(2.0 ** -int(rng.integers(3, 13)) * int(rng.integers(1, 8))), float(rng.uniform(1.0, 200.0))))
        elif mode == 2:     # this build's fixed-S spacing
            total = float(rng.uniform(5.0, 2000.0))
            cases.append((total / (int(rng.integers(1000, 100000)) - 1.5), total))
        else:               # spacing above the path length
            cases.append((float(rng.uniform(0.3, 1.3)), float(rng.uniform(0.2, 10.0))))
    for dd, total in cases:
        want = []
        s = 0.0
        while s < total:
            want.append(s)
            s += dd
        want = np.array(want)
        got = np.full(len(want) + 4, -1.0)
        n = ctypes.c_long(0)
        rc = L.vap_grid_distances(dd, total, len(got), got.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), ctypes.byref(n))
        assert rc == 0 and n.value == len(want), (dd, total, n.value, len(want))
        assert np.array_equal(got[:n.value], want), (dd, total)
        assert np.all(got[n.value:] == -1.0)
    n = ctypes.c_long(0)
    assert L.vap_grid_distances(0.0, 1.0, 0, None, ctypes.byref(n)) != 0
    assert L.vap_grid_distances(0.01, float("inf"), 0, None, ctypes.byref(n)) != 0
    assert L.vap_grid_distances(0.01, 2.0, 0, None, ctypes.byref(n)) == 0 and n.value == 200


def test_turn_profile_matches_oracle_time_domain_rows():
    """motion_profile_angle + generate_trapezoidal_profile against the turn rows of the reference's
    own output (feat_turn golden: the samples inserted at node 4, MPG:487-507)."""
    from vexautonomousplanner_amd.motion_profiling_v2.motion_profile_generator import Constraints, motion_profile_angle
    g = gu.load("feat_turn")
    c = Constraints(*g["constraints"])
    hs, ws = motion_profile_angle(np.radians(90), c, 0.01)
    lin = g["profile_linear_vels"]
    start = int(g["profile_nodes_map"][4])
    # the inserted rows have linear velocity 0 and carry the angular velocities of the turn profile
    rows = slice(start, start + len(ws))
    assert np.all(lin[rows] == 0)
    np.testing.assert_allclose(g["profile_angular_vels"][rows], ws, rtol=1e-12, atol=1e-12)


def test_trajectory_file_formats(tmp_path):
    """row layout, insertion of action rows, .txt and routes.h text."""
    from vexautonomousplanner_amd import trajectory_io as tio
    g = gu.load("feat_action")
    T = len(g["profile_times"])
    nodes_map = [int(v) for v in g["profile_nodes_map"]] + [T]          # gui/path.py:342
    actions_map = [int(v) for v in g["profile_actions_map"]]
    prof = (list(g["profile_times"]), list(g["profile_positions"]), list(g["profile_linear_vels"]),
            list(g["profile_accelerations"]), list(g["profile_headings"]), list(g["profile_angular_vels"]),
            nodes_map, actions_map, [row for row in g["profile_coords"]])
    node_vals = [[i, 0] for i in range(len(nodes_map))]
    ap_vals = [[9, j] for j in range(len(actions_map))]
    rows = tio.trajectory_rows(prof, node_vals, ap_vals)
    assert len(rows) == T + len(nodes_map) + len(actions_map)
    assert rows[0] == [1, 0, 0]                                          # first node's actions lead the file
    assert rows[1][0] == 0 and rows[1][1] == prof[0][0]
    assert rows[1][2] == prof[8][0][0] * 12 and rows[1][3] == prof[8][0][1] * -12
    assert rows[1][5] == prof[2][0] * 12 and rows[1][6] == prof[5][0]
    assert rows[-1] == [1, len(nodes_map) - 1, 0]                        # last node after the last step
    headers = [i for i, r in enumerate(rows) if r[0] == 1]
    assert len(headers) == len(nodes_map) + len(actions_map)
    txt = tio.format_txt(rows[:2])
    assert txt.splitlines()[0] == "1 0 0 " and txt.endswith(" \n")
    p = tmp_path / "routes.h"
    tio.update_routes_header(str(p), "alpha", rows[:3])
    tio.update_routes_header(str(p), "beta", [[1, 2], [0, 0.5, 1.0, 2.0, 3.0, 4.0, 5.0]])
    tio.update_routes_header(str(p), "alpha", rows[:2])
    text = p.read_text()
    assert text.count("std::vector<std::vector<double>> alpha =") == 1
    assert "std::vector<std::vector<double>> beta = {{1, 2}, {0, 0.5, 1.0, 2.0, 3.0, 4.0, 5.0}};" in text
    assert text.strip().endswith("#endif") and text.index("beta") < text.index("#endif")
    js = tio.route_json([[1.5, -2.0, 1, 0, 0, 0, 0, 0, None, None, None]], [[0.0, 1.0, 2.5, 0, 0]])
    assert js == "[[[1.5,-2.0,1,0,0,0,0,0,null,null,null]],[[0.0,1.0,2.5,0,0]]]"
    assert tio.parse_route_json(js)[1][0][2] == 2.5


# ------------------------------------------------------------------------------------------------
# file formats against a file the reference itself wrote (src/routes.h of the reference; the
# fixture keeps its skeleton and the exact text of a slice of its rows — oracle/gen_routes_fixture.py)
# ------------------------------------------------------------------------------------------------
def _routes_fixture():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "files", "routes_h_slice.json")) as f:
        return json.load(f)


def _parse_row(text):
    import re
    return [int(tok) if re.fullmatch(r"-?\d+", tok) else float(tok) for tok in text.split(", ")]


def test_routes_header_entry_reproduces_the_reference_file_text():
    from vexautonomousplanner_amd import trajectory_io as tio
    fx = _routes_fixture()
    rows = [_parse_row(t) for t in fx["rows_text"]]
    want = "std::vector<std::vector<double>> %s = {%s};\n" % (fx["name"], ", ".join("{" + t + "}" for t in fx["rows_text"]))
    assert tio.routes_header_entry(fx["name"], rows) == want
    # the same values as a .txt trajectory: one row per line, every value followed by one space (gui_manager.py:220-230)
    txt = tio.format_txt(rows)
    assert txt.split("\n")[:-1] == [t.replace(", ", " ") + " " for t in fx["rows_text"]]
    # shape of what the reference wrote: time steps of 25 ms in 7-value rows, action rows first and last
    steps = [r for r in rows if r[0] == 0]
    assert all(len(r) == 7 for r in rows) and rows[0][0] == 1 and rows[-1][0] == 1
    assert abs(steps[1][1] - steps[0][1] - 0.025) < 1e-12


def test_update_routes_header_writes_the_reference_skeleton(tmp_path):
    from vexautonomousplanner_amd import trajectory_io as tio
    fx = _routes_fixture()
    rows = [_parse_row(t) for t in fx["rows_text"]]
    entry = tio.routes_header_entry(fx["name"], rows)
    path = tmp_path / "routes.h"
    tio.update_routes_header(str(path), fx["name"], rows)          # new file
    want = "\n".join(entry[:-1] if l == "<ENTRY>" else l for l in fx["skeleton"])
    assert path.read_text() == want
    tio.update_routes_header(str(path), fx["name"], rows[:10])     # replace in place: still one entry, same skeleton
    lines = path.read_text().split("\n")
    assert [("<ENTRY>" if l.startswith("std::vector") else l) for l in lines] == fx["skeleton"]
    tio.update_routes_header(str(path), "other", rows[:3])         # a second route goes in front of #endif
    lines = path.read_text().split("\n")
    assert sum(l.startswith("std::vector") for l in lines) == 2 and lines[-2] == "#endif"


def test_generated_chain_loops_match_their_generator(tmp_path):
    """vap_chain_asm.h (the lane-per-path kernel's chain loops, inline assembly) is written by tools/gen_chain_asm.py:
    the committed header must be exactly what the generator writes, so that the two cannot drift apart."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "vap_chain_asm.h"
    subprocess.run([sys.executable, os.path.join(root, "tools", "gen_chain_asm.py"), str(out)], check=True, capture_output=True)
    committed = open(os.path.join(root, "vexautonomousplanner_amd", "csrc", "vap_chain_asm.h")).read()
    assert out.read_text() == committed
