"""TEST INFRASTRUCTURE ONLY — ctypes binding of the fp64 CPU restatement (oracle/vap_oracle.c).

Allowed importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.  The product
package (vexautonomousplanner_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libvap_oracle.so")
_lib = None

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)
c_long_p = C.POINTER(C.c_long)


class _Nodes(C.Structure):
    _fields_ = [("is_reverse", c_int_p), ("turn", c_double_p), ("stop", c_int_p),
                ("wait_time", c_double_p), ("max_velocity", c_double_p),
                ("max_acceleration", c_double_p), ("tangent", c_double_p),
                ("magnitudes", c_double_p)]


class _Actions(C.Structure):
    _fields_ = [("M", C.c_int), ("t", c_double_p), ("stop", c_int_p), ("wait_time", c_double_p),
                ("max_velocity", c_double_p), ("max_acceleration", c_double_p)]


def build(force=False):
    if force or not os.path.exists(LIB_PATH) or any(
            os.path.getmtime(os.path.join(HERE, f)) > os.path.getmtime(LIB_PATH)
            for f in ("vap_oracle.c", "vap_oracle.h")):
        subprocess.check_call(["make", "-s", "-C", HERE, "libvap_oracle.so"])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.vapo_path_create.restype = C.c_void_p
        L.vapo_path_create.argtypes = [C.c_int, c_double_p, C.POINTER(_Nodes), C.POINTER(_Actions)]
        L.vapo_path_destroy.argtypes = [C.c_void_p]
        L.vapo_basis.argtypes = [C.c_int, C.c_double, c_double_p]
        L.vapo_n_splines.argtypes = [C.c_void_p]
        L.vapo_n_segments.argtypes = [C.c_void_p]
        L.vapo_get_segments.argtypes = [C.c_void_p, c_double_p, c_double_p, c_double_p]
        for f in ("vapo_point", "vapo_derivative", "vapo_second_derivative"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_double, c_double_p]
        L.vapo_rebuild_tables.argtypes = [C.c_void_p]
        L.vapo_build_tables_sized.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.vapo_build_tables_sized.restype = C.c_int
        L.vapo_lut_size.argtypes = [C.c_void_p]
        L.vapo_get_lut.argtypes = [C.c_void_p, c_double_p, c_double_p, c_double_p]
        L.vapo_table_size.argtypes = [C.c_void_p]
        L.vapo_get_table.argtypes = [C.c_void_p, c_double_p, c_double_p, c_double_p]
        L.vapo_total_arc_length.restype = C.c_double
        L.vapo_total_arc_length.argtypes = [C.c_void_p]
        for f in ("vapo_distance_to_time", "vapo_curvature", "vapo_heading"):
            getattr(L, f).restype = C.c_double
            getattr(L, f).argtypes = [C.c_void_p, C.c_double]
        L.vapo_count_samples.restype = C.c_long
        L.vapo_count_samples.argtypes = [C.c_void_p, C.c_double]
        L.vapo_dd_for_samples.restype = C.c_double
        L.vapo_dd_for_samples.argtypes = [C.c_void_p, C.c_long]
        L.vapo_forward_backward.restype = C.c_long
        L.vapo_forward_backward.argtypes = [C.c_void_p, c_double_p, C.c_double, C.c_double,
                                            C.c_double, C.c_long] + [c_double_p] * 6
        L.vapo_profile_batch.restype = C.c_int
        L.vapo_profile_batch.argtypes = [C.c_int, C.c_int, C.c_long, c_double_p, c_double_p,
                                         C.c_double, C.c_double] + [c_double_p] * 6 + [C.c_int]
        L.vapo_generate_motion_profile.restype = C.c_long
        L.vapo_generate_motion_profile.argtypes = [C.c_void_p, c_double_p, C.c_double, C.c_double,
                                                   C.c_long, c_double_p, c_long_p, c_int_p,
                                                   c_long_p, c_int_p]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(c_double_p) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(c_int_p) if a is not None else None


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a) != 0, dtype=np.int32)


class OraclePath:
    """One path of the CPU restatement.  Mirrors the calls the reference's L2 makes on its manager."""

    def __init__(self, waypoints, nodes=None, actions=None):
        L = lib()
        self._wp = _f64(waypoints)
        self.W = len(self._wp)
        keep = []
        nptr = None
        if nodes:
            n = _Nodes()
            for key, conv, setp in (("is_reverse", _i32, _ip), ("turn", _f64, _dp), ("stop", _i32, _ip),
                                    ("wait_time", _f64, _dp), ("max_velocity", _f64, _dp),
                                    ("max_acceleration", _f64, _dp), ("tangent", _f64, _dp),
                                    ("magnitudes", _f64, _dp)):
                arr = conv(nodes.get(key))
                keep.append(arr)
                setattr(n, key, setp(arr))
            nptr = C.byref(n)
            keep.append(n)
        aptr = None
        self.M = 0
        if actions and len(actions["t"]):
            a = _Actions()
            a.M = self.M = len(actions["t"])
            for key, conv, setp in (("t", _f64, _dp), ("stop", _i32, _ip), ("wait_time", _f64, _dp),
                                    ("max_velocity", _f64, _dp), ("max_acceleration", _f64, _dp)):
                arr = conv(actions.get(key))
                keep.append(arr)
                setattr(a, key, setp(arr))
            aptr = C.byref(a)
            keep.append(a)
        self._h = L.vapo_path_create(self.W, _dp(self._wp), nptr, aptr)
        if not self._h:
            raise ValueError("build_path failed")
        self._L = L

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.vapo_path_destroy(self._h)
            self._h = None

    @property
    def n_splines(self):
        return self._L.vapo_n_splines(self._h)

    def segments(self):
        G = self.W - 1
        seg = np.empty((G, 6, 2))
        sl = np.empty(G)
        pl = np.empty(self.n_splines)
        self._L.vapo_get_segments(self._h, _dp(seg), _dp(sl), _dp(pl))
        return seg, sl, pl

    def _ev(self, fn, t):
        out = np.empty(2)
        fn(self._h, float(t), _dp(out))
        return out

    def point(self, t):
        return self._ev(self._L.vapo_point, t)

    def derivative(self, t):
        return self._ev(self._L.vapo_derivative, t)

    def second_derivative(self, t):
        return self._ev(self._L.vapo_second_derivative, t)

    def rebuild_tables(self):
        self._L.vapo_rebuild_tables(self._h)

    def build_tables_sized(self, min_samples, samples_per_node):
        if self._L.vapo_build_tables_sized(self._h, int(min_samples), int(samples_per_node)) != 0:
            raise IndexError("min_samples < 2 (spline_manager.py:444)")

    def lut(self):
        n = self._L.vapo_lut_size(self._h)
        d = np.empty(n)
        p = np.empty(n)
        tot = C.c_double()
        self._L.vapo_get_lut(self._h, _dp(d), _dp(p), C.byref(tot))
        return d, p, tot.value

    def table(self):
        n = self._L.vapo_table_size(self._h)
        p = np.empty(n)
        k = np.empty(n)
        h = np.empty(n)
        self._L.vapo_get_table(self._h, _dp(p), _dp(k), _dp(h))
        return p, k, h

    def total_arc_length(self):
        return self._L.vapo_total_arc_length(self._h)

    def distance_to_time(self, s):
        return self._L.vapo_distance_to_time(self._h, float(s))

    def curvature(self, t):
        return self._L.vapo_curvature(self._h, float(t))

    def heading(self, t):
        return self._L.vapo_heading(self._h, float(t))

    def dd_for_samples(self, S):
        return self._L.vapo_dd_for_samples(self._h, int(S))

    def forward_backward(self, constraints, dd, start_vel=0.01, end_vel=0.01):
        c = _f64(constraints)
        N = self._L.vapo_count_samples(self._h, float(dd))
        outs = [np.empty(N) for _ in range(6)]
        n = self._L.vapo_forward_backward(self._h, _dp(c), float(dd), start_vel, end_vel, N,
                                          *[_dp(o) for o in outs])
        assert n == N
        return dict(zip(("t", "x", "y", "heading", "curvature", "velocity"), outs))

    def generate_motion_profile(self, constraints, dt=0.01, dd=0.005, cap=200000):
        c = _f64(constraints)
        out = np.empty((cap, 8))
        nmap = np.zeros(self.W + 1, dtype=np.int64)
        amap = np.zeros(self.M + 1, dtype=np.int64)
        nn = C.c_int()
        na = C.c_int()
        T = self._L.vapo_generate_motion_profile(self._h, _dp(c), dt, dd, cap, _dp(out),
                                                 nmap.ctypes.data_as(c_long_p), C.byref(nn),
                                                 amap.ctypes.data_as(c_long_p), C.byref(na))
        if T < 0:
            raise ValueError(f"generate_motion_profile failed ({T})")
        return out[:T].copy(), nmap[:nn.value].copy(), amap[:na.value].copy()


def basis(order, ts):
    """QHS:288-469: rows [H0..H5] of the basis of the given derivative order (0..3) at local parameters ts."""
    ts = np.atleast_1d(np.asarray(ts, dtype=np.float64))
    out = np.empty((len(ts), 6), dtype=np.float64)
    L = lib()
    for i, t in enumerate(ts):
        L.vapo_basis(int(order), float(t), _dp(out[i]))
    return out


def profile_batch(waypoints, S, constraints, start_vel=0.01, end_vel=0.01, n_threads=1,
                  want=("x", "y", "heading", "curvature", "velocity")):
    """Whole hot path for a (B,W,2) batch of plain-node paths on the fixed-sample grid."""
    wp = _f64(waypoints)
    B, W, _ = wp.shape
    c = _f64(constraints)
    outs = {k: (np.empty((B, S)) if k in want else None)
            for k in ("x", "y", "heading", "curvature", "velocity")}
    total = np.empty(B)
    err = lib().vapo_profile_batch(B, W, S, _dp(wp), _dp(c), start_vel, end_vel,
                                   _dp(outs["x"]), _dp(outs["y"]), _dp(outs["heading"]),
                                   _dp(outs["curvature"]), _dp(outs["velocity"]), _dp(total),
                                   n_threads)
    if err:
        raise ValueError(f"path {err - 1} failed")
    outs = {k: v for k, v in outs.items() if v is not None}
    outs["total_length"] = total
    return outs
