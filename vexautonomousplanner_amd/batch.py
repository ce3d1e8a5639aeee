"""Batched trajectory generation on one MI355X: the new batch-of-paths axis (north_star).

Device memory, streams and the multi-GPU launcher come from PyTorch-ROCm (plumbing only); all
numerics run in the HIP kernels of libvap.so through the C-ABI of include/vap.h.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .synth import DEFAULT_CONSTRAINTS, END_VEL, START_VEL

FIELDS = ("x", "y", "heading", "curvature", "velocity")


def _torch_dtype(dtype):
    if dtype in ("f32", "fp32", torch.float32, np.float32, _lib.VAP_F32):
        return torch.float32, _lib.VAP_F32
    if dtype in ("f64", "fp64", torch.float64, np.float64, _lib.VAP_F64):
        return torch.float64, _lib.VAP_F64
    raise ValueError(f"unsupported dtype {dtype!r}")


class BatchedTrajectoryGenerator:
    """rebuild_tables + forward_backward_pass (SM:582-594, MPG:70-316) for B independent
    plain-node paths per call, outputs resident in HBM as (B, S) tensors."""

    def __init__(self, device=0, dtype="f32", timing=False, velocity_kernel="auto"):
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: vexautonomousplanner_amd has no CPU path")
        self.device = torch.device("cuda", device)
        self.tdtype, self.vdtype = _torch_dtype(dtype)
        self.ctx = _lib.Context(device)
        if timing:
            self.ctx.set_timing(True)
        self._L = _lib.lib()
        self.set_velocity_kernel(velocity_kernel)

    def set_velocity_kernel(self, which):
        """"auto" | "seq_literal" | "seq_fast" | "relax" | "relax_block" | "relax_wave" (VAP_OPT_VELOCITY_KERNEL)."""
        table = {"auto": _lib.VELOCITY_AUTO, "seq_literal": _lib.VELOCITY_SEQ_LITERAL,
                 "seq_fast": _lib.VELOCITY_SEQ_FAST, "relax": _lib.VELOCITY_RELAX,
                 "relax_block": _lib.VELOCITY_RELAX_BLOCK, "relax_wave": _lib.VELOCITY_RELAX_WAVE}
        self.ctx.set_option(_lib.OPT_VELOCITY_KERNEL, table[which])

    def profile(self, waypoints, constraints=DEFAULT_CONSTRAINTS, samples=None, dd=None,
                start_vel=START_VEL, end_vel=END_VEL, want=FIELDS, out=None, capacity=None):
        """waypoints: (B, W, 2) tensor on this device, dtype matching the generator.

        Exactly one of:
          samples=S : fixed grid of S samples per path, dd_b = L_b/(S-1.5)   (bench semantics)
          dd=0.005  : the reference's grid (MPG:112-122); ``capacity`` = row length of the
                      outputs (default: enough for a 64 ft path), n_samples per path in meta[:,3]
        Returns a dict of (B, S) tensors for the requested fields plus
          meta  (B,4) fp64: parameters[-1], total_length, dd, n_samples
          flags (B,)  int32 bit-mask (VAP_FLAG_*)
        """
        if (samples is None) == (dd is None):
            raise ValueError("give exactly one of samples= or dd=")
        wp = waypoints
        if wp.device != self.device or wp.dtype != self.tdtype or wp.dim() != 3 or wp.shape[2] != 2:
            raise ValueError(f"waypoints must be a (B,W,2) {self.tdtype} tensor on {self.device}")
        wp = wp.contiguous()
        B, W, _ = wp.shape
        if samples is not None:
            S, ddv = int(samples), 0.0
        else:
            ddv = float(dd)
            S = int(capacity) if capacity is not None else int(np.ceil(64.0 / ddv)) + 2
        c = _lib.make_constraints(constraints)
        res = {} if out is None else out
        for f in FIELDS:
            if f in want or f == "velocity":
                t = res.get(f)
                if t is None or t.shape != (B, S) or t.dtype != self.tdtype:
                    res[f] = torch.empty((B, S), dtype=self.tdtype, device=self.device)
        if "meta" not in res or res["meta"].shape != (B, 4):
            res["meta"] = torch.empty((B, 4), dtype=torch.float64, device=self.device)
        if "flags" not in res or res["flags"].shape != (B,):
            res["flags"] = torch.empty((B,), dtype=torch.int32, device=self.device)

        def p(name):
            t = res.get(name) if (name in want or name == "velocity") else None
            return C.c_void_p(t.data_ptr()) if t is not None else None

        self.ctx.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        st = self._L.vap_profile_batch(self.ctx.handle, self.vdtype, B, W, S, ddv,
                                       C.c_void_p(wp.data_ptr()), C.byref(c), float(start_vel),
                                       float(end_vel), p("x"), p("y"), p("heading"),
                                       p("curvature"), p("velocity"),
                                       C.c_void_p(res["meta"].data_ptr()),
                                       C.c_void_p(res["flags"].data_ptr()))
        _lib.check(st, "vap_profile_batch")
        self._last_shape = (B, W, S)
        return res

    def time_profile(self, result, constraints=DEFAULT_CONSTRAINTS, dt=0.01, capacity_rows=None, out=None):
        """Time-domain resample (the loop of generate_motion_profile, MPG:413-628) of the batch that
        ``profile`` has just produced with this generator: ``result`` is its return value (the
        velocity rows and meta are read from it, the spline tables from the context).

        Returns a dict with
          rows      (B, capacity_rows, 8) fp64: time, position, velocity, acceleration, heading,
                    angular velocity, x, y per time step of ``dt`` seconds (MPG:558-592)
          counts    (B, 2) int32: rows written, entries of nodes_map
          nodes_map (B, W) int32: row index at which each node is passed (MPG:420, 527-529)
        Paths that need more than capacity_rows rows are cut there and flagged (result["flags"]).
        """
        vel, meta = result["velocity"], result["meta"]
        B, S = vel.shape
        last = getattr(self, "_last_shape", None)
        if last is None or (last[0], last[2]) != (B, S):
            raise ValueError("time_profile needs the result of this generator's last profile() call")
        W = last[1]
        if capacity_rows is None:
            capacity_rows = 4096
        res = {} if out is None else out
        if "rows" not in res or res["rows"].shape != (B, capacity_rows, 8):
            res["rows"] = torch.empty((B, capacity_rows, 8), dtype=torch.float64, device=self.device)
        if "counts" not in res or res["counts"].shape != (B, 2):
            res["counts"] = torch.empty((B, 2), dtype=torch.int32, device=self.device)
        if "nodes_map" not in res or res["nodes_map"].shape != (B, W):
            res["nodes_map"] = torch.empty((B, W), dtype=torch.int32, device=self.device)
        c = _lib.make_constraints(constraints)
        self.ctx.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        st = self._L.vap_time_profile(self.ctx.handle, self.vdtype, B, W, S, None, None,
                                      C.c_void_p(meta.data_ptr()), C.c_void_p(vel.data_ptr()), C.byref(c),
                                      float(dt), int(capacity_rows), C.c_void_p(res["rows"].data_ptr()),
                                      C.c_void_p(res["counts"].data_ptr()), C.c_void_p(res["nodes_map"].data_ptr()),
                                      C.c_void_p(result["flags"].data_ptr()))
        _lib.check(st, "vap_time_profile")
        return res

    def timing(self):
        return self.ctx.last_timing()
