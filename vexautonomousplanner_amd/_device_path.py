"""Single-path plumbing shared by the drop-in classes: one fitted spline resident on the device
(fp64, so the GUI-facing scalar accessors agree with the reference to rounding), reached only through
the C-ABI of include/vap.h.  torch is used for the device buffers, nothing else."""
import ctypes as C

import numpy as np
import torch

from . import _lib


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class DevicePath:
    """Segments / arc-length table of ONE spline on the GPU + host mirrors of the small arrays."""

    def __init__(self, device=0):
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: vexautonomousplanner_amd has no CPU path")
        self.device = torch.device("cuda", device)
        self.ctx = _lib.default_context(device)
        self._L = _lib.lib()
        self.W = 0
        self.segments = None        # host (G,6,2) fp64
        self.segment_lengths = None  # host (G,)
        self.param_last = None
        self.lut = None             # host (1000,) fp64
        self.total = None
        self._d = {}

    # -- K1 -------------------------------------------------------------------------------------
    def fit(self, points, tan_in=None, tan_out=None):
        pts = np.ascontiguousarray(points, dtype=np.float64)
        W = len(pts)
        dev = self.device
        d = self._d = {
            "wp": torch.tensor(pts[None], dtype=torch.float64, device=dev),
            "seg": torch.empty((1, W - 1, 6, 2), dtype=torch.float64, device=dev),
            "seglen": torch.empty((1, W - 1), dtype=torch.float64, device=dev),
            "meta": torch.zeros((1, 4), dtype=torch.float64, device=dev),
            "flags": torch.zeros((1,), dtype=torch.int32, device=dev),
            "tin": None, "tout": None,
        }
        if tan_in is not None:
            d["tin"] = torch.tensor(np.asarray(tan_in, dtype=np.float64)[None], device=dev)
            d["tout"] = torch.tensor(np.asarray(tan_out, dtype=np.float64)[None], device=dev)
        self.ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        st = self._L.vap_fit(self.ctx.handle, _lib.VAP_F64, 1, W, _ptr(d["wp"]), _ptr(d["tin"]),
                             _ptr(d["tout"]), _ptr(d["seg"]), _ptr(d["seglen"]), _ptr(d["meta"]),
                             _ptr(d["flags"]))
        if st == _lib.VAP_ERR_INVALID:
            return False
        _lib.check(st, "vap_fit")
        self.W = W
        self.segments = d["seg"][0].cpu().numpy()
        self.segment_lengths = d["seglen"][0].cpu().numpy()
        self.param_last = float(d["meta"][0, 0].item())
        self.lut = None
        self.total = None
        return True

    # -- K2 -------------------------------------------------------------------------------------
    def build_lut(self):
        d = self._d
        d["lut"] = torch.empty((1, _lib.LUT_SAMPLES), dtype=torch.float64, device=self.device)
        self.ctx.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(self._L.vap_build_lut(self.ctx.handle, 1, self.W, _ptr(d["seg"]), _ptr(d["lut"]),
                                         _ptr(d["meta"]), _ptr(d["flags"])), "vap_build_lut")
        self.lut = d["lut"][0].cpu().numpy()
        self.total = float(self.lut[-1])

    # -- scalar / vector accessors ---------------------------------------------------------------
    def eval(self, order, ts):
        ts = np.ascontiguousarray(np.atleast_1d(ts), dtype=np.float64)
        out = np.empty((len(ts), 2), dtype=np.float64)
        dp = C.POINTER(C.c_double)
        _lib.check(self._L.vap_eval_host(self.ctx.handle, self.W, self.segments.ctypes.data_as(dp),
                                         self.param_last, int(order), len(ts), ts.ctypes.data_as(dp),
                                         out.ctypes.data_as(dp)), "vap_eval_host")
        return out

    def lookup(self, what, xs):
        xs = np.ascontiguousarray(np.atleast_1d(xs), dtype=np.float64)
        out = np.empty(len(xs), dtype=np.float64)
        dp = C.POINTER(C.c_double)
        _lib.check(self._L.vap_lookup_host(self.ctx.handle, self.W, self.segments.ctypes.data_as(dp),
                                           self.param_last, self.lut.ctypes.data_as(dp), int(what), len(xs),
                                           xs.ctypes.data_as(dp), out.ctypes.data_as(dp)), "vap_lookup_host")
        return out

    # -- K3..K5 ----------------------------------------------------------------------------------
    def forward_backward(self, constraints, dd, start_vel, end_vel, want=("velocity",)):
        """MPG:70-316 on the reference grid for this path; returns dict of host fp64 arrays."""
        d = self._d
        total = self.total
        cap = int(np.ceil(total / dd)) + 8
        dev = self.device
        outs = {k: torch.empty((1, cap), dtype=torch.float64, device=dev)
                for k in ("x", "y", "heading", "curvature", "velocity", "dtheta")}
        c = _lib.make_constraints(constraints)
        self.ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        L = self._L
        _lib.check(L.vap_sample(self.ctx.handle, _lib.VAP_F64, 1, self.W, cap, float(dd), _ptr(d["seg"]),
                                _ptr(d["lut"]), _ptr(d["meta"]), _ptr(outs["x"]), _ptr(outs["y"]),
                                _ptr(outs["heading"]), _ptr(outs["curvature"]), _ptr(outs["dtheta"]),
                                _ptr(d["flags"])), "vap_sample")
        _lib.check(L.vap_velocity_pass(self.ctx.handle, _lib.VAP_F64, 1, cap, C.byref(c), float(start_vel),
                                       float(end_vel), _ptr(d["meta"]), _ptr(outs["curvature"]),
                                       _ptr(outs["dtheta"]), None, _ptr(outs["velocity"]), _ptr(d["flags"])),
                   "vap_velocity_pass")
        n = int(d["meta"][0, 3].item())
        return {k: outs[k][0, :n].cpu().numpy() for k in want}, n
