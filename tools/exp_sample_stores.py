#!/usr/bin/env python3
"""Developer experiment (GPU box): K3+K4 time as a function of which output rows it writes (config 3) —
is the sampling kernel bound by its stores?"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vexautonomousplanner_amd import _lib
from vexautonomousplanner_amd.synth import make_waypoints

B, W, S = 4096, 32, 10000
dev = torch.device("cuda:0")
L = _lib.lib()
ctx = _lib.Context(0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
wp = torch.tensor(make_waypoints(B, W, 3), device=dev, dtype=torch.float32)
seg = torch.empty((B, W - 1, 6, 2), dtype=torch.float64, device=dev)
seglen = torch.empty((B, W - 1), dtype=torch.float64, device=dev)
meta = torch.zeros((B, 4), dtype=torch.float64, device=dev)
flags = torch.zeros((B,), dtype=torch.int32, device=dev)
lut = torch.empty((B, _lib.LUT_SAMPLES), dtype=torch.float64, device=dev)
o = {k: torch.empty((B, S), dtype=torch.float32, device=dev) for k in ("x", "y", "heading", "curvature", "dtheta")}
_lib.check(L.vap_fit(ctx.handle, _lib.VAP_F32, B, W, p(wp), None, None, p(seg), p(seglen), p(meta), p(flags)), "fit")
_lib.check(L.vap_build_lut(ctx.handle, B, W, p(seg), p(lut), p(meta), p(flags)), "lut")


def run(keep):
    a = {k: (o[k] if k in keep else None) for k in o}
    def call():
        _lib.check(L.vap_sample(ctx.handle, _lib.VAP_F32, B, W, S, 0.0, p(seg), p(lut), p(meta), p(a["x"]), p(a["y"]), p(a["heading"]),
                                p(a["curvature"]), p(a["dtheta"]), p(flags)), "sample")
    for _ in range(3):
        call()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(10):
        call()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / 10


for keep in (("x", "y", "heading", "curvature", "dtheta"), ("x", "y", "heading", "curvature"), ("curvature", "dtheta"), ("curvature",), ()):
    print(f"rows written {len(keep)} ({', '.join(keep) or 'none'}): vap_sample {run(keep):.3f} ms (includes power/slopes/grid helper launches)", flush=True)
