#!/usr/bin/env python3
"""Developer tool (GPU box): cost of node limits on a config-3 batch — vap_route_limits + the velocity pass with
per-sample initial velocities (relaxation kernel), against the plain velocity stage."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vexautonomousplanner_amd.batch import BatchedTrajectoryGenerator
from vexautonomousplanner_amd.synth import make_waypoints, DEFAULT_CONSTRAINTS

B, W, S = 4096, 32, 10000
gen = BatchedTrajectoryGenerator(0, "f32", timing=True)
wp = torch.tensor(make_waypoints(B, W, 3), device="cuda:0", dtype=torch.float32)
rng = np.random.default_rng(1)
mv = np.where(rng.random((B, W)) < 0.3, rng.uniform(1.5, 3.5, (B, W)), 0.0)
stop = rng.random((B, W)) < 0.1
stop[:, 0] = stop[:, -1] = False
r = gen.profile(wp, DEFAULT_CONSTRAINTS, samples=S)
torch.cuda.synchronize()
print("plain stages (ms):", {k: round(v, 4) for k, v in gen.timing().items()})
for _ in range(2):
    gen.apply_node_limits(r, DEFAULT_CONSTRAINTS, node_max_velocity=mv, node_stop=stop)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
torch.cuda.synchronize()
ev[0].record()
for _ in range(5):
    gen.apply_node_limits(r, DEFAULT_CONSTRAINTS, node_max_velocity=mv, node_stop=stop)
ev[1].record()
torch.cuda.synchronize()
print(f"apply_node_limits (host event packing + vap_route_limits + velocity pass with limits): {ev[0].elapsed_time(ev[1]) / 5:.3f} ms per call")
# device part alone: the two C calls on prepared event tensors
import ctypes as C
from vexautonomousplanner_amd import _lib
L = _lib.lib()
d_mv = torch.tensor(mv, device="cuda:0")
d_stop = torch.tensor(stop.astype(np.int32), device="cuda:0")
vcap = torch.empty((B, S), dtype=torch.float64, device="cuda:0")   # limit rows: fp64 in the default mode (vap_limit_rows_dtype)
c = _lib.make_constraints(DEFAULT_CONSTRAINTS)
p = lambda t: C.c_void_p(t.data_ptr())
def dev_calls():
    _lib.check(L.vap_route_limits(gen.ctx.handle, _lib.VAP_F32, B, W, 0, S, None, p(r["meta"]), p(d_mv), None, p(d_stop), None, None, None, None,
                                  C.byref(c), 0.01, p(vcap), None, None, None, None, None), "limits")
def vel_call():
    _lib.check(L.vap_velocity_pass(gen.ctx.handle, _lib.VAP_F32, B, S, C.byref(c), 0.01, 0.01, p(r["meta"]), p(r["curvature"]), None, p(vcap),
                                   p(r["velocity"]), p(r["flags"])), "vel")
for name, fn in (("vap_route_limits", dev_calls), ("vap_velocity_pass with limits", vel_call)):
    fn(); torch.cuda.synchronize()
    ev[0].record()
    for _ in range(10):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    print(f"{name}: {ev[0].elapsed_time(ev[1]) / 10:.3f} ms")
