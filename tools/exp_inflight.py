#!/usr/bin/env python3
"""Experiment: config 3 with n batches in flight — n generators (one context, one HIP stream, one set of rows each)
taking the steps in turn — against one.  The velocity kernel is bound by one wave per CU (the chain) and leaves vector
issue slots idle; the sampling kernel of the next batch can take them.

    python tools/exp_inflight.py [n ...]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from vexautonomousplanner_amd.batch import BatchedTrajectoryGenerator
from vexautonomousplanner_amd.synth import DEFAULT_CONSTRAINTS, make_waypoints

B, W, S, K = 4096, 32, 10000, 40
dev = torch.device("cuda", 0)
wp = torch.tensor(make_waypoints(B, W, 3, dtype=np.float32), device=dev)
for n in [int(a) for a in sys.argv[1:]] or [1, 2, 3]:
    gens = [BatchedTrajectoryGenerator(0, "f32") for _ in range(n)]
    streams = [torch.cuda.Stream(dev) for _ in range(n)]
    outs = [None] * n

    def run(k):
        for i in range(k):
            q = i % n
            with torch.cuda.stream(streams[q]):
                outs[q] = gens[q].profile(wp, constraints=DEFAULT_CONSTRAINTS, samples=S, out=outs[q])

    run(60)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    run(K)
    torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t0) * 1e3 / K
    print(f"in flight {n}: {ms:.4f} ms per step, {B * S / ms * 1e3:.4g} points/s", flush=True)
    del gens, outs
