#!/usr/bin/env python3
"""Developer tool (GPU box): the large-W sweep's worst case (tools/fuzz_parity.py big, case 115) against the REAL
reference's velocity rows (fixtures big_w2048_p*, oracle/gen_golden.py --big), per path, row type and velocity kernel:
which form of the recurrence the reference's own amplification acts on (FAST form or the sqrt)."""
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import oracle
from vexautonomousplanner_amd.batch import BatchedTrajectoryGenerator

files = sorted(glob.glob("tests/golden/big_w2048_p*.npz") + glob.glob("tests/golden_tmp/big_w2048_p*.npz"))
for f in files:
    g = np.load(f)
    N = int(g["n_samples"])
    ref = g["velocity_full"]
    op = oracle.OraclePath(g["waypoints"])
    op.rebuild_tables()
    vo = op.forward_backward(tuple(g["constraints"]), dd=float(g["dd"]))["velocity"]
    line = [os.path.basename(f), f"oracle {np.max(np.abs(vo - ref) / ref):.2e}"]
    for dtype in ("f32", "f64"):
        for kern in ("auto", "seq_fast", "seq_literal"):
            gen = BatchedTrajectoryGenerator(0, dtype, velocity_kernel=kern)
            wp = torch.tensor(g["waypoints"][None], dtype=gen.tdtype, device=gen.device)
            r = gen.profile(wp, tuple(g["constraints"]), dd=float(g["dd"]), capacity=N + 2)
            torch.cuda.synchronize()
            v = r["velocity"][0, :N].double().cpu().numpy()
            e = np.abs(v - ref) / ref
            eo = np.abs(v - vo) / vo
            line.append(f"{dtype}/{kern} ref {e.max():.2e} (oracle {eo.max():.2e})")
    print(" | ".join(line), flush=True)
