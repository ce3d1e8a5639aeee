import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from test_gpu_parity import make_gen, run_gpu
from vexautonomousplanner_amd.synth import make_waypoints
B, W, S, seed = [int(v) for v in sys.argv[1:5]] if len(sys.argv) > 4 else (16, 32, 10000, 3)
wp = make_waypoints(B, W, seed).astype(np.float64)
f = run_gpu(torch, make_gen("f32", velocity_kernel="lanes"), wp, samples=S)
s = run_gpu(torch, make_gen("f32", velocity_kernel="lanes", fused_sampling=False), wp, samples=S)
for k in ("x", "y", "heading", "curvature", "velocity"):
    d = f[k] != s[k]
    print(k, "mismatches", int(d.sum()), "of", d.size)
    if d.any():
        idx = np.argwhere(d)
        print("  first:", idx[:8].tolist())
        b, j = idx[0]
        print("  values fused", f[k][b, max(0, j - 2):j + 3], "staged", s[k][b, max(0, j - 2):j + 3])
        print("  cols histogram (j % 64):", np.bincount(idx[:, 1] % 64, minlength=64).tolist())
