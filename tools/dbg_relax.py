import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from vexautonomousplanner_amd.batch import BatchedTrajectoryGenerator
from vexautonomousplanner_amd.synth import make_waypoints
wp = torch.tensor(make_waypoints(8, 32, 3), device="cuda:0")
outs = {}
for which in ("relax", "seq_fast"):
    gen = BatchedTrajectoryGenerator(0, "f32", velocity_kernel=which)
    r = gen.profile(wp, samples=10000)
    torch.cuda.synchronize()
    outs[which] = r["velocity"].cpu().numpy()
d = outs["relax"] != outs["seq_fast"]
print("mismatches per path", d.sum(axis=1))
for b in range(8):
    idx = np.nonzero(d[b])[0]
    if len(idx):
        print(b, "first", idx[:5], "last", idx[-5:], "max rel", np.max(np.abs(outs["relax"][b][idx]-outs["seq_fast"][b][idx])/outs["seq_fast"][b][idx]))
