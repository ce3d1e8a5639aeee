import sys, os, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import golden_util as gu
from vexautonomousplanner_amd.nodes import Node
from vexautonomousplanner_amd.splines.spline_manager import QuinticHermiteSplineManager
g = gu.load("c1_w8")
m = QuinticHermiteSplineManager()
print(m.build_path(g["waypoints"], [Node() for _ in g["waypoints"]], []))
dev = m._dev()
print("meta", dev._d["meta"].cpu().numpy(), "flags", dev._d["flags"].cpu().numpy())
dev.build_lut()
print("lut", dev.lut[:5], dev.lut[-3:], "meta", dev._d["meta"].cpu().numpy())
