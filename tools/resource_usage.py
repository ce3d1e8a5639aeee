#!/usr/bin/env python3
"""Developer tool: per-kernel register / LDS / occupancy table from hipcc's -Rpass-analysis=kernel-resource-usage.
   python tools/resource_usage.py [file.hip ...] [--grep PATTERN]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vexautonomousplanner_amd", "csrc")
args = [a for a in sys.argv[1:] if not a.startswith("--")]
pat = None
if "--grep" in sys.argv:
    pat = sys.argv[sys.argv.index("--grep") + 1]
    args = [a for a in args if a != pat]
files = args or ["vap_kernels.hip"]
for f in files:
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
           "-Rpass-analysis=kernel-resource-usage", "-c", f, "-o", "/dev/null"]
    out = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr
    cur = {}
    rows = []
    for line in out.splitlines():
        m = re.search(r"remark:\s+(.+?): (\S+) \[-Rpass", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2).strip()
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        else:
            cur[k] = v
    for r in rows:
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name)
        if pat and not re.search(pat, name):
            continue
        print(f"{name:90s} vgpr {r.get('VGPRs','?'):>4} agpr {r.get('AGPRs','?'):>3} spill {r.get('VGPRs Spill','?'):>3} "
              f"scratch {r.get('ScratchSize [bytes/lane]','?'):>4} occ {r.get('Occupancy [waves/SIMD]','?'):>2} lds {r.get('LDS Size [bytes/block]','?')}")
