#!/usr/bin/env python3
"""Developer tool: turn the raw output of tools/profile_round.sh (gpurun_out/prof_<tag>/) into the committed
summaries profiles/<tag>_{kernel_stats.csv,traffic.json,bench.json,sq_counters.txt}.
   python tools/profile_summary.py <tag>"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

tag = sys.argv[1]
out = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
prof = os.path.join(ROOT, "profiles")
os.makedirs(prof, exist_ok=True)
shutil.copy(glob.glob(out + "/trace/*/*kernel_stats.csv")[0], os.path.join(prof, f"{tag}_kernel_stats.csv"))


def per_kernel(d, name):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name and "vap::" in r["Kernel_Name"]:
            kname = r["Kernel_Name"].replace("(anonymous namespace)::", "")
            acc[kname.split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch = per_kernel(out + "/fetch", "FETCH_SIZE")
write = per_kernel(out + "/write", "WRITE_SIZE")
# the bench line of the FETCH_SIZE pass says which kernel sources the counters were taken on (hashed while that process ran,
# not when this summary is written); the line that is committed as <tag>_bench.json is taken after the PMC passes, so it
# carries the traffic figure of this very profile
pmc_line = [json.loads(l) for l in open(out + "/fetch.log") if l.startswith('{"metric"')][-1]
line = [json.loads(l) for l in open(out + "/trace.log") if l.startswith('{"metric"')][-1]
cfg = line["config"]
res = {"unit": "bytes per launch",
       "note": "FETCH_SIZE/WRITE_SIZE are reported in KiB; per MI355X_MICROARCH.md (HBM section) FETCH_SIZE counts half the bytes "
               "of a wide coalesced read on gfx950 and is doubled here; WRITE_SIZE is exact for 16-byte-per-lane stores",
       "kernel_source_sha": pmc_line["kernel_source_sha"],
       "bench": {"workload": cfg["workload"].split(":")[0], "dtype": line["dtype"], "paths": cfg["paths_per_gpu"],
                 "recurrence": "f64" if "f64" in cfg["recurrence"] else "f32"},
       "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    fr = fetch.get(k, 0.0) * 1024
    wr = write.get(k, 0.0) * 1024
    res["kernels"][k] = {"fetch_raw": fr, "fetch_corrected": 2 * fr, "write": wr, "hbm_bytes": 2 * fr + wr}
json.dump(res, open(os.path.join(prof, f"{tag}_traffic.json"), "w"), indent=1)
final = out + "/bench_final.log"
if os.path.exists(final):
    line = [json.loads(l) for l in open(final) if l.startswith('{"metric"')][-1]
json.dump(line, open(os.path.join(prof, f"{tag}_bench.json"), "w"), indent=1)
if glob.glob(out + "/sq/*/*counter_collection.csv"):
    txt = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), out + "/sq"], capture_output=True, text=True).stdout
    open(os.path.join(prof, f"{tag}_sq_counters.txt"), "w").write(txt)
print(json.dumps(res, indent=1))
