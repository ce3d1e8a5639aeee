#!/usr/bin/env python3
"""Developer tool: per-kernel, per-wave averages from a rocprofv3 --pmc counter_collection.csv."""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:52]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES":
        cnt[k] += 1
for k in acc:
    if "vap" not in k:
        continue
    n = cnt[k]
    a = acc[k]
    w = a["SQ_WAVES"] / n
    rest = " ".join(f"{c[3:]} {a[c] / n / w:.0f}" for c in sorted(a) if c != "SQ_WAVES")
    print(f"{k:52s} x{n} waves {w:.0f} | per wave: {rest}")
