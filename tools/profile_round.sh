#!/bin/bash
# Developer tool (GPU box): the evidence bench.py's roofline entry is checked against.
#   tools/profile_round.sh <tag>      e.g. r01
# 1. rocprofv3 --kernel-trace --stats of the default bench command  -> profiles/<tag>_kernel_stats.csv
# 2. two separate PMC passes (FETCH_SIZE, WRITE_SIZE cannot share a pass on gfx950) -> profiles/<tag>_traffic.json
set -e
tag=${1:-r01}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/prof_$tag
mkdir -p $out $root/profiles $root/gpurun_out/profiles
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $out/trace.log 2>&1
cp $out/trace/*/*kernel_stats.csv $root/profiles/${tag}_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -- python $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -- python $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/write.log 2>&1
python - <<PY
import csv, glob, json, collections
def per_kernel(d, name):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name and "vap::" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}
fetch = per_kernel("$out/fetch", "FETCH_SIZE")
write = per_kernel("$out/write", "WRITE_SIZE")
res = {"unit": "bytes per launch", "note": "FETCH_SIZE/WRITE_SIZE are reported in KiB; per MI355X_MICROARCH.md (HBM section) "
       "FETCH_SIZE counts half the bytes of a wide coalesced read on gfx950 and is doubled here; WRITE_SIZE is exact for "
       "16-byte-per-lane stores", "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    fr = fetch.get(k, 0.0) * 1024
    wr = write.get(k, 0.0) * 1024
    res["kernels"][k] = {"fetch_raw": fr, "fetch_corrected": 2 * fr, "write": wr, "hbm_bytes": 2 * fr + wr}
json.dump(res, open("$root/profiles/${tag}_traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
# gpurun merges only gpurun_out/ back: leave copies there for the caller to move into profiles/
cp $root/profiles/${tag}_kernel_stats.csv $root/profiles/${tag}_traffic.json $root/gpurun_out/profiles/
