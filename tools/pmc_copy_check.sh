#!/bin/bash
# Developer tool (GPU box): FETCH_SIZE / WRITE_SIZE of known-size kernels -> gpurun_out/pmc_copy_check.txt
set -e
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/pmc_copy
mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $out/pmc_copy_check $root/tools/pmc_copy_check.hip
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -- $out/pmc_copy_check > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -- $out/pmc_copy_check > $out/write.log 2>&1
python3 - <<PY > $root/gpurun_out/pmc_copy_check.txt
import csv, glob, collections
res = collections.defaultdict(lambda: collections.defaultdict(list))
for kind in ("fetch", "write"):
    for f in glob.glob("$out/%s/*/*counter_collection.csv" % kind):
        for row in csv.DictReader(open(f)):
            res[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
GiB = float(1 << 30)
print("kernel, FETCH_SIZE KiB -> bytes / actual read bytes, WRITE_SIZE KiB -> bytes / actual written bytes (1 GiB each; read16 writes nothing)")
for k, v in sorted(res.items()):
    f = sum(v.get("FETCH_SIZE", [0])) / max(len(v.get("FETCH_SIZE", [1])), 1) * 1024
    w = sum(v.get("WRITE_SIZE", [0])) / max(len(v.get("WRITE_SIZE", [1])), 1) * 1024
    print(f"{k:40s} fetch {f / GiB:.3f} x   write {w / GiB:.3f} x")
PY
cat $root/gpurun_out/pmc_copy_check.txt
