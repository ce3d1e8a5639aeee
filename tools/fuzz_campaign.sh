#!/bin/bash
# Developer tool (GPU box): every randomised checker for a fixed time each; the last line of each goes to
#   gpurun_out/profiles/<tag>_fuzz.txt    (copy into profiles/ to commit)      tools/fuzz_campaign.sh r04 [seconds each]
tag=${1:-r04}
secs=${2:-60}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/profiles
mkdir -p $out
: > $out/${tag}_fuzz.txt
run() {   # name, command...
    name=$1; shift
    "$@" > $root/gpurun_out/fuzz_$name.log 2>&1
    rc=$?
    echo "$name (rc $rc): $(grep -v amdgpu.ids $root/gpurun_out/fuzz_$name.log | tail -n 1)" | tee -a $out/${tag}_fuzz.txt
}
run lanes            python3 $root/tools/fuzz_lanes.py $secs
run parity           python3 $root/tools/fuzz_parity.py $secs
run long_rows        python3 $root/tools/fuzz_long_rows.py $secs
run time_profile     python3 $root/tools/fuzz_time_profile.py $secs
run time_quad_vs_lane python3 $root/tools/ab_time_quad.py 60
run batch_routes     python3 $root/tools/fuzz_batch_routes.py $secs
run batch_split_routes python3 $root/tools/fuzz_batch_split_routes.py $secs
run routes_batch_kernels python3 $root/tools/fuzz_routes.py $secs 21 --batch-kernels
run routes_one_lane  python3 $root/tools/fuzz_routes.py $secs 21
