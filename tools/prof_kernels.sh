#!/bin/bash
# Developer tool (GPU box): rocprofv3 kernel stats of one bench command -> gpurun_out/<tag>_kernel_stats.csv
#   tools/prof_kernels.sh <tag> [bench args...]
set -e
tag=$1
shift
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-mode --parity-paths 0 "$@" > $out/trace.log 2>&1 < /dev/null
cp $out/trace/*/*kernel_stats.csv $root/gpurun_out/${tag}_kernel_stats.csv
cut -d, -f1-4 $root/gpurun_out/${tag}_kernel_stats.csv | cut -c1-160
