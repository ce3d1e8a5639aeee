#!/bin/bash
# Developer tool (GPU box): one rocprofv3 PMC pass over bench.py, summarised per wave.
#   tools/pmc.sh <out-name> <counters...> -- <bench args>
name=$1; shift
ctrs=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do ctrs+=("$1"); shift; done
shift
out=$GRAFT_REPO_ROOT/gpurun_out/$name
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu-baseline > $out.log 2>&1
python $GRAFT_REPO_ROOT/tools/pmc_summary.py $out
