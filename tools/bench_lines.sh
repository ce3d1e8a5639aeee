#!/bin/bash
# Developer tool (GPU box): one bench line each for the workloads and modes that are not the default command, as
#   gpurun_out/profiles/<tag>_bench_<name>.json   (copy into profiles/ to commit)
#   tools/bench_lines.sh r03
tag=${1:-r03}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/profiles
mkdir -p $out
line() {   # name, bench args...
    name=$1; shift
    python3 $root/bench.py "$@" > $root/gpurun_out/bench_line_$name.log 2>&1
    grep '^{"metric"' $root/gpurun_out/bench_line_$name.log | tail -1 > $out/${tag}_bench_$name.json
    echo "$name: $(python3 -c "import json,sys; d=json.load(open('$out/${tag}_bench_$name.json')); print(round(d['ms_per_step'],4), 'ms', '%.3g' % d['value'], 'points/s', d['pipeline']['stage_ms'])")"
}
line c2 --workload c2 --steps 20 --warmup 3 --no-cpu-baseline --no-other-mode
line c4 --workload c4 --steps 20 --warmup 3 --no-cpu-baseline --no-other-mode
line c5 --workload c5 --steps 10 --warmup 2 --no-cpu-baseline --no-other-mode --tolerance-sweep
line f64 --dtype f64 --steps 20 --warmup 3 --no-cpu-baseline
line f32rec --recurrence f32 --steps 20 --warmup 3 --no-cpu-baseline --no-other-mode
line timedomain --steps 10 --warmup 2 --no-cpu-baseline --no-other-mode --time-domain
