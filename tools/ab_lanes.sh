#!/bin/bash
# Developer tool (GPU box): same-box A/B of two builds of libvap.so (vexautonomousplanner_amd/libvap8.so.ab, libvap12.so.ab),
# alternating, for the workloads given:  tools/ab_lanes.sh c3 c4 c5
root=$GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for v in 8 12; do
    cp $root/vexautonomousplanner_amd/libvap$v.so.ab $root/vexautonomousplanner_amd/libvap.so
    for w in "$@"; do
      python3 $root/bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-other-mode --parity-paths 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['pipeline']['stage_ms']
print('waves=$v rep=$rep $w step %.4f sample %.4f velocity %.4f' % (d['ms_per_step'], s['sample'], s['velocity']))"
    done
  done
done
