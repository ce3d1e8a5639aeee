#!/bin/bash
# Developer tool: same-box A/B of builds of libvap.so that differ in compile-time defines of ONE source file.
#   build container:  tools/ab_variants.sh build <source.hip> <name>=<-Dflags> [<name>=<-Dflags> ...]
#                     -> vexautonomousplanner_amd/libvap_<name>.so.ab (and libvap_base.so.ab = the tree as it is)
#   GPU box:          tools/ab_variants.sh run <workloads...> -- <names...>      (alternating, two rounds, bench.py stage times)
# (the variants are scratch: *.so.ab and build_ab_* are git-ignored)
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/vexautonomousplanner_amd/csrc
if [ "$1" = build ]; then
  file=$2; shift 2
  make -j8 -C $src > /dev/null || exit 1
  cp $root/vexautonomousplanner_amd/libvap.so $root/vexautonomousplanner_amd/libvap_base.so.ab
  obj=$(basename $file .hip).o
  for spec in "$@"; do
    name=${spec%%=*}; flags=${spec#*=}
    mkdir -p $src/build_ab_$name
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $flags -c $src/$file -o $src/build_ab_$name/$obj || exit 1
    objs=$(ls $src/build/*.o | grep -v "/$obj")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/vexautonomousplanner_amd/libvap_$name.so.ab $objs $src/build_ab_$name/$obj || exit 1
  done
  exit 0
fi
shift
wl=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do wl+=("$1"); shift; done
shift
cp $root/vexautonomousplanner_amd/libvap.so /tmp/libvap_keep.so
for rep in 1 2; do
  for v in "$@"; do
    cp $root/vexautonomousplanner_amd/libvap_$v.so.ab $root/vexautonomousplanner_amd/libvap.so
    for w in "${wl[@]}"; do
      python3 $root/bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-other-mode --no-dropin-c1 --parity-paths 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['pipeline']['stage_ms']
print('$v rep=$rep $w step %.4f lut %.4f sample %.4f velocity %.4f' % (d['ms_per_step'], s['lut'], s['sample'], s['velocity']))"
    done
  done
done
cp /tmp/libvap_keep.so $root/vexautonomousplanner_amd/libvap.so
