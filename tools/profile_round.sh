#!/bin/bash
# Developer tool (GPU box): the evidence bench.py's roofline entry is checked against.
#   tools/profile_round.sh <tag> [bench args...]      e.g.  tools/profile_round.sh r02
# 1. rocprofv3 --kernel-trace --stats of the default bench command  -> profiles/<tag>_kernel_stats.csv
# 2. two separate PMC passes (FETCH_SIZE, WRITE_SIZE cannot share a pass on gfx950) -> profiles/<tag>_traffic.json,
#    which records the bench mode and the hash of the kernel sources it was taken on (bench.py refuses a stale one)
# 3. one SQ pass (instructions / busy cycles per wave) -> profiles/<tag>_sq_counters.txt
# 4. the bench line itself, taken last -> profiles/<tag>_bench.json (its traffic entries quote the passes above)
set -e
tag=${1:-r02}
shift || true
extra="$@"
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/prof_$tag
mkdir -p $out $root/profiles $root/gpurun_out/profiles
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-mode --no-dropin-c1 --parity-paths 0 $extra > $out/trace.log 2>&1
cp $out/trace/*/*kernel_stats.csv $root/profiles/${tag}_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-mode --no-dropin-c1 --parity-paths 0 $extra > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-mode --no-dropin-c1 --parity-paths 0 $extra > $out/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $out/sq -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-mode --no-dropin-c1 --parity-paths 0 $extra > $out/sq.log 2>&1 || true
python3 $root/tools/profile_summary.py $tag
# the bench line to commit: after the PMC passes, so that its roofline.traffic quotes this profile (cpu_baseline included)
python3 $root/bench.py --steps 20 --warmup 3 $extra > $out/bench_final.log 2>&1
python3 $root/tools/profile_summary.py $tag > /dev/null
# gpurun merges only gpurun_out/ back: leave copies there for the caller to move into profiles/
cp $root/profiles/${tag}_* $root/gpurun_out/profiles/
