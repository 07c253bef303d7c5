#!/bin/bash
# kernel-trace statistics of bench.py's graph replays with a given number of scene graphs in flight (1 = per-kernel cost without
# co-scheduled kernels of other scenes; 3 = the default bench configuration):  profile_stats.sh <workload> <tag> <slots>
WL=$1; TAG=$2; SLOTS=${3:-1}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --workload $WL --no-secondary --no-cpu-baseline --no-profile --steps 4 --warmup 2 --scenes-per-step 3 --windows 1 --slots $SLOTS"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ARGS > $OUT/bench_under_rocprof_slots$SLOTS.json 2> $OUT/stats.err
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_slots$SLOTS.csv
rm -rf $OUT/stats
head -12 $OUT/kernel_stats_slots$SLOTS.csv | cut -c1-200
