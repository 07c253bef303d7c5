#!/bin/bash
# kernel sequence of one replayed scene graph (one scene in flight):  profile_sequence.sh <workload> <tag> <nodes per scene>
WL=$1; TAG=$2; N=$3
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --workload $WL --no-secondary --no-cpu-baseline --no-profile --steps 2 --warmup 1 --scenes-per-step 2 --windows 1 --slots 1 --scenes 2"
rocprofv3 --kernel-trace --output-format csv -d $OUT/seq -- python3 $ARGS > $OUT/seq_bench.json 2> $OUT/seq.err
python3 $ROOT/scripts/scene_sequence.py $OUT/seq $N > $OUT/scene_sequence_$WL.log
rm -rf $OUT/seq
tail -1 $OUT/scene_sequence_$WL.log
