#!/bin/bash
# rocprofv3 kernel stats of training steps at the ScanNet shape under bf16 autocast:  profile_train.sh <tag>
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CNRMA_PROBE_AB=0     # 2 warm-up + 28 timed steps of the default configuration (30 steps in the kernel stats)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/scripts/train_probe.py S bf16 > $OUT/train_probe.log 2> $OUT/stats.err
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/stats
tail -3 $OUT/train_probe.log
