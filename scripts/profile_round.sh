#!/bin/bash
# rocprofv3 passes of bench.py for profiles/ (run on the GPU box through gpurun):  profile_round.sh <workload> <tag>
WL=$1; TAG=$2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
python3 - > $OUT/pmc_meta.json <<PYEOF
import hashlib, json, time
print(json.dumps(dict(so_sha256=hashlib.sha256(open("$ROOT/cn-rma_amd/csrc/libcnrma_hip.so","rb").read()).hexdigest(), workload="$WL", taken=time.strftime("%Y-%m-%d %H:%M:%S"))))
PYEOF
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --workload $WL --no-secondary --no-cpu-baseline --no-profile --steps 4 --warmup 2 --scenes-per-step 3 --windows 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ARGS > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 $ARGS > /dev/null 2> $OUT/pmc_$c.err
  python3 $ROOT/scripts/pmc_sum.py $OUT/pmc_$c $OUT/pmc_$c.csv
done
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_TCC -- python3 $ARGS > /dev/null 2> $OUT/pmc_TCC.err
python3 $ROOT/scripts/pmc_sum.py $OUT/pmc_TCC $OUT/pmc_TCC.csv
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/stats $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_TCC
ls -la $OUT
