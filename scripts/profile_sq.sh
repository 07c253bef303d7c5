#!/bin/bash
# one SQ counter pass of bench.py (wave-time split, MFMA busy, LDS conflicts) -> per-kernel sums:  profile_sq.sh <workload> <tag>
WL=$1; TAG=$2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
python3 - > $OUT/pmc_meta.json <<PYEOF
import hashlib, json, time
print(json.dumps(dict(so_sha256=hashlib.sha256(open("$ROOT/cn-rma_amd/csrc/libcnrma_hip.so","rb").read()).hexdigest(), workload="$WL", taken=time.strftime("%Y-%m-%d %H:%M:%S"))))
PYEOF
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench.py --workload $WL --no-secondary --no-cpu-baseline --no-profile --steps 4 --warmup 2 --scenes-per-step 3 --windows 1 --slots 1"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --output-format csv -d $OUT/pmc_SQ -- python3 $ARGS > /dev/null 2> $OUT/pmc_SQ.err
python3 $ROOT/scripts/pmc_sum.py $OUT/pmc_SQ $OUT/pmc_SQ.csv
rm -rf $OUT/pmc_SQ
