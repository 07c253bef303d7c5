#!/bin/bash
# counter passes over the dense kernel alone (scripts/dense_ab.py):  profile_dense.sh <shape> <tag> <spec> [spec ...]
WL=$1; TAG=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/dense_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {  # name, counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ROOT/scripts/dense_ab.py $WL $SPECS > $OUT/$name.log 2> $OUT/$name.err
  python3 $ROOT/scripts/pmc_sum.py $OUT/$name $OUT/$name.csv > /dev/null
  rm -rf $OUT/$name
}
SPECS="$@"
if [ -n "$DENSE_PASSES_SHORT" ]; then
pass TCC TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
pass FETCH FETCH_SIZE
grep -h "backproject" $OUT/*.csv | cut -c1-60,150- | head -80
exit 0
fi
pass SQ SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD
pass SQ2 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES
pass TCC TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
pass FETCH FETCH_SIZE
pass WRITE WRITE_SIZE
pass TCP TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
pass TA TA_BUSY_avr TA_TA_BUSY_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum
grep -h "backproject" $OUT/*.csv | cut -c1-60,150- | head -80
