#!/bin/bash
# counter passes over ONE convolution layer class (scripts/conv_sweep.py with ONLY=Cin,Cout,K,minrows):
#   profile_conv_layer.sh <workload> <tag> <Cin,Cout,K,minrows> <spec> [spec ...]
WL=$1; TAG=$2; export ONLY=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/convlayer_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SPECS="$@"
pass() {
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ROOT/scripts/conv_sweep.py $WL $SPECS > $OUT/$name.log 2> $OUT/$name.err
  python3 $ROOT/scripts/pmc_sum.py $OUT/$name $OUT/$name.csv > /dev/null
  rm -rf $OUT/$name
}
pass SQ SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA
pass SQ2 SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES
pass SQ3 SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT
pass TCP TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
pass TA TA_BUSY_avr TA_TA_BUSY_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum
pass TCC TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
grep -h "sparse_conv" $OUT/*.csv | cut -c30-100,150- | head -120
