#!/bin/bash
# executed-MFMA counter pass over the convolutions of one eager scene:  profile_mfma.sh <workload> <tag> [precision]
WL=$1; TAG=$2; PREC=${3:-f16x3}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/mfma_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
MOPS=SQ_INSTS_VALU_MFMA_MOPS_F16; [ "$PREC" = f32 ] && MOPS=SQ_INSTS_VALU_MFMA_MOPS_F32
rocprofv3 --kernel-trace --pmc $MOPS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc \
  -- python3 $ROOT/scripts/mfma_count.py $WL $OUT/layers.json $PREC > $OUT/run.log 2> $OUT/run.err
python3 $ROOT/scripts/mfma_join.py $OUT/pmc $OUT/layers.json > $OUT/mfma_${WL}_${PREC}.log 2>> $OUT/run.err
tail -4 $OUT/mfma_${WL}_${PREC}.log
rm -rf $OUT/pmc
