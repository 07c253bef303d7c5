#!/bin/bash
# every committed measurement of a round from ONE build, laid out with the names profiles/ uses:  profile_final.sh <round tag, e.g. r06>
# (run through gpurun; copies land in gpurun_out/<tag>_final/, to be copied into profiles/)
R=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
F=$ROOT/gpurun_out/${R}_final
mkdir -p $F
for WL in S NS; do
  wl=$(echo $WL | tr A-Z a-z)
  bash $ROOT/scripts/profile_round.sh $WL ${R}f_$wl > /dev/null 2>&1
  P=$ROOT/gpurun_out/prof_${R}f_$wl
  cp $P/kernel_stats.csv $F/${R}_${wl}_graph_kernel_stats.csv                 # three scene graphs in flight (the bench configuration)
  cp $P/bench_under_rocprof.json $F/${R}_${wl}_bench_under_rocprof.json
  for c in FETCH_SIZE WRITE_SIZE TCC; do cp $P/pmc_$c.csv $F/${R}_${wl}_pmc_$c.csv; done
  [ $WL = NS ] && cp $P/pmc_meta.json $F/${R}_pmc_meta.json
  bash $ROOT/scripts/profile_sq.sh $WL ${R}f_sq_$wl > /dev/null 2>&1
  cp $ROOT/gpurun_out/prof_${R}f_sq_$wl/pmc_SQ.csv $F/${R}_${wl}_pmc_SQ.csv
  bash $ROOT/scripts/profile_stats.sh $WL ${R}f_1_$wl 1 > /dev/null 2>&1
  cp $ROOT/gpurun_out/prof_${R}f_1_$wl/kernel_stats_slots1.csv $F/${R}_${wl}_graph_kernel_stats_1slot.csv   # ONE scene in flight: per-kernel cost
done
NS_N=$(python3 -c "import json;print(json.load(open('$F/${R}_ns_bench_under_rocprof.json'))['graph_nodes_per_scene'])")
S_N=$(python3 -c "import json;print(json.load(open('$F/${R}_s_bench_under_rocprof.json'))['graph_nodes_per_scene'])")
bash $ROOT/scripts/profile_sequence.sh S ${R}f_seq_s $S_N > /dev/null 2>&1
bash $ROOT/scripts/profile_sequence.sh NS ${R}f_seq_ns $NS_N > /dev/null 2>&1
cp $ROOT/gpurun_out/prof_${R}f_seq_s/scene_sequence_S.log $F/${R}_s_scene_sequence_1slot.log
cp $ROOT/gpurun_out/prof_${R}f_seq_ns/scene_sequence_NS.log $F/${R}_ns_scene_sequence_1slot.log
bash $ROOT/scripts/profile_train.sh ${R}f_train > /dev/null 2>&1
cp $ROOT/gpurun_out/prof_${R}f_train/kernel_stats.csv $F/${R}_train_step_S_bf16_kernel_stats.csv
cp $ROOT/gpurun_out/prof_${R}f_train/train_probe.log $F/${R}_train_probe.log
ls -la $F
