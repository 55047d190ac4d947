#!/bin/bash
# Round-3 profile collection on the GPU box (run through gpurun from the repo root): counters for EVERY BASELINE config's rollout
# kernel and for the matrix-core kernels.  Every rocprofv3 call puts the program itself after `--`; counters are collected in their own
# passes with --kernel-trace only (MI355X_MICROARCH.md, HBM/rocprofv3).  usage: tools/profile_r03.sh <tag> <workload>...
#   workloads: ant spider rec1024 rec4096 (tools/prof_workload.py: 5 priming launches of 20 steps, then 3 profiled ones) | mfma
set -o pipefail
export TMPDIR=/tmp
TAG=${1:-r03}; shift
K=20
for W in "$@"; do
  O=gpurun_out/prof_${TAG}_$W
  rm -rf $O; mkdir -p $O
  P="python3 tools/prof_workload.py $W --steps $K --launches 3"
  case $W in
    ant)     KS="sumo_rollout_kernel<28, 0,"; ENVS=4096;;
    spider)  KS="sumo_rollout_kernel<44, 0,"; ENVS=4096;;
    rec1024) KS="sumo_rollout_kernel<28, 1,"; ENVS=1024;;
    rec4096) KS="sumo_rollout_kernel<28, 1,"; ENVS=4096;;
    mfma)    KS="ppo_grad_kernel"; ENVS=0;;
  esac
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o $TAG -- $P > $O/stats.log 2>&1 || exit 11
  cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/${TAG}_${W}_kernel_stats.csv
  grep "^{" $O/stats.log | tail -1 > $O/${TAG}_${W}_workload.json
  echo "$W stats done"
  if [ "$W" = "mfma" ]; then
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/mf -o $TAG -- $P > $O/mf.log 2>&1 || exit 12
    MF=$(find $O/mf -name "*counter_collection.csv" | head -1)
    # algorithmic flops: ppo_grad 3 x 2 x (pi + vf) MACs x 16384 rows; ppo_selfplay 2 x (2 x (2 pi + vf)) x 4096 envs (bench.py::mfma_probe)
    python3 tools/pmc_summary.py mfma $MF $O/${TAG}_pmc_mfma_grad.json "kernel=ppo_grad_kernel" skip=4 flops=2384461824 "note=16384-row PPO2 minibatch, Ant MLP(64,64) pi + vf" > $O/mfma_grad_summary.log 2>&1 || exit 13
    python3 tools/pmc_summary.py mfma $MF $O/${TAG}_pmc_mfma_selfplay.json "kernel=ppo_selfplay_kernel" skip=4 flops=599785472 "note=4096 envs, the five evaluations of a rollout step" > $O/mfma_selfplay_summary.log 2>&1 || exit 14
    rm -rf $O/mf $O/stats
    echo "mfma pmc done"
    continue
  fi
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sq -o $TAG -- $P > $O/sq.log 2>&1 || exit 12
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $O/mix -o $TAG -- $P > $O/mix.log 2>&1 || exit 13
  echo "$W sq/mix done"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o $TAG -- $P > $O/fetch.log 2>&1 || exit 14
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o $TAG -- $P > $O/write.log 2>&1 || exit 15
  echo "$W traffic done"
  SQ=$(find $O/sq -name "*counter_collection.csv" | head -1); MIX=$(find $O/mix -name "*counter_collection.csv" | head -1)
  FE=$(find $O/fetch -name "*counter_collection.csv" | head -1); WR=$(find $O/write -name "*counter_collection.csv" | head -1)
  NOTE="note=tools/prof_workload.py $W: $ENVS envs, one env group, launches of $K steps, 5 priming launches skipped"
  python3 tools/pmc_summary.py sq2 $SQ $MIX $O/${TAG}_${W}_pmc_sq.json "kernel=$KS" steps=$K envs=$ENVS skip=5 "$NOTE" > $O/sq_summary.log 2>&1 || exit 16
  python3 tools/pmc_summary.py traffic $FE $WR $O/${TAG}_${W}_pmc_traffic.json "kernel=$KS" steps=$K envs=$ENVS skip=5 "$NOTE" > $O/traffic_summary.log 2>&1 || exit 17
  rm -rf $O/sq $O/mix $O/fetch $O/write $O/stats
  ls $O
done
