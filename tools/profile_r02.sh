#!/bin/bash
# Round-2 profile collection on the GPU box (run through gpurun from the repo root).  Every rocprofv3 call puts the program
# itself after `--`; counters are collected in their own passes with --kernel-trace only (MI355X_MICROARCH.md, HBM/rocprofv3).
# Profiles the bench default: the fused rollout launch (sumo_rollout_kernel), 4096 envs x K steps per launch.
set -o pipefail
export TMPDIR=/tmp
TAG=${1:-r02b}
K=20
O=gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
B="python3 bench.py --steps $K --warmup $K --state-warmup 100 --no-cpu-baseline --ppo-nsteps 0 --spider-steps 0 --recurrent-steps 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o $TAG -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --spider-steps 0 --recurrent-steps 32 > $O/stats.log 2>&1 || exit 11
echo stats done
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sq -o $TAG -- $B > $O/sq.log 2>&1 || exit 12
echo sq done
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $O/mix -o $TAG -- $B > $O/mix.log 2>&1 || exit 13
echo mix done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o $TAG -- $B > $O/fetch.log 2>&1 || exit 14
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o $TAG -- $B > $O/write.log 2>&1 || exit 15
echo traffic done
SQ=$(find $O/sq -name "*counter_collection.csv" | head -1); MIX=$(find $O/mix -name "*counter_collection.csv" | head -1)
FE=$(find $O/fetch -name "*counter_collection.csv" | head -1); WR=$(find $O/write -name "*counter_collection.csv" | head -1)
# launches of K steps: the state warm-up passes (100 steps as ring passes of K) and the timing warm-up come first; skip=1 drops the first
python3 tools/pmc_summary.py sq2 $SQ $MIX $O/${TAG}_pmc_sq.json kernel=sumo_rollout_kernel steps=$K envs=4096 > $O/sq_summary.log 2>&1 || exit 16
python3 tools/pmc_summary.py traffic $FE $WR $O/${TAG}_pmc_traffic.json kernel=sumo_rollout_kernel steps=$K envs=4096 > $O/traffic_summary.log 2>&1 || exit 17
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench_kernel_stats.csv
rm -rf $O/sq $O/mix $O/fetch $O/write $O/stats
ls -la $O
