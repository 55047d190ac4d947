#!/bin/bash
# development: WRITE_SIZE / FETCH_SIZE of the fused rollout launch only (two short passes)
export TMPDIR=/tmp
O=gpurun_out/qpmc; rm -rf $O; mkdir -p $O
B="python3 bench.py --steps 20 --warmup 20 --state-warmup 40 --no-cpu-baseline --ppo-nsteps 0 --spider-steps 0"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o q -- $B > $O/write.log 2>&1 || exit 15
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o q -- $B > $O/fetch.log 2>&1 || exit 14
python3 tools/pmc_summary.py traffic $(find $O/fetch -name "*counter_collection.csv") $(find $O/write -name "*counter_collection.csv") $O/q_traffic.json kernel=sumo_rollout_kernel steps=20 envs=4096 | grep -E "per_env_step|write_bytes|fetch_bytes"
rm -rf $O/write $O/fetch
