#!/bin/bash
# Dev tool: A/B the env engine on the bench's main workload.  usage (through gpurun, repo root): tools/ab_bench.sh <reps> <lib.so>...
# Each library (SUMO_HIP_LIB override of capi.lib()) runs the bench's Ant 4096-env segment `reps` times, interleaved with the others;
# prints value (env-steps/s) and the fused launch's HIP-event time per run.
REPS=$1; shift
for r in $(seq $REPS); do
  for L in "$@"; do
    SUMO_HIP_LIB=$PWD/$L python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --ppo-nsteps 0 --spider-steps 0 --recurrent-steps 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$L', 'rep $r', '%.0f env-steps/s' % d['value'], 'kernel %.3f ms' % d['roofline']['kernel_ms'], 'contacts/fwd %.3f newton/fwd %.3f' % (d['config']['mean_contacts_per_forward'], d['config']['mean_newton_iters_per_forward']))"
  done
done
