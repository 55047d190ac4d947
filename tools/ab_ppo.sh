#!/bin/bash
# Dev tool: A/B the PPO library on the bench's update segment.  usage (through gpurun, repo root): tools/ab_ppo.sh <reps> <lib.so>...
# Each library (SUMO_PPO_LIB override of ppo_capi.lib()) first runs the grad/selfplay parity tests, then the bench's PPO2 segment
# `reps` times, interleaved; prints iters/s, the SGD wall time and ppo_grad's per-call time.
REPS=$1; shift
for L in "$@"; do
  SUMO_PPO_LIB=$PWD/$L python3 -m pytest tests/test_gpu_ppo.py tests/test_gpu_sharded_update.py -q -x 2>&1 | tail -1 | sed "s|^|$L tests: |"
done
for r in $(seq $REPS); do
  for L in "$@"; do
    SUMO_PPO_LIB=$PWD/$L python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --spider-steps 0 --recurrent-steps 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
p=d['config']['ppo2']; m=d.get('roofline_mfma',{})
g=m.get('ppo_grad_kernel',{})
print('$L', 'rep $r', 'iters/s %.3f' % p['iters_per_sec'], 'sgd %.4f s' % p['sgd_s'], 'grad call %.1f us frac %.3f' % (g.get('us_per_call',0), g.get('frac',0)))"
  done
done
