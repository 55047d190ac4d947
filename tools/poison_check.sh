#!/bin/bash
# Development check (GPU box, repo root): does any phase of the env engine read an LDS word nobody wrote?  Builds the engine with
# -DSUMO_DBG_POISON_LDS=<kind> -- every LDS word of the wave holds 1e300 (kind 1) / a NaN pattern (kind 2) / 0 (kind 3) before each env
# step of the per-step kernel and before each ticket of the fused rollout kernel (the world geoms' centres, which ctx_init keeps in LDS
# for the whole launch, are re-written) -- and runs the HIP-vs-oracle parity tests, the determinism / batch-independence tests and the
# fused-launch bit-identity tests against that library.  Round 3: kinds 1 and 2 pass all 26 (profiles/r03_poison_check.txt).
set -o pipefail
KIND=${1:-2}
C=robosumo_selfplay_amd/csrc
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -DSUMO_DBG_POISON_LDS=$KIND -I include -o $C/libsumo_hip_poison.so $C/sumo_engine.hip || exit 1
SUMO_HIP_LIB=$PWD/$C/libsumo_hip_poison.so python3 -m pytest tests/test_gpu_env_parity.py tests/test_gpu_ppo.py -q -k "parity or rollout_kernel_matches or determinism"
