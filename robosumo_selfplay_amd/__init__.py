"""MI355X-native RoboSumo self-play hot path (device-resident vectorised env step + PPO2 rollout/update).

See DESIGN.md for scope.  Importing this package does not touch the GPU and does not load the HIP
library; ``robosumo_selfplay_amd.capi`` does that lazily and fails loudly when it is missing.
"""
from . import hostcfg as _hostcfg

_hostcfg.apply()          # cap BLAS / OpenMP / torch pools at the cgroup CPU quota BEFORE numpy spawns them (see hostcfg.py)

from .mjcf import SumoModel, compile_env, compile_scene, load_model, registry  # noqa: E402,F401

__all__ = ["SumoModel", "compile_env", "compile_scene", "load_model", "registry"]
