"""MI355X-native RoboSumo self-play hot path (device-resident vectorised env step + PPO2 rollout/update).

See DESIGN.md for scope.  Importing this package does not touch the GPU and does not load the HIP
library; ``robosumo_selfplay_amd.capi`` does that lazily and fails loudly when it is missing.
"""
from .mjcf import SumoModel, compile_env, compile_scene, load_model, registry  # noqa: F401

__all__ = ["SumoModel", "compile_env", "compile_scene", "load_model", "registry"]
