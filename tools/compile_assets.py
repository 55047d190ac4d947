#!/usr/bin/env python3
"""Dev-time tool: compile every registered RoboSumo scene from an MJCF asset directory (by default the
reference checkout's ``robosumo/robosumo/envs/assets``) into ``robosumo_selfplay_amd/assets/<env-id>.json``.
The JSON files hold only derived constant tables (no MJCF text); they are what travels to the GPU box.
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from robosumo_selfplay_amd import mjcf  # noqa: E402


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/robosumo/robosumo/envs/assets"
    out = os.path.join(os.path.dirname(__file__), "..", "robosumo_selfplay_amd", "assets")
    os.makedirs(out, exist_ok=True)
    for env_id in mjcf.registry():
        if mjcf.canonical_id(env_id) != env_id:
            continue
        m = mjcf.compile_env(env_id, src)
        with open(os.path.join(out, env_id + ".json"), "w") as f:
            f.write(m.to_json())
        print(env_id, m.dims(), "obs", m.obs_dims, "act", m.act_dims, "mass", round(float(m.body_mass.sum()), 4))


if __name__ == "__main__":
    main()
