"""CPU-side checks of the drop-in boundary: the library loads and exports every symbol include/sumo_hip.h declares;
the product path refuses to run without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT, has_gpu
from robosumo_selfplay_amd import build, capi


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b((?:sumo|ppo)_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    build.build_all()
    path = build.lib_path("libsumo_hip.so")
    assert os.path.exists(path)
    L = ctypes.CDLL(path)
    names = _declared("sumo_hip.h")
    assert set(names) >= set(capi.EXPORTS) - {"sumo_last_error"} and len(names) >= 9
    for n in names:
        assert hasattr(L, n), n


def test_model_header_parser_roundtrip(ant_model):
    # sumo_model_parse is header-only; the oracle library embeds it -- a malformed blob must be rejected
    from oracle import oracle
    L = oracle.lib()
    blob = bytearray(ant_model.to_blob())
    blob[8] ^= 0xFF  # corrupt the magic
    buf = (ctypes.c_char * len(blob)).from_buffer(blob)
    assert not L.so_create(ctypes.byref(buf), ctypes.c_size_t(len(blob)), 1)
    assert b"bad model blob" in L.so_last_error()


@pytest.mark.skipif(has_gpu(), reason="only meaningful without a GPU")
def test_no_cpu_fallback(ant_model):
    with pytest.raises(capi.SumoHipError):
        capi.Engine(ant_model, 4)
    from robosumo_selfplay_amd.vec_env import SumoVecEnv
    with pytest.raises(capi.SumoHipError):
        SumoVecEnv("RoboSumo-Ant-vs-Ant-v0", num_envs=2)


def test_product_package_never_imports_oracle():
    """The oracle is test infrastructure: nothing under robosumo_selfplay_amd/ (nor run.py) may import, load or link it."""
    import re
    bad = re.compile(r"^\s*(from\s+oracle|import\s+oracle)|libsumo_oracle|oracle/|#include\s+\"[^\"]*oracle", re.M)
    files = [os.path.join(ROOT, "run.py")]
    for dp, _, fs in os.walk(os.path.join(ROOT, "robosumo_selfplay_amd")):
        files += [os.path.join(dp, f) for f in fs if f.endswith((".py", ".hip", ".h", ".cpp"))]
    for f in files:
        m = bad.search(open(f).read())
        assert m is None, (f, m.group(0))


def test_ppo_library_exports_every_declared_symbol():
    from robosumo_selfplay_amd import ppo_capi
    build.build_all()
    path = build.lib_path("libsumo_ppo.so")
    assert os.path.exists(path)
    L = ctypes.CDLL(path)
    names = _declared("sumo_ppo.h")
    assert set(names) == set(ppo_capi.EXPORTS)
    for n in names:
        assert hasattr(L, n), n
    L.ppo_param_count.restype = ctypes.c_int
    assert L.ppo_param_count(121, 8) == 24529           # MLP(64,64) + copy value net on 121-d obs


def test_static_layout_header_is_current():
    """csrc/layout_static.h (compile-time Layout / model / aux constants of the flagship scene, used by the static kernel variants)
    must equal what the engine's own host code computes for RoboSumo-Ant-vs-Ant-v0 -- otherwise sumo_create silently falls back to
    the runtime-Layout kernels (results identical, ~5 % slower)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_static_layout.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-500:] + r.stderr[-500:]


def test_layout_fits_the_residency_and_the_pool_takes_the_spare_lds():
    """build_layout through the host-only export: every scene's env record fits its residency (8 envs per CU for Ant-vs-Ant, 4 for the
    Spiders: 160 KB in granules of 1280 B), the contact-Jacobian pool holds at least one half per contact record and takes the LDS the
    residency leaves unused (round 3: Ant 18 records / 29 halves, Spider 40 / 52)."""
    import ctypes as C
    import re
    from robosumo_selfplay_amd import build, mjcf
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "robosumo_selfplay_amd", "csrc", "sumo_engine.hip")).read()
    body = src[src.index("struct Layout {"):]
    body = re.sub(r"//[^\n]*", "", body[body.index("{") + 1:body.index("};")])
    fields = [f.strip() for decl in body.split(";") if decl.strip() for f in decl.strip()[4:].split(",")]
    lib = C.CDLL(build.build_all()[0])
    lib.sumo_debug_layout.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
    want = {"RoboSumo-Ant-vs-Ant-v0": (8, 18, 29), "RoboSumo-Spider-vs-Spider-v0": (4, 40, 52)}
    for env_id in ("RoboSumo-Ant-vs-Ant-v0", "RoboSumo-Bug-vs-Bug-v0", "RoboSumo-Spider-vs-Spider-v0", "RoboSumo-Ant-vs-Spider-v0"):
        blob = mjcf.load_model(env_id).to_blob()
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        out = (C.c_int32 * 256)()
        n = lib.sumo_debug_layout(C.cast(buf, C.c_void_p), len(blob), out, 256)
        assert n == len(fields)
        L = dict(zip(fields, [int(out[i]) for i in range(n)]))
        slots = 160 * 1024 // L["total_bytes"]
        granules = -(-L["total_bytes"] // 1280)
        assert slots >= 4 and slots * granules * 1280 <= 160 * 1024, (env_id, L["total_bytes"], slots)
        assert L["maxcon"] <= L["jbcap"] <= 2 * L["maxcon"], (env_id, L["maxcon"], L["jbcap"])
        spare = 160 * 1024 // slots // 1280 * 1280 - L["total_bytes"]
        assert 0 <= spare < 192 or L["jbcap"] == 2 * L["maxcon"], (env_id, spare)       # no room for another half (192 B) is left unused
        if env_id in want:
            assert (slots, L["maxcon"], L["jbcap"]) == want[env_id], (env_id, slots, L["maxcon"], L["jbcap"])
