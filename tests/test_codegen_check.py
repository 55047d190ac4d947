"""The built HIP libraries must not contain the partial-EXEC save copies described in robosumo_selfplay_amd/codegen_check.py (a register
allocator placement that made the Ant kernels nondeterministic in round 3).  CPU test: disassembles the in-tree code objects."""
import os
import shutil

import pytest

from robosumo_selfplay_amd import build, codegen_check


def _tools():
    return shutil.which("objcopy") and os.path.exists(os.path.join(codegen_check.LLVM, "llvm-objdump"))


@pytest.mark.skipif(not _tools(), reason="objcopy / llvm-objdump not available")
@pytest.mark.parametrize("lib", ["libsumo_hip.so", "libsumo_ppo.so"])
def test_no_partial_exec_save_copies(lib):
    paths = [p for p in build.build_all() if p.endswith(lib)]
    assert paths, "library %s was not built" % lib
    hits = codegen_check.scan_library(paths[0])
    assert not hits, {k: [t for _, t, _ in v] for k, v in hits.items()}


def test_checker_recognises_the_pattern():
    """The signature on a hand-made instruction list: a save copy inside a branch-free predicated region (flagged) next to a conditional
    assignment and a copy that the region itself consumes (both fine)."""
    body = [
        "v_mov_b32_e32 v5, 0",                       # x = 0
        "v_mov_b32_e32 v9, 1",
        "s_and_saveexec_b64 s[2:3], s[92:93]",
        "ds_read_b64 v[116:117], v12 offset:6176",
        "v_mov_b32_e32 v5, v9",                      # if (c) x = y: v5 has an earlier write -> fine
        "v_mov_b32_e32 v13, v12",                    # consumed inside the region -> fine
        "ds_read_b64 v[118:119], v13 offset:8",
        "v_mov_b32_e32 v140, v237",                  # save slot: only write of v140, read much later -> flagged
        "s_mov_b32 s50, s11",
        "s_or_b64 exec, exec, s[2:3]",
        "v_add_u32_e32 v237, v247, v8",
        "v_mov_b32_e32 v237, v140",
        "v_add_u32_e32 v1, v5, v237",
        "s_endpgm",
    ]
    hits = codegen_check.scan_kernel(body)
    assert [t for _, t, _ in hits] == ["v_mov_b32_e32 v140, v237"]
