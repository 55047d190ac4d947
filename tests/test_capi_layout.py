"""The ctypes mirrors of the C-ABI structs (capi.Rollout, capi.RolloutLstm, ppo_capi.LstmNet) against the layout the C compiler gives
the structs of include/*.h: sizes and field offsets, from a small program compiled with gcc.  (A hand-written mirror that drifts from
its header passes garbage pointers to a kernel.)"""
import ctypes as C
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from robosumo_selfplay_amd import capi, ppo_capi  # noqa: E402

STRUCTS = [("sumo_rollout", capi.Rollout), ("sumo_rollout_lstm", capi.RolloutLstm), ("ppo_lstm_net", ppo_capi.LstmNet)]


def _c_layout(tmp_path):
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "sumo_hip.h"', '#include "sumo_ppo.h"', 'int main(void) {']
    for cname, st in STRUCTS:
        lines.append('  printf("%s size %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in st._fields_:
            lines.append('  printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    table = {}
    for ln in out.splitlines():
        a, b, v = ln.split()
        table[(a, b)] = int(v)
    return table


def test_ctypes_mirrors_match_the_headers(tmp_path):
    table = _c_layout(tmp_path)
    for cname, st in STRUCTS:
        assert C.sizeof(st) == table[(cname, "size")], cname
        for fname, _ in st._fields_:
            assert getattr(st, fname).offset == table[(cname, fname)], (cname, fname)
        # every field of the header is mirrored: the last mirrored field ends where the struct (up to tail padding) ends
        last = st._fields_[-1][0]
        assert getattr(st, last).offset + getattr(st, last).size + 8 > table[(cname, "size")], cname


def test_headers_are_plain_c(tmp_path):
    """The boundary is a C ABI: both headers compile as C99 on their own, in either order."""
    for order in (("sumo_hip.h", "sumo_ppo.h"), ("sumo_ppo.h", "sumo_hip.h"), ("sumo_model.h",)):
        src = tmp_path / "hdr.c"
        src.write_text("".join('#include "%s"\n' % h for h in order) + "int main(void) { return 0; }\n")
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)], check=True)
