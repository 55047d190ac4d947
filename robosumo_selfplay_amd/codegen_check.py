"""Static check of a built HIP library for a code-generation hazard of this toolchain found in round 3 ("partial-EXEC
save copies"): the register allocator parks a long-lived VGPR in a spare register with a `v_mov_b32 vA, vB` (or re-materialises a constant
into vA) and places that copy at the top of a join block BEFORE the `s_or_b64 exec, exec, ...` that re-activates the lanes -- i.e. inside
the predicated region `s_and_saveexec_b64 ... s_or_b64 exec`.  The copy then happens only in the lanes the region left active; the later
restore `v_mov_b32 vB, vA` runs with all lanes active and hands the other lanes whatever vA held before: the previous wave's leftovers.
Observed in the Ant kernels (256-register budget): lanes >= nv got stale LDS addresses / offsets from the second forward-dynamics call on,
a few env steps per 10^4 diverged, run-to-run nondeterministically (tools/determinism_diag.py; tools/scrub_bisect.py located the register).

The signature looked for: inside a branch-free predicated region, a plain VGPR move whose destination (a) is not consumed inside the region,
(b) has NO other write in the whole kernel (a conditional assignment `if (c) x = v` always has one: x's earlier value) and (c) is read
somewhere.  `scan_library` lists such copies per kernel; the build warns about them and tests/test_codegen_check.py fails on them."""
import os
import re
import subprocess
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"

def disassemble(lib):
    d = tempfile.mkdtemp()
    fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    return subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", co], text=True)


def vregs(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def kernels(asm):
    cur, body = None, []
    for line in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            if cur:
                yield cur, body
            cur, body = m.group(1), []
            continue
        if cur:
            t = line.split("//")[0].strip()
            if t:
                body.append(t)
    if cur:
        yield cur, body


_BREAK = ("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc", "s_and_saveexec_b64", "s_or_saveexec_b64")


def regions(body):
    """Yields (first, last) instruction indices of the bodies of branch-free predicated regions `s_and_saveexec_b64 sX, .. ; ... ; s_or_b64 exec, exec, sX`."""
    n = len(body)
    nxt = 0
    for i, t in enumerate(body):
        if i < nxt or not t.startswith("s_and_saveexec_b64 "):
            continue
        save = t[len("s_and_saveexec_b64 "):].split(",", 1)[0].strip()
        close = "s_or_b64 exec, exec, " + save
        j, ok = i + 1, False
        while j < n and j < i + 200:
            u = body[j]
            if u.startswith(_BREAK):
                break
            if u.startswith(close):
                ok = True
                break
            j += 1
        if ok:
            yield i + 1, j
            nxt = j + 1


def check(body):
    """-> list of (index, text) of register moves inside such a region whose destination the region itself does not consume."""
    bad = []
    for a, b in regions(body):
        region = body[a:b]
        for k, t in enumerate(region):
            mm = re.match(r"v_mov_b32_e32 v(\d+), (.+)$", t) or re.match(r"v_mov_b64_e32 v\[(\d+):\d+\], (.+)$", t)
            if not mm:
                continue
            dst = int(mm.group(1))
            used = any(dst in vregs(u.split(None, 1)[1] if " " in u else "") and not re.match(r"v_mov_b(32|64)_e32 v\[?%d\b" % dst, u)
                       for u in region[k + 1:])
            if not used:
                bad.append((a + k, t))
    return bad


def check_before_endcf(body):
    """-> list of (index, text): register moves that sit directly in front of an `s_or_b64 exec, exec, sX` (the END_CF at the top of a join
    block) -- the placement itself, whether or not the predicated region in front of it still has its skip branch."""
    bad = []
    for j, t in enumerate(body):
        if not t.startswith("s_or_b64 exec, exec, "):
            continue
        k = j - 1
        while k >= 0 and j - k <= 8 and body[k].startswith(("v_mov_b32_e32", "v_mov_b64_e32", "s_mov_b32", "s_mov_b64", "s_nop")):
            if body[k].startswith("v_mov"):
                bad.append((k, body[k]))
            k -= 1
    return bad


def check_spills(body):
    """-> list of (index, text): VGPR spill stores inside such a region (the same placement through the scratch frame: only the region's
    lanes reach the slot).  The shipped kernels have none."""
    return [(k, body[k]) for a, b in regions(body) for k in range(a, b) if body[k].startswith("scratch_store")]


_STORES = ("ds_write", "global_store", "scratch_store", "flat_store", "buffer_store", "global_atomic", "ds_add")
_NO_VDST = ("s_", "v_cmp", "v_cmpx")


def scan_kernel(body):
    """-> [(instruction index, text, reads)] of suspicious copies in one kernel's instruction list."""
    cands = check(body)
    seen = {i for i, _ in cands}
    cands += [(i, t) for i, t in check_before_endcf(body) if i not in seen]
    out = []
    if cands:
        writes, reads = {}, {}                      # per VGPR: instructions that write / read it (one pass over the kernel)
        for u in body:
            if " " not in u or "v" not in u:
                continue
            op, rest = u.split(None, 1)
            ops = rest.split(",")
            if not op.startswith(_STORES) and not op.startswith(_NO_VDST):
                for r in vregs(ops[0]):
                    writes[r] = writes.get(r, 0) + 1
                src = ",".join(ops[1:])
            else:
                src = rest
            for r in vregs(src):
                reads[r] = reads.get(r, 0) + 1
        for idx, t in cands:
            dst = int(re.search(r"v\[?(\d+)", t).group(1))
            if writes.get(dst, 0) == 1 and reads.get(dst, 0) >= 1:
                out.append((idx, t, reads[dst]))
    out += [(idx, t, 0) for idx, t in check_spills(body)]
    return out


def scan_library(lib, name_filters=()):
    """-> {kernel name: [(index, text, reads)]} for the kernels of `lib` (a .so with an embedded gfx950 code object) that hold suspicious copies."""
    hits = {}
    for name, body in kernels(disassemble(lib)):
        if name_filters and not any(s in name for s in name_filters):
            continue
        h = scan_kernel(body)
        if h:
            hits[name] = h
    return hits
