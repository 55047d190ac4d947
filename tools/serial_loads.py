#!/usr/bin/env python3
"""Development: per kernel of a built library, how many vector-memory loads does the code wait for ONE (or two) AT A TIME (`s_waitcnt vmcnt(0)`
with at most two operations in flight)?  Such loads are serial trips to memory; the usual cause in this code base was `ok ? p[i] : 0` -- a
select right behind a load makes the compiler predicate the load and wait inside the predicated block -- and per-lane indexing of small tables
in kernel arguments.  Fixes that worked: load unconditionally at a clamped index and select afterwards; unroll table lookups over compile-time
counts; `__builtin_amdgcn_sched_barrier(0)` after a block of loads that the scheduler sinks to their uses (ppo_grad_kernel 66 -> 48 us per call,
ppo_wgrad_kernel 156 -> 63 us, ppo_lstm_seq_fwd_kernel 970 -> 863 us).  Static count: weigh it by how often the code runs, and note that the LAST one or two loads of a
properly pipelined batch (vmcnt 2, 1, 0) are counted as well -- look at the disassembly before acting on a number.
usage: serial_loads.py lib.so [kernel-name-substring ...]"""
import os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robosumo_selfplay_amd import codegen_check

subs = sys.argv[2:]
for name, body in codegen_check.kernels(codegen_check.disassemble(sys.argv[1])):
    if subs and not any(s in name for s in subs):
        continue
    outstanding = ser = nl = 0
    for t in body:
        op = t.split()[0]
        if op.startswith(("global_load", "scratch_load", "flat_load", "buffer_load")):
            outstanding += 1; nl += 1
        elif op.startswith(("global_store", "scratch_store", "global_atomic", "flat_store")):
            outstanding += 1
        elif op == "s_waitcnt" and "vmcnt" in t:
            k = int(re.search(r"vmcnt\((\d+)\)", t).group(1))
            if outstanding > k:
                if outstanding <= 2 and k == 0:
                    ser += outstanding
                outstanding = k
    if nl:
        print("%-90s instructions %6d  vector loads %4d  waited one or two at a time %4d" % (name[:90], len(body), nl, ser))
