#!/usr/bin/env python3
"""CLI of robosumo_selfplay_amd/codegen_check.py: lists the partial-EXEC save copies of a built library.  usage: exec_copy_check.py lib.so [kernel-name-substring ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robosumo_selfplay_amd import codegen_check


def main():
    hits = codegen_check.scan_library(sys.argv[1], tuple(sys.argv[2:]))
    total = 0
    for name, h in hits.items():
        print("%s: %d suspicious copies" % (name[:90], len(h)))
        for idx, t, nr in h[:12]:
            print("   @%d  %s   (its only write in the kernel; read at %d places)" % (idx, t, nr))
        total += len(h)
    print("total suspicious copies: %d" % total)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
