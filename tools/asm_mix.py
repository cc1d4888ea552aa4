#!/usr/bin/env python3
"""Instruction mix per basic block of one kernel in a hipcc -S listing (blocks with many MFMAs = the persistent loops).
Usage: asm_mix.py file.s <substring of the mangled kernel name> [min_mfma]"""
import collections
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and key in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, cur, name = [], [], "entry"
    for l in (x.strip() for x in lines[start + 1:end]):
        if re.match(r"^\.LBB\d+_\d+:", l):
            blocks.append((name, cur))
            name, cur = l, []
        elif l and l[0] not in ";.":
            cur.append(l)
    blocks.append((name, cur))
    for name, b in blocks:
        if sum(1 for x in b if x.startswith("v_mfma")) < min_mfma:
            continue
        c = collections.Counter()
        for x in b:
            op = x.split()[0]
            if op.startswith("v_mfma"):
                c["mfma"] += 1
            elif op.startswith(("ds_", "global_", "buffer_", "scratch_", "flat_")):
                c[op] += 1
            elif op.startswith("s_waitcnt"):
                c["s_waitcnt"] += 1
            elif op.startswith("s_"):
                c["salu"] += 1
            elif op in ("v_exp_f32", "v_rcp_f32", "v_log_f32", "v_sqrt_f32", "v_rsq_f32"):
                c["trans:" + op] += 1
            elif op.startswith("v_"):
                c["valu"] += 1
            else:
                c[op] += 1
        print(name, len(b), dict(sorted(c.items())))


if __name__ == "__main__":
    main()
