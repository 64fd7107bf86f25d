#!/usr/bin/env python3
"""Instruction mix of the barrier-delimited hot loops of one kernel of the built library (no GPU needed):

    python tools/loop_stats.py <liblns_hip.so> <kernel symbol substring> [min MFMAs per segment]

Splits the kernel's disassembly at s_barrier and prints, for every segment with at least that many MFMAs, the count of
MFMA / VALU / transcendental / SALU / LDS / vector-memory / waitcnt instructions (accumulator zero-initialisation excluded).
"""
import collections
import re
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import check_isa


def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_accvgpr"): return "acc_mov"
    if op.startswith("ds_read"): return "ds_read"
    if op.startswith("ds_write"): return "ds_write"
    if op.startswith(("global_load", "buffer_load")): return "vmem_load"
    if op.startswith(("global_store", "buffer_store")): return "vmem_store"
    if op.startswith(("v_exp", "v_rcp", "v_log", "v_rsq", "v_sqrt")): return "trans"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_"): return "salu"
    return op


def segments(text, sub, min_mfma=12):
    """[(kernel symbol, {class: count}, Counter of VALU opcodes)] for every barrier-delimited segment with >= min_mfma MFMAs
    of the kernels whose symbol contains `sub`."""
    out = []
    for m in re.finditer(r"^[0-9a-f]+ <([^>]+)>:$", text, re.M):
        if sub not in m.group(1):
            continue
        end = re.search(r"^[0-9a-f]+ <[^>]+>:$", text[m.end():], re.M)
        body = text[m.end(): m.end() + end.start()] if end else text[m.end():]
        ops = [l.split()[0] for l in body.splitlines() if l.strip() and not l.strip().startswith("//")]
        seg = []
        for op in ops + ["s_barrier"]:
            if op == "s_barrier":
                c = collections.Counter(classify(o) for o in seg)
                if c["mfma"] >= min_mfma:
                    out.append((m.group(1), dict(c), collections.Counter(o for o in seg if classify(o) == "valu")))
                seg = []
            else:
                seg.append(op)
    return out


def main():
    text = check_isa.disassemble(sys.argv[1])
    min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    last = None
    for name, c, v in segments(text, sys.argv[2], min_mfma):
        if name != last:
            print(name)
            last = name
        print("  ", {k: c[k] for k in ("mfma", "valu", "trans", "salu", "ds_read", "ds_write", "vmem_load", "waitcnt", "acc_mov") if c.get(k)})
        print("     ", v.most_common(12))


if __name__ == "__main__":
    main()
