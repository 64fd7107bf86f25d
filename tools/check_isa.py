#!/usr/bin/env python3
"""Disassemble the gfx950 code object of a HIP shared library and count / forbid instructions.

    python tools/check_isa.py liblns_hip.so --forbid 'v_pk_(fma|mul|add)_f32'      # exit 1 if any match
    python tools/check_isa.py liblns_hip.so --count v_mfma_f32_32x32x16_f16 --count 's_waitcnt lgkmcnt'

Used by csrc/Makefile (the link step fails if a packed-fp32 arithmetic instruction survives in the device code:
the `-packed-fp32-ops` feature flag is part of the product's correctness, DESIGN.md "co-residency") and by
tests/test_abi_cpu.py.  Needs only llvm-objdump from the ROCm toolchain (no GPU).
"""
import argparse
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")


def disassemble(lib_path, arch="gfx950"):
    """Text of `llvm-objdump -d` over the `arch` code object bundled in lib_path."""
    tmp = tempfile.mkdtemp(prefix="lns_isa_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib_path, local)                       # --offloading extracts next to its input
        subprocess.check_call([OBJDUMP, "--offloading", local], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        objs = [f for f in os.listdir(tmp) if f.endswith(arch)]
        if not objs:
            raise RuntimeError("no %s code object in %s" % (arch, lib_path))
        text = []
        for f in objs:
            text.append(subprocess.check_output([OBJDUMP, "-d", os.path.join(tmp, f)], stderr=subprocess.DEVNULL).decode())
        return "\n".join(text)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def count(text, pattern):
    return len(re.findall(pattern, text))


def kernels_with(text, pattern):
    """{kernel symbol: matches} for the functions whose body matches `pattern`."""
    out, cur = {}, None
    rx = re.compile(pattern)
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = m.group(1)
            continue
        if cur and rx.search(line):
            out[cur] = out.get(cur, 0) + 1
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("lib")
    ap.add_argument("--arch", default="gfx950")
    ap.add_argument("--forbid", action="append", default=[])
    ap.add_argument("--count", action="append", default=[])
    a = ap.parse_args()
    text = disassemble(a.lib, a.arch)
    rc = 0
    for p in a.count:
        print("%-40s %d" % (p, count(text, p)))
    for p in a.forbid:
        hits = kernels_with(text, p)
        if hits:
            rc = 1
            print("check_isa: FORBIDDEN instruction /%s/ in the %s code object of %s:" % (p, a.arch, a.lib), file=sys.stderr)
            for k, n in sorted(hits.items(), key=lambda kv: -kv[1])[:10]:
                print("    %6d  %s" % (n, k), file=sys.stderr)
    return rc


if __name__ == "__main__":
    sys.exit(main())
