#!/usr/bin/env python3
"""Per-kernel resources of the gfx950 code object in the built library (no GPU needed): VGPR / AGPR / SGPR counts, scratch
(spill) bytes, static LDS -- from the code object's metadata notes.

    python tools/kernel_resources.py [liblns_hip.so] [substring ...]
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def resources(lib_path, arch="gfx950"):
    tmp = tempfile.mkdtemp(prefix="lns_res_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib_path, local)
        subprocess.check_call([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out = {}
        for f in os.listdir(tmp):
            if not f.endswith(arch):
                continue
            notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)], stderr=subprocess.DEVNULL).decode()
            for blk in notes.split("- .agpr_count:")[1:]:
                blk = ".agpr_count:" + blk
                g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, None])[1]
                name = g("name")
                if name:
                    out[name] = dict(vgpr=int(g("vgpr_count") or 0), agpr=int(g("agpr_count") or 0), sgpr=int(g("sgpr_count") or 0),
                                     scratch=int(g("private_segment_fixed_size") or 0), lds=int(g("group_segment_fixed_size") or 0),
                                     vgpr_spill=int(g("vgpr_spill_count") or 0), sgpr_spill=int(g("sgpr_spill_count") or 0))
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    args = sys.argv[1:]
    lib = args.pop(0) if args and args[0].endswith(".so") else os.path.join(ROOT, "lns-latent-neural-pde-solver_amd", "liblns_hip.so")
    res = resources(lib)
    for name in sorted(res):
        if args and not any(a in name for a in args):
            continue
        r = res[name]
        print("%-110s vgpr %3d agpr %3d sgpr %3d scratch %5d B  spills v%d s%d" % (name[:110], r["vgpr"], r["agpr"], r["sgpr"], r["scratch"], r["vgpr_spill"], r["sgpr_spill"]))
