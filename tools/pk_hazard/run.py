#!/usr/bin/env python3
"""Drives tools/pk_hazard/libpk_hazard.so (see pk_hazard.hip): counts bitwise mismatches between v_pk_fma_f32 (op_sel
operand selection) and scalar v_fma_f32 results of the same wave, alone and while an MFMA-dense kernel co-runs.

    python tools/pk_hazard/run.py [rounds] [launches]      -> one JSON line
"""
import ctypes
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libpk_hazard.so")


def build():
    src = os.path.join(HERE, "pk_hazard.hip")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                               "-Wno-unused-result", src, "-o", LIB])
    return LIB


def run(rounds=20, launches=10, B=16, C=64, HW=64 * 64):
    L = ctypes.CDLL(build())
    L.pk_run.argtypes = [ctypes.c_int] * 6 + [ctypes.POINTER(ctypes.c_ulonglong)] * 2
    out = {}
    for name, agg in (("alone", 0), ("corun", 1)):
        bp, bs = ctypes.c_ulonglong(0), ctypes.c_ulonglong(0)
        rc = L.pk_run(B, C, HW, rounds, launches, agg, ctypes.byref(bp), ctypes.byref(bs))
        if rc != 0:
            raise RuntimeError("pk_run rc=%d" % rc)
        out[name] = {"packed_mismatches": int(bp.value), "scalar_mismatches": int(bs.value),
                     "words_checked": B * C * HW * rounds * launches}
    return out


if __name__ == "__main__":
    r = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    l = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    print(json.dumps(run(r, l)))
