#!/usr/bin/env python3
"""Which arithmetic change moved the long-horizon distance to the reference's fp64 run?  

One child process per arithmetic variant of the engine (the arithmetic is fixed when the library packs its weights),
each running the long-horizon golden rollouts and reporting the decoded rel-L2 to the REAL reference's fp64 run at the
stored steps, beside the fixture's ensemble of the reference's own fp32 runs (min / median / max over 10 members):

    default        shipped library
    oneacc0        1x1 kernels with two accumulators (build/variants/oneacc0: -DLNS_CONV1_ONEACC=0)
    fixedscale16   activation scale forced to round 1's constant x16 (build/variants/fixedscale16: -DLNS_FIXED_ACT_SCALE=16)
    bf16x3         3x3 convolutions on the three-term bf16 split (LNS_CONV3_SPLIT=bf16x3)
    strict_fp32    every contraction on v_mfma_f32_32x32x2_f32

    python tools/drift_attribution.py [case ...]        -> table on stdout, JSON in gpurun_out/r3_drift_attribution.json
Build the two variant libraries first (CPU container): tools/build_variant.sh oneacc0 -DLNS_CONV1_ONEACC=0 ; ...
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ["ns2d_128_T256", "sw_96x192x5_T64", "twophase_cond_T128"]


def variants():
    v = [("default", {})]
    for name in ("oneacc0", "fixedscale16"):
        lib = os.path.join(ROOT, "build", "variants", name, "pkg", "liblns_hip.so")
        if os.path.exists(lib):
            v.append((name, {"LNS_HIP_LIB": lib}))
    v.append(("bf16x3", {"LNS_CONV3_SPLIT": "bf16x3"}))
    v.append(("strict_fp32", {"LNS_CONV_FP32_MFMA": "1", "LNS_CONV1_FP32_MFMA": "1", "LNS_FA_SANDWICH_FP32": "1", "LNS_ATTN_FP32": "1"}))
    return v


def child(cases):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import torch
    import gpu_checks as gc
    from helpers import load_golden, case_args, case_inputs, rel_l2
    out = {}
    for case in cases:
        meta, g = load_golden(case)
        args = case_args(meta)
        model, _ = gc.build_models(args, meta["weight_seed"])
        x, param = case_inputs(meta, args)
        xd = torch.from_numpy(x).cuda()
        extra = (torch.from_numpy(param).cuda(),) if param is not None else ()
        dec = model.predict(xd, meta["T"], *extra, to_x=True).cpu().numpy()
        sub = meta["sub"]
        out[case] = {str(s): rel_l2(dec[:, s - 1][..., ::sub, ::sub], g["dec_f64"][:, i]) for i, s in enumerate(meta["steps"])}
    print("@@" + json.dumps(out))


def main(cases):
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import load_golden
    res = {}
    for name, env in variants():
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + cases, env=dict(os.environ, **env),
                           capture_output=True, text=True, timeout=900)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("@@")]
        if not line:
            print(name, "FAILED", p.stderr[-500:])
            continue
        res[name] = json.loads(line[0][2:])
    rec = {"variants": res, "ensemble": {}}
    for case in cases:
        meta, g = load_golden(case)
        ens = g["ref_ens_err_sub"]
        print("\n%s: decoded rel-L2 to the reference's fp64 run" % case)
        print("  %-14s" % "step" + "".join("%11d" % s for s in meta["steps"]))
        for lab, row in (("ens min", ens.min(0)), ("ens median", np.median(ens, 0)), ("ens max", ens.max(0))):
            print("  %-14s" % ("ref " + lab) + "".join("%11.2e" % v for v in row))
        rec["ensemble"][case] = {"steps": meta["steps"], "min": ens.min(0).tolist(), "median": np.median(ens, 0).tolist(),
                                 "max": ens.max(0).tolist(), "members": [str(d) for d in g["ref_ens_desc"]]}
        for name in res:
            print("  %-14s" % name + "".join("%11.2e" % res[name][case][str(s)] for s in meta["steps"]))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r3_drift_attribution.json"), "w") as f:
        json.dump(rec, f, indent=1)


if __name__ == "__main__":
    if sys.argv[1:2] == ["--child"]:
        child(sys.argv[2:])
    else:
        main(sys.argv[1:] or CASES)
