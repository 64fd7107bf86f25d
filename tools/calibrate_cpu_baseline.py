#!/usr/bin/env python3
"""Calibrate bench.py's `cpu_baseline` (the C/OpenMP oracle PORT of the reference path) against the REAL reference.

Build container only (needs /root/reference, imported on CPU through oracle/ref_shim.py; nothing here travels to the
GPU box except the JSON it writes).  Times BASELINE.json configs[0] -- NS2d 128x128x3, B=4, T=16,
LatentDynamics.predict(x, T, to_x=True), train_stage2_ns2d.py:143-158 -- on the same cores, same weights, same input,
once through the reference's PyTorch CPU path and once through the port, and writes

    profiles/r02_cpu_calibration.json   {reference_traj_steps_per_s, port_traj_steps_per_s, reference_ratio, ...}

bench.py multiplies the port's throughput measured on the GPU box's host by `reference_ratio` to quote a
reference-equivalent CPU baseline.

    python tools/calibrate_cpu_baseline.py [--repeats 3]
"""
import argparse
import json
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--rollout", type=int, default=16)
    a = ap.parse_args()
    import lns_oracle
    import ref_models
    from lns_amd import config, filler
    ncpu = len(os.sched_getaffinity(0))
    torch.set_num_threads(ncpu)
    lns_oracle.set_num_threads(ncpu)
    args = config.preset("ns2d_128")
    B, T = a.batch, a.rollout
    x = filler.normal("xcal", (B, args.in_channels, args.Ly, args.Lx), 3)
    model = ref_models.build_reference_dynamics(args, 1)
    xt = torch.from_numpy(x)
    ref_t = []
    with torch.no_grad():
        model.predict(xt[:2], 1, to_x=True)
        for _ in range(a.repeats):
            t0 = time.perf_counter()
            y_ref = model.predict(xt, T, to_x=True)
            ref_t.append(time.perf_counter() - t0)
    sd = {k: v.numpy() for k, v in model.state_dict().items()}
    orc = lns_oracle.OracleDynamics(args, sd)
    orc.predict(x[:1], 1, to_x=True)
    port_t = []
    for _ in range(a.repeats):
        t0 = time.perf_counter()
        y_port = orc.predict(x, T, to_x=True)
        port_t.append(time.perf_counter() - t0)
    err = float(np.sqrt(((y_port - y_ref.numpy()) ** 2).sum() / (y_ref.numpy() ** 2).sum()))
    ref_v, port_v = B * T / min(ref_t), B * T / min(port_t)
    cpu = ""
    try:
        cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    rec = dict(workload="NS2d 128x128x3, B=%d, T=%d, predict(to_x=True) (BASELINE.json configs[0])" % (B, T),
               cores=ncpu, cpu_model=cpu, torch=torch.__version__, python=platform.python_version(),
               reference_traj_steps_per_s=ref_v, port_traj_steps_per_s=port_v, reference_ratio=ref_v / port_v,
               reference_seconds=ref_t, port_seconds=port_t, port_vs_reference_rel_l2=err,
               note="best of %d; reference = the real LatentDynamics on PyTorch CPU kernels (oneDNN conv), "
                    "port = oracle/lns_oracle.{c,py} (plain C loops + OpenMP)" % a.repeats)
    out = os.path.join(ROOT, "profiles", "r02_cpu_calibration.json")
    with open(out, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
