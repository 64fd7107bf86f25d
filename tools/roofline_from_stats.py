#!/usr/bin/env python3
"""Recompute the bench line's `roofline` numbers from a rocprofv3 kernel-stats CSV.

    python tools/roofline_from_stats.py <kernel_stats.csv> <bench_line.json> [rollouts in the profile]

The CSV comes from `rocprofv3 --kernel-trace --stats -- python3 bench.py --serial --steps K --warmup W --no-check ...`
(single stream: no co-running kernels); `rollouts` = K + W + 1 (the roofline pass), default: inferred from the call count of
the FABlock sandwich kernel (two launches per decode step... i.e. from the bench line's launch counts).  For every kernel FORM
of the bench line (`roofline.entries`) the kernels of that form are summed in the CSV and

    frac(form) = algorithmic FLOP of the form per rollout (bench line: algorithmic_tflops x ms) / rocprof time per rollout / peak

with peak = 2516.6 / 3.333 TFLOP/s for the nine-tap f16x2 kernel (`roofline.frac`), and executed_frac = executed MFMA FLOP /
time / 2516.6 TFLOP/s.  Prints both beside the bench line's own (HIP-event) numbers: they must agree to +-0.02.
"""
import csv
import json
import re
import sys

F16 = 2516.6e12
PATTERNS = [   # form -> regex on the demangled kernel name
    ("f16x2 3x3 nine-tap + fused 1x1", r"conv3_bf16x3_kernel<\d+, \d+, true, \d+, 2, 9"),
    ("f16x2 3x3 nine-tap", r"conv3_bf16x3_kernel<\d+, \d+, false, \d+, 2, 9"),
    ("f16x2 3x3 four-tap phase form + fused 1x1", r"conv3_bf16x3_kernel<\d+, \d+, true, \d+, 2, 4"),
    ("f16x2 3x3 four-tap phase form", r"conv3_bf16x3_kernel<\d+, \d+, false, \d+, 2, 4"),
    ("f16x2 1x1 input-stationary", r"conv1s_bf16x3_kernel"),
    ("f16x2 1x1 + fused 1x1", r"conv1_bf16x3_kernel<(true|false), true>"),
    ("f16x2 1x1 streaming", r"conv1_bf16x3_kernel<(true|false), false>"),
    ("thin 1x1 projection (VALU)", r"conv1_thin_kernel"),
    ("fp32 MFMA 3x3", r"conv_mfma_kernel<\d+, \d+, \d+, \d+, 3,"),
    ("fp32 MFMA 1x1", r"conv_mfma_kernel<\d+, \d+, \d+, \d+, 1,"),
    ("f16x2 FABlock sandwich", r"fa_sandwich_f_kernel"),
    ("f16x2 FABlock in_proj + sandwich", r"fa_fused_kernel"),
    ("FABlock input split (VALU)", r"fa_gsplit_kernel"),
    ("f16x2 attention", r"attention_f_kernel"),
]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    line = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    roof = line["roofline"]
    entries = {e["form"]: e for e in roof["entries"]}
    calls = {f: 0 for f, _ in PATTERNS}
    total_ns = {f: 0.0 for f, _ in PATTERNS}
    for r in rows:
        for form, pat in PATTERNS:
            if re.search(pat, r["Name"]):
                calls[form] += int(r["Calls"])
                total_ns[form] += float(r["TotalDurationNs"])
                break
    if len(sys.argv) > 3:
        rollouts = int(sys.argv[3])
    else:       # launches per rollout are in the bench line
        ref = next(f for f in entries if calls.get(f))
        rollouts = round(calls[ref] / entries[ref]["launches"])
    print("rollouts in the profile: %d" % rollouts)
    print("%-46s %9s %9s %10s %10s | %s" % ("form", "launches", "avg us", "ms/rollout", "bench ms", "fractions (rocprof | bench line)"))
    for form, _ in PATTERNS:
        e = entries.get(form)
        if not e or not calls[form]:
            continue
        ms = total_ns[form] / rollouts / 1e6
        alg = (e["algorithmic_tflops"] or 0.0) * 1e12 * e["ms"] * 1e-3                 # FLOP per rollout
        exe = (e["executed_mfma_tflops"] or 0.0) * 1e12 * e["ms"] * 1e-3
        peak = F16 if e["pipe"].startswith("fp16") else (157.3e12 if e["pipe"].startswith("fp32") else None)
        msg = ""
        if peak and exe:
            msg = "executed %.3f | %.3f" % (exe / (ms * 1e-3) / peak, e["executed_frac_of_pipe_peak"])
        if form == "f16x2 3x3 nine-tap":
            frac = alg / (ms * 1e-3) / (F16 / (3.0 * 10.0 / 9.0))
            msg += "   frac %.3f | %.3f   (algorithmic %.1f TFLOP/s of 755)" % (frac, roof["frac"], alg / (ms * 1e-3) / 1e12)
        print("%-46s %9d %9.2f %10.3f %10.3f | %s" % (form, calls[form] // rollouts, total_ns[form] / calls[form] / 1e3, ms, e["ms"], msg))
    pm = roof.get("mfma_busy_pmc")
    if pm:
        print("PMC (own pass, %s): MFMA pipe busy %.3f at an effective clock of %s GHz" % (pm["source"], pm["mfma_busy"], pm.get("effective_clock_ghz")))


if __name__ == "__main__":
    main()
