"""Builds profiles/rNN_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of
`python bench.py --serial --steps 1 --warmup 0 --no-cpu-baseline --no-roofline`.

Usage: python tools/make_traffic_json.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
"""
import collections
import csv
import json
import sys

KERNELS = {   # class -> (substring of the kernel name, apply the guide's x2 FETCH_SIZE correction?)
    "conv3x3": ("conv3_bf16x3_kernel<1, 1, false, 2, 2, 9", False),
    "conv3x3_up2_fused": ("conv3_bf16x3_kernel<1, 1, true, 2, 2, 4", False),
    "conv1x1": ("conv1_bf16x3_kernel<true, false>", False),
    "conv1x1_stationary": ("conv1s_bf16x3_kernel", False),
    "fa_sandwich": ("fa_sandwich_f_kernel<2, 2, true, true", True),
    "fa_sandwich_32": ("fa_sandwich_f_kernel<1, 1, true, true", True),
}


def per_kernel(path, counter):
    tot = collections.defaultdict(float)
    cnt = collections.defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        tot[r["Kernel_Name"]] += float(r["Counter_Value"])
        cnt[r["Kernel_Name"]] += 1
    return tot, cnt


def main():
    f_tot, f_cnt = per_kernel(sys.argv[1], "FETCH_SIZE")
    w_tot, w_cnt = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes over `python bench.py --serial "
                   "--steps 1 --warmup 0 --no-cpu-baseline --no-roofline` (MI355X). Counter unit: KB (x1024 = bytes). "
                   "gfx950 correction: FETCH_SIZE reports half of the bytes of 16-byte-per-lane streaming reads; it is "
                   "doubled for kernels whose reads are of that kind (fa_sandwich) and left raw for the convolution "
                   "kernels, whose activation reads are 4 or 8 bytes per lane (uncalibrated).",
           "kernels": {}}
    for cls, (sub, x2) in KERNELS.items():
        names = [n for n in f_tot if sub in n]
        if not names:
            continue
        n = names[0]
        launches = f_cnt[n]
        fetch = f_tot[n] * 1024.0 / launches * (2.0 if x2 else 1.0)
        write = w_tot.get(n, 0.0) * 1024.0 / max(1, w_cnt.get(n, 1))
        out["kernels"][cls] = {"kernel": n, "launches": launches, "fetch_bytes_per_launch": fetch,
                               "write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write,
                               "fetch_x2_correction": x2}
    # whole single-stream rollout (all kernels): raw counters, plus the x2 FETCH_SIZE correction for the kernels whose reads are
    # 16-byte-per-lane streaming reads (the ones flagged above)
    corr = sum(f_tot[n] * 1024.0 for sub, x2 in KERNELS.values() if x2 for n in f_tot if sub in n)
    out["rollout_total"] = {"fetch_bytes_raw": sum(f_tot.values()) * 1024.0, "fetch_x2_correction_bytes": corr,
                            "write_bytes": sum(w_tot.values()) * 1024.0,
                            "hbm_bytes": sum(f_tot.values()) * 1024.0 + corr + sum(w_tot.values()) * 1024.0,
                            "workload": "python bench.py --serial --steps 1 --warmup 0 (NS2d 128x128x3, B=64, T=64: ONE single-stream rollout incl. its encode)",
                            "top_fetch_raw": {n.split("(")[0]: v * 1024.0 for n, v in sorted(f_tot.items(), key=lambda kv: -kv[1])[:8]},
                            "top_write": {n.split("(")[0]: v * 1024.0 for n, v in sorted(w_tot.items(), key=lambda kv: -kv[1])[:8]}}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out["kernels"], indent=1))


if __name__ == "__main__":
    main()
