#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (counter_collection.csv + kernel_trace.csv) into one JSON record per kernel:
MFMA-pipe utilisation, LDS-array utilisation, bank-conflict share, wait breakdown.

    python tools/pmc_summary.py out.json dir1 [dir2 ...]        (one directory per --pmc pass)

Derivations (MI355X_MICROARCH.md): SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD (32 per v_mfma_*_32x32x16), so
mfma_util = that / (4 SIMDs x 256 CUs x kernel cycles); SQ_LDS_IDX_ACTIVE counts LDS-array cycles per CU, so
lds_util = that / (256 CUs x kernel cycles); kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs (the effective clock follows as
cycles / duration).  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycle counts: only their ratios are used."""
import csv, glob, json, os, sys, collections

NCU, NSIMD, NXCD = 256, 4, 8


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(lambda: collections.defaultdict(set))
    dur = collections.defaultdict(list)
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if not k.startswith("void lns::"):
                    continue
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                disp[k][r["Counter_Name"]].add((f, r["Dispatch_Id"]))
                if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                    dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    res = {}
    for k, c in agg.items():
        n = {name: max(1, len(s)) for name, s in disp[k].items()}
        per = {name: c[name] / n[name] for name in c}            # per dispatch
        rec = {"dispatches_profiled": max(n.values()), "counters_per_dispatch": {a: round(b, 1) for a, b in per.items()}}
        cyc = per.get("GRBM_GUI_ACTIVE", 0) / NXCD
        if cyc:
            rec["kernel_cycles"] = round(cyc)
            if dur[k]:
                us = sum(dur[k]) / len(dur[k]) / 1e3
                rec["avg_duration_us_profiled"] = round(us, 2)
                rec["effective_clock_ghz"] = round(cyc / (us * 1e3), 3)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in per:
                rec["mfma_util"] = round(per["SQ_VALU_MFMA_BUSY_CYCLES"] / (NSIMD * NCU * cyc), 4)
            if "SQ_LDS_IDX_ACTIVE" in per:
                rec["lds_array_util"] = round(per["SQ_LDS_IDX_ACTIVE"] / (NCU * cyc), 4)
            if "SQ_LDS_BANK_CONFLICT" in per and per.get("SQ_LDS_IDX_ACTIVE"):
                rec["lds_bank_conflict_share"] = round(per["SQ_LDS_BANK_CONFLICT"] / per["SQ_LDS_IDX_ACTIVE"], 4)
        if per.get("SQ_WAVE_CYCLES"):
            w = per["SQ_WAVE_CYCLES"]
            for a, b in (("SQ_WAIT_ANY", "wave_parked_share"), ("SQ_WAIT_INST_ANY", "issue_stall_share"),
                         ("SQ_WAIT_INST_LDS", "lds_issue_stall_share"), ("SQ_ACTIVE_INST_ANY", "issuing_share"),
                         ("SQ_ACTIVE_INST_VALU", "valu_issue_share"), ("SQ_ACTIVE_INST_LDS", "lds_issue_share")):
                if a in per:
                    rec[b] = round(per[a] / w, 4)
        res[k] = rec
    with open(out, "w") as f:
        json.dump({"note": __doc__.split("\n\n")[1].replace("\n", " "), "kernels": res}, f, indent=1)
    for k, r in res.items():
        print(k[:90], {a: r[a] for a in ("mfma_util", "lds_array_util", "lds_bank_conflict_share", "wave_parked_share",
                                         "issue_stall_share", "valu_issue_share", "effective_clock_ghz") if a in r})


if __name__ == "__main__":
    main()
