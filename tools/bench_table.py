#!/usr/bin/env python3
"""Rewrites the round-3 bench table of DESIGN.md (section 6d) from profiles/r03_*: python tools/bench_table.py"""
import json,csv
rows=[]
for name,f in [("NS2d 128x128x3, B=64, T=64 (the headline)","r03_bench_line.json"),("NS2d T=256","r03_bench_line_ns2d_T256.json"),("SW 96x192x5, B=64, T=64","r03_bench_line_sw_96x192x5.json"),("two-phase conditional, B=32, T=128","r03_bench_line_twophase_cond.json")]:
    d=json.load(open('profiles/'+f)); r=d['roofline']
    rows.append("| {} | {:.1f}k | {:.1f} | {:.3f} ({:.0f} TFLOP/s) | `profiles/{}` |".format(name, d['value']/1e3, d['ms_per_step'], r['frac'], r['achieved'], f))
d=json.load(open('profiles/r03_bench_line.json'))
def tot(f, subs=None):
    return sum(float(r["TotalDurationNs"]) for r in csv.DictReader(open(f)) if subs is None or any(s in r["Name"] for s in subs))/1e6
def row(f, sub):
    for r in csv.DictReader(open(f)):
        if sub in r["Name"]: return int(r["Calls"]), float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3
n='profiles/r03_serial_kernel_stats.csv'; o='profiles/r02_serial_kernel_stats.csv'
c9=row(n,"conv3_bf16x3_kernel<1, 1, false, 2, 2, 9>"); fu=row(n,"conv3_bf16x3_kernel<1, 1, true, 2, 2, 4>"); u4=row(n,"conv3_bf16x3_kernel<1, 1, false, 2, 2, 4>")
o9=row(o,"conv3_bf16x3_kernel<1, 1, false, 2, 2>"); ofu=row(o,"conv3_bf16x3_kernel<1, 1, true, 2, 2>")
txt='''**Bench table, end of round 3** (one MI355X, `tools/final_profiles.sh`, all lines from one box and one build; round 2's lines
for comparison: 32.0k / 32.6k / 20.3k / 45.2k):

| workload | trajectory-steps/s | ms per rollout | 3x3 class: fraction of the f16x2 roofline | line |
|---|---|---|---|---|
'''+"\n".join(rows)+'''

The headline measured 33.7k - 35.5k across the boxes of the round with the same build (this set comes from a middling one; config 4\nmeasured 69.4 - 76.6 ms) (the guide's rule 24: devices differ
by several percent), which is why every change above is quoted as a same-box A/B: phase form +6 % (34.3k vs 32.3k),
instruction trims +2.1 % (34.87k vs 34.15k), config 4's ragged-tile statistics +15 % (53.8k vs 46.9k) and 32-cout tiles +4 % (55.0k vs 52.8k).  Per kernel
(`profiles/r03_serial_kernel_stats.csv` vs `r02_serial_kernel_stats.csv`, 10 single-stream rollouts): 3x3 class {:.0f} -> {:.0f} ms
(nine-tap kernel {:.0f} -> {:.0f} ms over {} -> {} launches, fused final 128^2 layer {:.0f} -> {:.0f} us, the two `UpSampleBlock` convs
now four-tap launches of {:.0f} us), 1x1 class {:.0f} -> {:.0f} ms, all kernels {:.0f} -> {:.0f} ms.  Exact-fp32-MFMA engine on the same
box: {:.2f}x slower (`strict_fp32` record of the line); CPU port on the box's 16 threads {:.1f} trajectory-steps/s = {:.1f}
reference-equivalent (x{:.0f}).

'''.format(tot(o,["conv3_bf16x3"]), tot(n,["conv3_bf16x3"]), o9[1], c9[1], o9[0], c9[0], ofu[2], fu[2], u4[2], tot(o,["conv1_bf16x3","conv1s_"]), tot(n,["conv1_bf16x3","conv1s_"]), tot(o), tot(n),
           d['strict_fp32']['ratio_default_over_strict'], d['cpu_baseline']['value'], d['cpu_baseline']['reference_equivalent'], d['value']/d['cpu_baseline']['reference_equivalent'])
p='DESIGN.md'
s=open(p).read()
i=s.index("**Bench table, end of round 3**"); j=s.index("## 6c. Where the 3x3 kernel's time goes")
s=s[:i]+txt+s[j:]
open(p,'w').write(s)
print(txt)
