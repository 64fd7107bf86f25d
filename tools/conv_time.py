#!/usr/bin/env python3
"""Single conv layers through lns_op_conv2d, NREP launches each, for a rocprofv3 kernel trace (same-box A/B of library
builds via LNS_HIP_LIB):

    rocprofv3 --kernel-trace --output-format csv -d out -- python3 tools/conv_time.py [case ...]
    python tools/conv_time.py --parse out [case ...]      -> {case: {us (min over launches), tflops}}
"""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
cases = {
    "dec13": dict(B=64, Cin=64, Cout=64, H=64, W=64, k=3, up=(128, 128), ss=True, act=1),
    "c64": dict(B=64, Cin=64, Cout=64, H=64, W=64, k=3, ss=True, act=1),
    "c64_128": dict(B=64, Cin=128, Cout=128, H=64, W=64, k=3, ss=True, act=1),
    "c32": dict(B=64, Cin=128, Cout=128, H=32, W=32, k=3, ss=True, act=1),
    "c32_256": dict(B=64, Cin=256, Cout=128, H=32, W=32, k=3, ss=True, act=1),
    "lat": dict(B=64, Cin=128, Cout=128, H=16, W=16, k=3, ss=True, act=0),
    "f64": dict(B=64, Cin=8, Cout=64, H=64, W=64, k=3, ss=True, act=1),          # one stage: the fixed cost per block
    "f64_16": dict(B=64, Cin=16, Cout=64, H=64, W=64, k=3, ss=True, act=1),
    "flat": dict(B=64, Cin=8, Cout=128, H=16, W=16, k=3, ss=True, act=0),
    "flat_16": dict(B=64, Cin=16, Cout=128, H=16, W=16, k=3, ss=True, act=0),
    "f1blk": dict(B=1, Cin=8, Cout=64, H=16, W=8, k=3, ss=True, act=1),            # a single block
    "k1_lat": dict(B=64, Cin=128, Cout=128, H=16, W=16, k=1, ss=True, act=0),
    "k1_lat_up": dict(B=64, Cin=128, Cout=512, H=16, W=16, k=1, ss=True, act=0),
    "k1_lat_dn": dict(B=64, Cin=512, Cout=128, H=16, W=16, k=1, ss=True, act=0),
    "k1_inproj": dict(B=64, Cin=64, Cout=512, H=64, W=64, k=1, ss=True, act=0, v=8),    # input-stationary form
    "k1_toout": dict(B=64, Cin=512, Cout=64, H=64, W=64, k=1, ss=True, act=0),
    "k1_128": dict(B=64, Cin=64, Cout=64, H=128, W=128, k=1, ss=True, act=1),
    "k1_toout_gelu": dict(B=64, Cin=512, Cout=64, H=64, W=64, k=1, ss=True, act=0, act_out=2),
    "k1_toout_gelu_res": dict(B=64, Cin=512, Cout=64, H=64, W=64, k=1, ss=True, act=0, act_out=2, res=True),
    "lat_d4": dict(B=64, Cin=128, Cout=128, H=16, W=16, k=3, ss=True, act=0, dil=4),
    "lat_b256": dict(B=256, Cin=128, Cout=128, H=16, W=16, k=3, ss=True, act=0),
    "lat_d2": dict(B=64, Cin=128, Cout=128, H=16, W=16, k=3, ss=True, act=0, dil=2),
    "up32": dict(B=64, Cin=128, Cout=128, H=16, W=16, k=3, up=(32, 32), ss=False, act=0),
    "up64": dict(B=64, Cin=64, Cout=64, H=32, W=32, k=3, up=(64, 64), ss=False, act=0),          # the 64-channel UpSampleBlock conv
    "c32_64": dict(B=64, Cin=128, Cout=64, H=32, W=32, k=3, ss=True, act=1),
    "lat_tp": dict(B=32, Cin=128, Cout=128, H=7, W=15, k=3, ss=True, act=0),        # config 4's latent plane: one ragged tile
    "lat_tp_d4": dict(B=32, Cin=128, Cout=128, H=7, W=15, k=3, ss=True, act=0, dil=4),
}
NREP = 5
which = [a for a in sys.argv[1:] if not a.startswith("--") and a in cases] or list(cases)
if "--parse" in sys.argv:
    import csv, glob
    d = sys.argv[sys.argv.index("--parse") + 1]
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows = [r for r in rows if "conv" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    assert len(rows) == NREP * len(which), (len(rows), which)
    out = {}
    for i, name in enumerate(which):
        c = cases[name]
        Hv, Wv = c.get("up", (c["H"], c["W"]))
        us = min((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[i * NREP:(i + 1) * NREP])
        gflop = 2.0 * c["B"] * Hv * Wv * c["Cout"] * c["Cin"] * c["k"] ** 2 / 1e9
        out[name] = (round(us, 1), round(gflop / us / 1e3, 1))
    print(json.dumps(out))
    sys.exit(0)
import numpy as np, torch
from lns_amd import _lib
import gpu_checks as gc
L = _lib.lib()
VARIANT = int(os.environ.get("CONV_VARIANT", "11"))      # 11 = f16x2 3x3 kernel (lns_kernels.h ConvVariant)
# CONV_LAYOUT=256 (0x100): x is OCT8, 512 (0x200): y + residual are, 768: both (timing only: the buffers keep their sizes)
LAYOUT = int(os.environ.get("CONV_LAYOUT", "0"))
out = {}
for name in which:
    c = cases[name]
    B, Cin, Cout, H, W, k = c["B"], c["Cin"], c["Cout"], c["H"], c["W"], c["k"]
    Hv, Wv = c.get("up", (H, W))
    torch.manual_seed(0)
    x = torch.randn(B, Cin, H, W, device="cuda")
    w = (np.random.RandomState(0).randn(Cout, Cin, k, k) / np.sqrt(Cin * k * k)).astype(np.float32)
    bias = np.zeros(Cout, np.float32)
    ss = torch.stack([1 + 0.1 * torch.randn(B, Cin), 0.1 * torch.randn(B, Cin)], -1).cuda().contiguous()
    dil = c.get("dil", 1)
    p = dil * (k - 1) // 2
    y = torch.empty(B, Cout, Hv, Wv, device="cuda")
    res = torch.randn(B, Cout, Hv, Wv, device="cuda") if c.get("res") else None
    def run():
        rc = L.lns_op_conv2d(x.data_ptr(), B, Cin, H, W, Hv, Wv, gc._hp(w), gc._hp(bias), Cout, k, 1, dil, p, p, p, p, 1, 1,
                             ss.data_ptr() if c.get("ss", True) else None, c["act"], c.get("act_out", 0), res.data_ptr() if res is not None else None, None, y.data_ptr(), c.get("v", VARIANT if k == 3 else 7) | (LAYOUT if (k == 3 and Cin % 8 == 0 and Cout % 8 == 0) else 0), None, None)
        assert rc == 0
    for _ in range(NREP):
        run()
    torch.cuda.synchronize()
print("ORDER " + " ".join(which) + " NREP %d" % NREP)
