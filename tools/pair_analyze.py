"""Analyse a dump of tools/pair_stress.py (side B = 1x1 conv, 64->64): which input element was wrong, and what it became."""
import sys
import numpy as np
d = sys.argv[1]
B, C, HW = 32, 64, 64 * 64
out = np.fromfile(d + "/out.bin", np.float32).reshape(B, C, HW)
ref = np.fromfile(d + "/ref.bin", np.float32).reshape(B, C, HW)
x = np.fromfile(d + "/x.bin", np.float32).reshape(B, C, HW)
w = np.fromfile(d + "/w.bin", np.float32).reshape(C, C).astype(np.float64)
ss = np.fromfile(d + "/ss.bin", np.float32).reshape(B, C, 2)
bad = np.argwhere((out != ref).any(1))
print("bad (sample, pixel) pairs:", len(bad), "samples", np.unique(bad[:, 0]), "pixel range", bad[:, 1].min(), bad[:, 1].max())
print("pixels mod 4:", np.unique(bad[:, 1] % 4, return_counts=True), " pixels mod 256 range:", (bad[:, 1] % 256).min(), (bad[:, 1] % 256).max())
winv = np.linalg.inv(w)
def swish(v): return v / (1 + np.exp(-v))
for (b, p) in bad[:8]:
    dout = (out[b, :, p].astype(np.float64) - ref[b, :, p])
    dx = winv @ dout
    cs = np.argsort(-np.abs(dx))[:3]
    v = x[b, :, p].astype(np.float64) * ss[b, :, 0] + ss[b, :, 1]
    t = swish(v)
    print("sample %d pixel %d (tile-local %d): dominant wrong channels %s dx %s" % (b, p, p % 256, cs.tolist(), np.round(dx[cs], 6).tolist()))
    c = cs[0]
    got = t[c] + dx[c]
    print("    channel %d: expected t=%.6f  became %.6f ; raw x=%.6f fma v=%.6f ; sigmoid=%.6f" % (c, t[c], got, x[b, c, p], v[c], t[c] / v[c] if v[c] else 0))
    # which (scale, shift) was applied?  solve from this pixel and the next wrong pixel of the same channel
    nxt = [(bb, pp) for (bb, pp) in bad if bb == b and pp > p]
    if nxt:
        p2 = nxt[0][1]
        dout2 = (out[b, :, p2].astype(np.float64) - ref[b, :, p2])
        dx2 = winv @ dout2
        v2 = x[b, c, p2].astype(np.float64) * ss[b, c, 0] + ss[b, c, 1]
        got2 = swish(v2) + dx2[c]
        def inv_swish(y):
            lo, hi = -0.27, 10.0
            for _ in range(80):
                mid = 0.5 * (lo + hi)
                if swish(mid) < y: lo = mid
                else: hi = mid
            return 0.5 * (lo + hi)
        va, vb = inv_swish(got), inv_swish(got2)
        xa, xb_ = float(x[b, c, p]), float(x[b, c, p2])
        s_est = (va - vb) / (xa - xb_)
        t_est = va - s_est * xa
        print("    applied scale %.5f shift %.5f ; correct scale %.5f shift %.5f" % (s_est, t_est, ss[b, c, 0], ss[b, c, 1]))
        cand = np.argwhere(np.abs(ss[b, :, 1] - t_est) < 3e-4).flatten().tolist()
        cand_s = np.argwhere(np.abs(ss[b, :, 0] - t_est) < 3e-4).flatten().tolist()
        print("    channels of this sample whose SHIFT equals the applied shift:", cand, " whose SCALE equals it:", cand_s)
        allb = np.argwhere(np.abs(ss[:, :, 1] - t_est) < 1e-4)
        print("    (any sample) entries with that shift:", allb[:6].tolist())
    # candidates: same channel other pixels, same pixel other channels (transformed)
    tt = swish(x[b].astype(np.float64) * ss[b, :, 0:1] + ss[b, :, 1:2])   # [C, HW]
    near = np.argwhere(np.abs(tt - got) < 2e-6)
    print("    transformed elements of this sample equal to the wrong value:", near[:6].tolist())
    raw = np.argwhere(np.abs(x[b].astype(np.float64) - got) < 2e-6)
    print("    raw elements equal:", raw[:4].tolist())
