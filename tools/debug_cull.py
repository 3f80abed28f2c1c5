import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import altair_raytracing_amd as isx
isx.load(); isx.init(0)
c = isx.default_config()
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
seed = 987654321
isx.set_option("bin_mode", 0); b, sb = isx.fluxmap(c, n, seed)
isx.set_option("bin_mode", 1); k, sk = isx.fluxmap(c, n, seed)
d = k.astype(np.int64) - b.astype(np.int64)
idx = np.argwhere(d != 0)
print("n", n, "differing bins", len(idx), "sum diff", d.sum(), "abs", np.abs(d).sum(), "brute inc", sb.bin_increments, "culled inc", sk.bin_increments)
for i, j in idx[:40]:
    print(i, j, d[i, j])
# narrow down: bisect ray ranges to find offending rays
def diffcount(first, cnt):
    isx.set_option("bin_mode", 0); b, _ = isx.fluxmap(c, cnt, seed, first)
    isx.set_option("bin_mode", 1); k, _ = isx.fluxmap(c, cnt, seed, first)
    return int(np.abs(k.astype(np.int64) - b.astype(np.int64)).sum())
lo, cnt = 0, n
while cnt > 1:
    h = cnt // 2
    if diffcount(lo, h) > 0: cnt = h
    else: lo, cnt = lo + h, cnt - h
print("offending ray", lo, diffcount(lo, 1))
st, npts, lp, dr = isx.trace_endstates(c, 1, seed, lo)
print("endstate", st, npts, lp.tolist(), dr.tolist())
isx.set_option("bin_mode", 0); b, _ = isx.fluxmap(c, 1, seed, lo)
isx.set_option("bin_mode", 1); k, _ = isx.fluxmap(c, 1, seed, lo)
dd = k.astype(np.int64) - b.astype(np.int64)
print("bins differing for this ray:", [(int(i), int(j), int(dd[i, j])) for i, j in np.argwhere(dd != 0)][:20], "hits brute", int(b.sum()), "culled", int(k.sum()))
print("rows hit (brute):", np.nonzero(b.sum(1))[0][[0, -1]].tolist())
