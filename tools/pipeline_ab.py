"""A/B of the fused flux kernel vs the two-kernel pipeline (GPU box):  python tools/pipeline_ab.py [flux|chord|brdf]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import altair_raytracing_amd as isx
isx.load(); isx.init(0)
which = sys.argv[1] if len(sys.argv) > 1 else "flux"
c = isx.default_config()
if which == "chord":
    c.trace_mode = 1
elif which == "brdf":
    c.source_model = isx.SOURCE_BRDF; c.brdf[0], c.brdf[1], c.brdf[2] = 0.3, 0.4, 0.6
    c.roughness_rad = 0.5; c.reflectance = 1.0; c.max_points = 10000; c.box_half = 200.0
n = 50_000_000
print("configuration:", which, flush=True)
res = {}
for mode in (0, 1):
    isx.set_option("pipeline", mode)
    isx.fluxmap(c, 100000, 1)
    ts = []
    for _ in range(4):
        h, st = isx.fluxmap(c, n, 5)
        ts.append(st.t_kernel_ms)
    res[mode] = (min(ts), h, st)
    print(f"pipeline={mode}: {min(ts):.2f} ms = {n/min(ts)/1e3:.1f} Mrays/s  (all: {[round(t,2) for t in ts]}) increments {st.bin_increments} counted {st.counted_below_z} last_kernel_ms {[round(x,2) for x in isx.last_kernel_ms()]}", flush=True)
print("histograms equal:", np.array_equal(res[0][1], res[1][1]), "census equal:", all(getattr(res[0][2], k) == getattr(res[1][2], k) for k in ("launched","exited","counted_below_z","absorbed","suspended","bin_increments","wall_hits")))
for bm in (2,):
    isx.set_option("bin_mode", bm)
    for mode in (0, 1):
        isx.set_option("pipeline", mode)
        t = min(isx.fluxmap(c, n, 5)[1].t_kernel_ms for _ in range(3))
        print(f"bin_mode={bm} (trace only) pipeline={mode}: {t:.2f} ms")
isx.set_option("bin_mode", 1)
