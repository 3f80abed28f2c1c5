"""Trace / binning kernel times of the flux-map pipeline for workgroup shapes of the trace kernel (options trace_block,
trace_blocks_per_cu); GPU box:  python tools/shape_sweep.py [flux|chord|brdf]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
print(f"== {which}: total ms (trace + bin) by trace_block x trace_blocks_per_cu", flush=True)
for block in (512,):
    for bpc in (8, 12, 16, 24):
        if block * bpc > 6144 or block * bpc < 1024:
            continue
        isx.set_option("trace_block", block); isx.set_option("trace_blocks_per_cu", bpc)
        isx.fluxmap(c, 100000, 1)
        best = (1e9, 0, 0)
        for _ in range(3):
            h, st = isx.fluxmap(c, n, 5)
            k = isx.last_kernel_ms()
            best = min(best, (st.t_kernel_ms, k[1], k[2]))
        print(f"block {block:5d} x {bpc:2d} per CU ({block * bpc // 256:2d} waves/SIMD asked): {best[0]:7.2f} = {best[1]:6.2f} + {best[2]:6.2f}", flush=True)
isx.set_option("trace_block", 512); isx.set_option("trace_blocks_per_cu", 0)
