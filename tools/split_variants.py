"""Timing of the headline and BRDF configurations for builds with different ISX_SPLIT_AT (variants/libisx_split<N>.so); run as
   for v in ...; do ISX_LIB_PATH=variants/libisx_split$v.so python tools/split_variants.py; done   (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import altair_raytracing_amd as isx
isx.load(); isx.init(0)
def t(c, n):
    isx.fluxmap(c, 100000, 1)
    return min(isx.fluxmap(c, n, 5)[1].t_kernel_ms for _ in range(3))
c = isx.default_config()
a = t(c, 50_000_000)
c.source_model = 1; c.roughness_rad = 0.5; c.reflectance = 1.0; c.max_points = 10000; c.box_half = 200.0
b = t(c, 20_000_000)
print(os.environ.get("ISX_LIB_PATH", "default"), f"headline 5e7: {a:.2f} ms = {5e4/a:.1f} Mrays/s ; brdf 2e7: {b:.2f} ms = {2e4/b:.1f} Mrays/s", flush=True)
