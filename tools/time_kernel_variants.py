import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import altair_raytracing_amd as isx
isx.load(); isx.init(0)
n = 10_000_000
def run(name, c, bm):
    isx.set_option("bin_mode", bm)
    isx.fluxmap(c, 100000, 1)
    h, st = isx.fluxmap(c, n, 5)
    print(f"{name:40s} bin_mode={bm} {st.t_kernel_ms:9.2f} ms  hits/ray {st.wall_hits/n:7.2f} exit {st.counted_below_z/n:.3f} susp {st.suspended}", flush=True)
c = isx.default_config()
for bm in (1, 2): run("default (lean)", c, bm)
c = isx.default_config(); c.hit_line_mode = 1
for bm in (1, 2): run("default + compat line (full kernel)", c, bm)
c = isx.default_config(); c.reflectance = 1.0; c.max_points = 10000; c.box_half = 200.0
for bm in (1, 2): run("rho=1 pencil (lean)", c, bm)
c.source_model = 1; c.brdf[0], c.brdf[1], c.brdf[2] = 0.3, 0.4, 0.6; c.roughness_rad = 0.5
for bm in (1, 2): run("rho=1 BRDF source (full)", c, bm)
c.reflectance = 0.99
for bm in (1, 2): run("rho=.99 BRDF source (full)", c, bm)
