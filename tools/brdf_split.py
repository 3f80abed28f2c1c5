import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import altair_raytracing_amd as isx
isx.load(); isx.init(0)
def cfg():
    c = isx.default_config()
    c.source_model = isx.SOURCE_BRDF; c.brdf[0], c.brdf[1], c.brdf[2] = 0.3, 0.4, 0.6
    c.roughness_rad = 0.5; c.reflectance = 1.0; c.max_points = 10000; c.box_half = 200.0
    return c
n = 20_000_000
out = {}
for bm in (1, 2):
    isx.set_option("bin_mode", bm)
    best = 1e9
    for _ in range(2):
        h, st = isx.fluxmap(cfg(), n, 5)
        best = min(best, st.t_kernel_ms)
    out[f"bin_mode{bm}"] = {"ms": best, "counted": st.counted_below_z, "exited": st.exited, "increments": st.bin_increments, "wall_hits": st.wall_hits}
isx.set_option("bin_mode", 1)
c = isx.default_config()
for bm in (1, 2):
    isx.set_option("bin_mode", bm)
    best = 1e9
    for _ in range(2):
        h, st = isx.fluxmap(c, n, 5)
        best = min(best, st.t_kernel_ms)
    out[f"headline_bin_mode{bm}"] = {"ms": best, "counted": st.counted_below_z, "increments": st.bin_increments, "wall_hits": st.wall_hits}
print(json.dumps(out, indent=1))
