import os, sys, math, json, zlib
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import altair_raytracing_amd as isx
isx.load(); isx.init(0)
def disc_positions():
    discs = []
    for th in np.arange(-45.0, 45.0 + 1e-9, 0.5):
        for ph in (0.0, 180.0):
            t, p = math.radians(th), math.radians(ph)
            x, y, z = 200 * math.sin(t) * math.cos(p), 200 * math.sin(t) * math.sin(p), -200 * math.cos(t)
            dx, dy, dz = 0 - x, 0 - y, -100 - z
            rot = -math.atan2(math.sqrt(dx * dx + dy * dy), dz)
            discs.append([x, y, z, math.sin(rot), 0.0, math.cos(rot)])
    return np.array(discs)
c = isx.default_config()
c.r_out = 105.0; c.reflectance = 1.0; c.roughness_rad = 0.0; c.max_points = 10000; c.box_half = 200.0; c.src[2] = -80.0
discs = disc_positions()
for pipe, ablock in ((0, 768), (1, 768), (1, 512), (1, 384), (1, 256)):
    isx.set_option("disc_pipeline", pipe); isx.set_option("assist_block", ablock)
    best = None
    for _ in range(3):
        h, st = isx.disc_sweep(c, discs, 5.0, 0.1, 10_000_000, 7)
        k = isx.last_kernel_ms()
        if best is None or st.t_kernel_ms < best[0]: best = (st.t_kernel_ms, k)
    print("pipeline", pipe, "assist_block", ablock, "ms", round(best[0], 3), [round(x, 3) for x in best[1]], "Mrays/s", round(1e7 / best[0] / 1e3, 1), "crc", zlib.crc32(h.tobytes()),
          "sum", int(h.sum()), {k: getattr(st, k) for k in ("launched", "exited", "counted_below_z", "absorbed", "suspended", "bin_increments", "wall_hits")}, flush=True)
