import sys, os, time, json
sys.path.insert(0, os.getcwd())
import numpy as np
import altair_raytracing_amd as isx
isx.load(); isx.init(0)
c = isx.default_config()
out = {}
for n in (50_000, 200_000):
    for pipe, grids in ((0, (0, 16, 32, 49, 64, 98, 128, 196, 256)),):
        isx.set_option("pipeline", pipe)
        for g in grids:
            isx.set_option("grid_blocks", g)
            isx.fluxmap(c, n, 1)
            t0 = time.perf_counter(); ks = [isx.fluxmap(c, n, 1, k * n)[1].t_kernel_ms for k in range(10)]
            out[f"n={n} fused grid={g}"] = {"wall_ms": (time.perf_counter() - t0) * 100, "kernel_ms": float(np.mean(ks))}
print(json.dumps(out, indent=1))
