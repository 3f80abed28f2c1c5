#!/usr/bin/env python3
"""Launch shape of SMALL calls (GPU box): kernel and wall time per call of isx_fluxmap / isx_trace_rays_detector for
n = 5e4 .. 5e6 rays over workgroup sizes (assist_block: tracer waves + one assist wave) and grid sizes.  A small call is
bound by its longest ray (~ln(n)/0.0175 bounces, one after the other), not by throughput: few waves per SIMD, few rays per lane."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import altair_raytracing_amd as isx  # noqa: E402

isx.load(); isx.init(0)
c = isx.default_config()
det = isx.detector_table(c)[90 * 90 + 45]
out = {}


def run(kind, n, reps=12):
    f = (lambda k: isx.fluxmap(c, n, 1, k * n)[1]) if kind == "fluxmap" else (lambda k: isx.trace_rays_detector(c, det, c.det_diameter, n, 1, k * n)[1])
    f(0)
    t0 = time.perf_counter()
    ks = [f(k).t_kernel_ms for k in range(reps)]
    return (time.perf_counter() - t0) * 1e3 / reps, float(np.mean(ks)), isx.last_kernel_ms()


for n in (50_000, 500_000, 5_000_000):
    for kind in ("detector", "fluxmap"):
        rows = []
        isx.set_option("assist_block", 768); isx.set_option("grid_blocks", 0)
        w, k, kk = run(kind, n)
        rows.append({"block": "default", "grid": "default", "wall_ms": round(w, 4), "kernel_ms": round(k, 4), "kinds": [round(x, 4) for x in kk]})
        for blk in (128, 256, 384, 768):
            tr = blk // 64 - 1
            for rays_per_lane in (1, 2, 4, 8, 16):
                g = max(1, min(65535, int(np.ceil(n / (tr * 64 * rays_per_lane)))))
                if g > 256 * 8:
                    continue
                isx.set_option("assist_block", blk); isx.set_option("grid_blocks", g)
                w, k, kk = run(kind, n, reps=6)
                rows.append({"block": blk, "grid": g, "rays_per_lane": rays_per_lane, "wall_ms": round(w, 4), "kernel_ms": round(k, 4),
                             "kinds": [round(x, 4) for x in kk]})
        out[f"{kind} n={n}"] = rows
        print(kind, n, "best", min(rows, key=lambda r: r["kernel_ms"]), file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
