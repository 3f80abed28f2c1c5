#!/usr/bin/env python3
"""Workgroup shape of the lobe / rough-specular trace kernels (GPU box): kernel time of 5e7 rays over assist_block."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import altair_raytracing_amd as isx  # noqa: E402

isx.load(); isx.init(0)
out = {}
for name, setup in (("lobe", lambda q: setattr(q, "surface_model", 1)),
                    ("rough", lambda q: (setattr(q, "lambertian", 0), setattr(q, "roughness_rad", 0.5)))):
    c = isx.default_config()
    setup(c)
    for blk in (0, 384, 512, 576, 640, 768):
        isx.set_option("assist_block", blk)
        best = None
        for _ in range(2):
            _, st = isx.fluxmap(c, 50_000_000, 0x5EED0001)
            k = isx.last_kernel_ms()
            if best is None or k[1] < best[0]:
                best = (k[1], k[2])
        out[f"{name} assist_block={blk}"] = {"trace_ms": round(best[0], 3), "bin_ms": round(best[1], 3)}
        print(name, blk, out[f"{name} assist_block={blk}"], file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
