#!/usr/bin/env python3
"""Per-call time of 5e4- and 5e5-ray isx_fluxmap / isx_trace_rays_detector calls for several library builds (GPU box):
python tools/small_call_ab.py libA.so libB.so ..."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("ISX_AB_CHILD"):
    sys.path.insert(0, ROOT)
    import numpy as np
    import altair_raytracing_amd as isx
    isx.load(); isx.init(0)
    c = isx.default_config()
    det = isx.detector_table(c)[90 * 90 + 45]
    out = {}
    for n in (50_000, 500_000):
        for name, f in (("fluxmap", lambda k: isx.fluxmap(c, n, 1, k * n)[1]), ("detector", lambda k: isx.trace_rays_detector(c, det, c.det_diameter, n, 1, k * n)[1])):
            for k in range(3): f(k)
            t0 = time.perf_counter(); ks = []; kk = None
            for k in range(40):
                ks.append(f(k).t_kernel_ms); kk = isx.last_kernel_ms()
            out[f"{name} {n}"] = {"wall_ms": round((time.perf_counter() - t0) * 1e3 / 40, 4), "kernel_ms": round(float(np.mean(ks)), 4), "last_kinds": [round(x, 4) for x in kk]}
    h, st = isx.fluxmap(c, 300_000, 9)
    import zlib
    out["crc_3e5"] = zlib.crc32(h.tobytes())
    print(json.dumps(out)); sys.exit(0)
for lib in sys.argv[1:]:
    env = dict(os.environ, ISX_AB_CHILD="1", ISX_LIB_PATH=os.path.join(ROOT, lib))
    r = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True, timeout=600)
    print(lib, r.stdout.strip() or r.stderr[-500:], flush=True)
