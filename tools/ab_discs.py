"""A/B of library builds on BASELINE configs[3] (362 disc positions share 1e7 rays) and on a 2e6-ray headline launch (where the end of
the launch weighs most):  python tools/ab_discs.py libA.so libB.so ...   (GPU box; each library in its own child process)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("ISX_AB_CHILD"):
    sys.path.insert(0, ROOT)
    import math, zlib
    import numpy as np
    import altair_raytracing_amd as isx
    isx.load(); isx.init(0)
    discs = []
    for th in np.arange(-45.0, 45.0 + 1e-9, 0.5):
        for ph in (0.0, 180.0):
            t, p = math.radians(th), math.radians(ph)
            x, y, z = 200 * math.sin(t) * math.cos(p), 200 * math.sin(t) * math.sin(p), -200 * math.cos(t)
            rot = -math.atan2(math.sqrt(x * x + y * y), -100 - z)
            discs.append([x, y, z, math.sin(rot), 0.0, math.cos(rot)])
    discs = np.array(discs)
    c = isx.default_config()
    c.r_out = 105.0; c.reflectance = 1.0; c.roughness_rad = 0.0; c.max_points = 10000; c.box_half = 200.0
    c.src[2] = -80.0
    isx.disc_sweep(c, discs, 5.0, 0.1, 100000, 7)
    best = None
    for _ in range(4):
        h, st = isx.disc_sweep(c, discs, 5.0, 0.1, 10_000_000, 7)
        k = isx.last_kernel_ms()
        if best is None or st.t_kernel_ms < best[0]: best = (st.t_kernel_ms, k[1], k[2])
    out = {"discs_ms": round(best[0], 3), "discs_trace_ms": round(best[1], 3), "discs_bin_ms": round(best[2], 3), "discs_Mrays_s": round(1e4 / best[0], 1),
           "discs_crc": zlib.crc32(h.tobytes())}
    hc = isx.default_config()
    b2 = None
    for _ in range(4):
        h, st = isx.fluxmap(hc, 2_000_000, 5)
        k = isx.last_kernel_ms()
        if b2 is None or st.t_kernel_ms < b2[0]: b2 = (st.t_kernel_ms, k[1], k[2])
    out.update({"head2e6_ms": round(b2[0], 3), "head2e6_trace_ms": round(b2[1], 3), "head2e6_crc": zlib.crc32(h.tobytes())})
    print(json.dumps(out))
    sys.exit(0)
for lib in sys.argv[1:]:
    env = dict(os.environ, ISX_AB_CHILD="1", ISX_LIB_PATH=os.path.join(ROOT, lib))
    r = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True, timeout=600)
    print(lib, r.stdout.strip() or r.stderr[-600:], flush=True)
