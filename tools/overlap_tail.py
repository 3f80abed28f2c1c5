"""Does a second trace kernel on another stream fill the tail of the first?  Long-lived rays (reflectance 1: ~144 bounces, the
longest of a launch thousands), 1e7 rays, flux-map pipeline with overlap = 0 / 2 / 3 / 4 chunks, trace kernels on one or two
streams.  GPU box: python tools/overlap_tail.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import altair_raytracing_amd as isx
isx.load(); isx.init(0)
c = isx.default_config()
c.r_out = 105.0; c.reflectance = 1.0; c.roughness_rad = 0.0; c.max_points = 10000; c.box_half = 200.0
c.src[2] = -80.0
n = 10_000_000
ref = None
for ov, ts in ((0, 1), (2, 1), (2, 2), (3, 2), (4, 2), (8, 2), (0, 1)):
    isx.set_option("overlap", ov); isx.set_option("overlap_trace_streams", ts)
    isx.fluxmap(c, 200_000, 3)
    best = 1e9
    for _ in range(4):
        h, st = isx.fluxmap(c, n, 7)
        best = min(best, st.t_kernel_ms)
    if ref is None: ref = h
    print(f"overlap {ov} trace streams {ts}: {best:.3f} ms  {n / best / 1e3:.1f} Mrays/s  same {np.array_equal(h, ref)}", flush=True)
