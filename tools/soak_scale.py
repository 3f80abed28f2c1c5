"""Soak of the cull's absolute slack terms: the whole geometry scaled by 0.01 ... 100 (sphere, box, source, detector sphere and
detector size, port plane), culled == brute through the pipeline and the fused kernels.  GPU box:  python tools/soak_scale.py"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import altair_raytracing_amd as isx
isx.load(); isx.init(0)
rng = np.random.default_rng(12)
bad = ran = 0
for s in (0.01, 0.1, 0.5, 3.0, 10.0, 30.0, 100.0):
    for rep in range(9):
        c = isx.default_config()
        c.r_in *= s; c.r_out *= s; c.box_half *= s
        for a in range(3): c.src[a] *= s
        c.exit_port_z *= s
        c.det_distance = float(100.0 * s * rng.choice([0.5, 1.0, 1.6]))
        c.det_diameter = float(c.det_distance * 2 * rng.choice([0.01, 0.2, 0.6]))
        c.theta_max_deg = float(rng.uniform(155, 176))
        c.n_theta, c.n_phi = int(rng.integers(5, 180)), int(rng.integers(5, 120))
        if rep % 3 == 1: c.source_model = 1
        if rep % 3 == 2: c.trace_mode = 1
        n = 50000
        try:
            isx.set_option("bin_mode", 0); brute, sb = isx.fluxmap(c, n, 60 + rep)
        except isx.IsxError as e:
            isx.set_option("bin_mode", 1); print("scale", s, "refused:", e.status, flush=True); break
        isx.set_option("bin_mode", 1)
        culled, sc = isx.fluxmap(c, n, 60 + rep)
        isx.set_option("pipeline", 0); fused, sf = isx.fluxmap(c, n, 60 + rep); isx.set_option("pipeline", 1)
        ok = np.array_equal(brute, culled) and np.array_equal(brute, fused)
        ran += 1; bad += (not ok)
        if not ok:
            print("MISMATCH scale", s, rep, c.det_diameter, c.det_distance, c.n_theta, c.n_phi, int(brute.sum()), int(culled.sum()), int(fused.sum()), flush=True)
    print("scale", s, "done; increments", int(sc.bin_increments), flush=True)
print("done: ran", ran, "mismatches", bad)
sys.exit(1 if bad else 0)
