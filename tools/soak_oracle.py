"""Soak test GPU == oracle on random configurations (every field of isx_config that the path reads: port, wall, box, source,
grid, detector, port plane, surface / source / hit-line / trace models): flux map + census, per-position map (both folds),
exit-dz histogram, end states.  GPU box (the oracle is the checker):  python tools/soak_oracle.py [n] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import altair_raytracing_amd as isx
import oracle as orc
isx.load(); isx.init(0)
NG = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
CENSUS = ("launched", "exited", "counted_below_z", "absorbed", "suspended", "bin_increments", "wall_hits")
bad = 0
for k in range(NG):
    v = dict(theta_max_deg=float(rng.uniform(150, 178)), reflectance=float(rng.choice([0.9, 0.97, 0.99, 1.0])),
             max_points=int(rng.choice([50, 400, 3000])), box_half=float(rng.choice([200.0, 300.0])),
             n_theta=int(rng.integers(1, 120)), n_phi=2 * int(rng.integers(1, 70)),
             det_distance=float(rng.choice([30.0, 100.0, 180.0])), exit_port_z=float(rng.choice([-100.0, -120.0, -99.0])),
             r_in=float(rng.choice([100.1, 60.0])), )
    v["r_out"] = v["r_in"] + float(rng.choice([0.9, 5.0]))
    v["det_diameter"] = float(v["det_distance"] * 2 * rng.choice([0.02, 0.2, 0.5, 0.8, 1.2]))
    mode = k % 6
    if mode == 1: v["source_model"] = 1
    elif mode == 2: v["trace_mode"] = 1
    elif mode == 3: v.update(lambertian=0, roughness_rad=float(rng.choice([0.05, 0.3])), reflectance=0.9)
    elif mode == 4: v["surface_model"] = 1
    elif mode == 5: v["hit_line_mode"] = 1
    src = [float(rng.uniform(-0.6, 0.6) * v["r_in"]), float(rng.uniform(-0.3, 0.3) * v["r_in"]), float(rng.uniform(-0.8, 0.4) * v["r_in"])]
    dr = [float(rng.uniform(1, 6)), float(rng.uniform(-3, 3)), float(rng.uniform(-2, 2))]
    ci, co = isx.default_config(), orc.default_config()
    for c in (ci, co):
        for f, x in v.items():
            setattr(c, f, x)
        for a in range(3):
            c.src[a] = src[a]; c.dir[a] = dr[a]
    n = 20000
    ok = True
    try:
        gh, gst = isx.fluxmap(ci, n, 100 + k, 12345 * k)
    except isx.IsxError as e:
        print(k, "refused:", e, flush=True); continue
    oh, ost = orc.fluxmap(co, n, 100 + k, 12345 * k)
    ok &= np.array_equal(gh, oh) and all(getattr(gst, f) == getattr(ost, f) for f in CENSUS)
    if mode in (0, 2):
        for fold in (1, 2):
            gp, _ = isx.fluxmap_per_position(ci, 7, 300 + k, fold)
            op, _ = orc.fluxmap_per_position(co, 7, 300 + k, fold)
            ok &= np.array_equal(gp, op)
        gd, _ = isx.exit_dz_hist(ci, n, 400 + k)
        od, _ = orc.exit_dz_hist(co, n, 400 + k)
        ok &= np.array_equal(gd, od)
    bad += (not ok)
    if not ok:
        print("MISMATCH", k, mode, v, flush=True)
    if k % 20 == 0:
        print(k, "configurations,", bad, "mismatches; increments", gst.bin_increments, flush=True)
print("done:", NG, "configurations,", bad, "mismatches")
sys.exit(1 if bad else 0)
