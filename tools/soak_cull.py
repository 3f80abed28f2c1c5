"""Soak test of the binning pre-selection: random geometries (detector radius up to several sphere radii, grids from 1x1 to 36 000
bins, port angles, port plane heights, box sizes; pencil / BRDF source, explicit / chord trace, rough specular surface), culled ==
brute force through the two-kernel pipeline AND the fused kernels.  GPU box:  python tools/soak_cull.py [n_geometries] [seed]"""
import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import altair_raytracing_amd as isx
isx.load(); isx.init(0)
NG = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 777)
bad = 0; tot = 0
for k in range(NG):
    c = isx.default_config()
    c.theta_max_deg = float(rng.uniform(150, 178))
    c.reflectance = float(rng.choice([0.9, 0.97, 0.99, 1.0])); c.max_points = 3000
    c.src[0], c.src[1], c.src[2] = float(rng.uniform(-70, 70)), float(rng.uniform(-30, 30)), float(rng.uniform(-85, 40))
    c.dir[0], c.dir[1], c.dir[2] = float(rng.uniform(1, 6)), float(rng.uniform(-3, 3)), float(rng.uniform(-2, 2))
    c.n_theta, c.n_phi = int(rng.integers(1, 200)), int(rng.integers(1, 180))
    if c.n_theta * c.n_phi > 36000: c.n_phi = 36000 // c.n_theta
    c.det_distance = float(rng.choice([30.0, 60.0, 100.0, 180.0]))
    c.det_diameter = float(c.det_distance * 2 * rng.choice([0.005, 0.05, 0.2, 0.3, 0.45, 0.55, 0.67, 0.8, 0.95, 1.05, 1.5, 3.0]))
    c.exit_port_z = float(rng.choice([-100.0, -120.0, -99.0]))
    c.box_half = float(rng.choice([200.0, 300.0]))
    mode = k % 4
    if mode == 1: c.source_model = 1
    elif mode == 2: c.trace_mode = 1
    elif mode == 3: c.lambertian = 0; c.roughness_rad = float(rng.choice([0.05, 0.2, 0.5])); c.reflectance = 0.9
    n = 100000
    isx.set_option("bin_mode", 0)
    try:
        brute, sb = isx.fluxmap(c, n, 5000 + k)
    except isx.IsxError as e:
        isx.set_option("bin_mode", 1); continue
    isx.set_option("bin_mode", 1)
    culled, sc = isx.fluxmap(c, n, 5000 + k)
    isx.set_option("pipeline", 0)
    fused, sf = isx.fluxmap(c, n, 5000 + k)
    isx.set_option("pipeline", 1)
    ok = np.array_equal(brute, culled) and np.array_equal(brute, fused)
    tot += 1; bad += (not ok)
    if not ok: print("MISMATCH", k, [getattr(c, f) for f in ("theta_max_deg", "n_theta", "n_phi", "det_diameter", "det_distance", "exit_port_z", "source_model", "trace_mode")], flush=True)
    if k % 50 == 0: print(k, "ok so far", tot - bad, "of", tot, "increments", sc.bin_increments, flush=True)
print("done:", tot, "geometries,", bad, "mismatches")
sys.exit(1 if bad else 0)
