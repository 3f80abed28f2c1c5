"""Soak test of the physical-disc sinks against the oracle: random disc sets (centres anywhere below / beside / inside the port
region, random unit axes, radii from 0.5 to 150 cm, half thickness 0.01 to 20 cm), shared-ray sweep and per-position sweep, pencil
source with random port angles.  GPU box (the oracle is the checker):  python tools/soak_discs.py [n_sets] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import altair_raytracing_amd as isx
import oracle as orc
isx.load(); isx.init(0)
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 100
if os.environ.get("ISX_DISC_PIPELINE"): isx.set_option("disc_pipeline", int(os.environ["ISX_DISC_PIPELINE"]))   # the pipeline form of the shared-ray sweep
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
bad = 0
for k in range(NS):
    ci, co = isx.default_config(), orc.default_config()
    tm = float(rng.uniform(150, 178)); refl = float(rng.choice([0.9, 0.99, 1.0]))
    src = [float(rng.uniform(-70, 70)), float(rng.uniform(-30, 30)), float(rng.uniform(-85, 40))]
    dr = [float(rng.uniform(1, 6)), float(rng.uniform(-3, 3)), float(rng.uniform(-2, 2))]
    for c in (ci, co):
        c.theta_max_deg = tm; c.reflectance = refl; c.max_points = 2000
        for a in range(3):
            c.src[a] = src[a]; c.dir[a] = dr[a]
        if k % 4 == 3:
            c.trace_mode = 1
    nd = int(rng.integers(1, 80))
    cen = np.stack([rng.uniform(-250, 250, nd), rng.uniform(-250, 250, nd), rng.uniform(-290, -60, nd)], 1)
    if k % 2:   # aim half of the sets at the cone below the port
        th = rng.uniform(0, 1.2, nd); ph = rng.uniform(0, 2 * np.pi, nd); rr = rng.uniform(5, 190, nd)
        cen = np.stack([rr * np.sin(th) * np.cos(ph), rr * np.sin(th) * np.sin(ph), -100 - rr * np.cos(th)], 1)
    ax = rng.standard_normal((nd, 3)); ax /= np.linalg.norm(ax, axis=1)[:, None]
    ca = np.concatenate([cen, ax], 1)
    radius = float(rng.choice([0.5, 3.0, 15.0, 40.0, 150.0])); half = float(rng.choice([0.01, 0.1, 2.0, 20.0]))
    n = 20000
    gh, gst = isx.disc_sweep(ci, ca, radius, half, n, 700 + k)
    oh, ost = orc.disc_sweep(co, ca, radius, half, n, 700 + k)
    ok = np.array_equal(gh, oh) and gst.exited == ost.exited
    rpp = 2000
    gp, _ = isx.disc_sweep_per_position(ci, ca, radius, half, rpp, 900 + k)
    op, _ = orc.disc_sweep_per_position(co, ca, radius, half, rpp, 900 + k)
    ok = ok and np.array_equal(gp, op)
    bad += (not ok)
    if not ok:
        print("MISMATCH", k, tm, nd, radius, half, int(gh.sum()), int(oh.sum()), int(gp.sum()), int(op.sum()), flush=True)
    if k % 20 == 0:
        print(k, "sets,", bad, "mismatches; hits", int(gh.sum()), int(gp.sum()), flush=True)
print("done:", NS, "disc sets,", bad, "mismatches")
sys.exit(1 if bad else 0)
