"""Soak tests of the remaining entry points and of the scheduling knobs (GPU box; the oracle is the checker):
  * launch-shape / scheduling invariance: random pipeline on/off, pipeline_chunk, trace_block, trace_blocks_per_cu, blocks_per_cu,
    grid_blocks, sched_mask, sched_min, first ray -- the map of a configuration never changes (three configurations: headline,
    chord, BRDF source);
  * isx_exit_directions, isx_trace_rays_detector, isx_fluxmap_series on random configurations == oracle.
python tools/soak_misc.py [n] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import altair_raytracing_amd as isx
import oracle as orc
isx.load(); isx.init(0)
NG = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
CENSUS = ("launched", "exited", "counted_below_z", "absorbed", "suspended", "bin_increments", "wall_hits")
bad = 0
def cfgs(mod):
    a = mod.default_config()
    b = mod.default_config(); b.trace_mode = 1
    c = mod.default_config(); c.source_model = 1; c.brdf[0], c.brdf[1], c.brdf[2] = 0.3, 0.4, 0.6
    c.roughness_rad = 0.5; c.reflectance = 1.0; c.max_points = 10000; c.box_half = 200.0
    return [a, b, c]
# ---- scheduling invariance
n = 300_000
refs = []
for c in cfgs(isx):
    first = int(rng.integers(0, 1 << 40))
    h, st = isx.fluxmap(c, n, 77, first)
    refs.append((c, first, h.copy(), st))
for k in range(NG):
    opts = dict(pipeline=int(rng.integers(0, 2)), pipeline_chunk=int(rng.choice([4096, 5000, 65537, 100_000, 1 << 20, 1 << 26])),
                trace_block=int(rng.choice([64, 128, 256, 512, 1024])), trace_blocks_per_cu=int(rng.integers(0, 33)),
                blocks_per_cu=int(rng.integers(1, 9)), grid_blocks=int(rng.choice([0, 0, 0, 1, 3, 17, 300])),
                sched_mask=int(rng.choice([0, 1, 3, 7, 15, 255])), sched_min=int(rng.integers(1, 66)),
                # round 3: ray queue, assist-wave trace kernels, slot-queue binning kernel, overlapped chunks
                ray_sub=int(rng.choice([0, 0, 64, 100, 777, 4096, 1 << 20])), assist=int(rng.integers(0, 2)),
                assist_block=int(rng.choice([128, 192, 256, 384, 512, 640, 768])), bin_slots=int(rng.integers(0, 2)), bin_cols=int(rng.integers(0, 3)),
                bin_block=int(rng.choice([256, 512, 1024])), bin_blocks_per_cu=int(rng.integers(0, 4)),
                overlap=int(rng.choice([0, 0, 0, 2, 3, 5])), overlap_trace_streams=int(rng.integers(1, 3)))
    for key, val in opts.items():
        isx.set_option(key, val)
    for (c, first, h0, st0) in refs:
        h, st = isx.fluxmap(c, n, 77, first)
        ok = np.array_equal(h, h0) and all(getattr(st, f) == getattr(st0, f) for f in CENSUS)
        bad += (not ok)
        if not ok:
            print("MISMATCH (schedule)", opts, c.trace_mode, c.source_model, flush=True)
for key, val in dict(pipeline=1, pipeline_chunk=1 << 26, trace_block=512, trace_blocks_per_cu=0, blocks_per_cu=1, grid_blocks=0,
                     sched_mask=3, sched_min=12, ray_sub=0, assist=1, assist_block=0, bin_slots=1, bin_cols=1, bin_block=512, bin_blocks_per_cu=0,
                     overlap=0, overlap_trace_streams=1).items():
    isx.set_option(key, val)
print("schedule soak:", NG, "settings x 3 configurations,", bad, "mismatches", flush=True)
# ---- other entry points vs oracle
for k in range(NG):
    v = dict(theta_max_deg=float(rng.uniform(150, 178)), reflectance=float(rng.choice([0.9, 0.99, 1.0])), max_points=int(rng.choice([200, 3000])))
    if k % 3 == 1: v["trace_mode"] = 1
    if k % 3 == 2: v.update(lambertian=0, roughness_rad=0.2, reflectance=0.9)
    ci, co = isx.default_config(), orc.default_config()
    for c in (ci, co):
        for f, x in v.items():
            setattr(c, f, x)
    m = 20000
    first = int(rng.integers(0, 1 << 50))
    gi, gd, gc, _ = isx.exit_directions(ci, m, 11 + k, first)
    oi, od, oc = orc.exit_directions(co, m, 11 + k, first)
    ok = gc == oc and np.array_equal(gi[:gc], oi[:oc]) and np.array_equal(gd[:gc].view(np.uint64), od[:oc].view(np.uint64))
    th, ph = float(rng.uniform(0, 1.5)), float(rng.uniform(0, 6.28))
    R = float(rng.choice([50.0, 100.0]))
    det = np.array([R * np.sin(th) * np.cos(ph), R * np.sin(th) * np.sin(ph), -100 - R * np.cos(th), *(lambda a: a / np.linalg.norm(a))(rng.standard_normal(3))])
    width = float(rng.choice([2.0, 40.0, 150.0]))
    gh, _ = isx.trace_rays_detector(ci, det, width, m, 21 + k, first)
    oh, _ = orc.trace_rays_detector(co, det, width, m, 21 + k, first)
    ok = ok and int(gh) == int(oh)
    bad += (not ok)
    if not ok:
        print("MISMATCH (entry points)", k, v, gc, oc, int(gh), int(oh), flush=True)
print("done:", bad, "mismatches")
sys.exit(1 if bad else 0)
