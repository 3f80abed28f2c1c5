#!/usr/bin/env python3
"""Kernel times of the BASELINE.json configurations that bench.py does not report (GPU box):
configs[2] nonLambertianFlux source model, 5e7 rays, 180x90 map; configs[3] physical-disc sweep (181x2 discs,
1e7 rays, shell 100.1-105, rho=1); the per-position map of the reference's 12 524 s run (5e4 rays x 16 200)."""
import json
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import altair_raytracing_amd as isx  # noqa: E402

import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--only", choices=["all", "flux", "chord", "brdf", "discs", "perpos", "lobe", "rough", "compat"], default="all",
                help="run one configuration only (what tools/profile.sh profiles)")
ap.add_argument("--reps", type=int, default=0, help="launches per configuration (0: the defaults below)")
ARGS = ap.parse_args()
isx.load(); isx.init(0)
out = {}


def want(name):
    return ARGS.only in ("all", name)


def best(fn, reps=3):
    ts = []
    for _ in range(ARGS.reps or reps):
        st = fn()
        ts.append(st.t_kernel_ms)
    return min(ts), st


n = 50_000_000
if want("flux"):
    c = isx.default_config()
    ms, st = best(lambda: isx.fluxmap(c, n, 0x5EED0001)[1])
    out["configs[1] lambertian 5e7"] = {"ms": ms, "Mrays_s": n / ms / 1e3}
if want("chord"):
    c = isx.default_config()
    c.trace_mode = 1
    ms, st = best(lambda: isx.fluxmap(c, n, 0x5EED0001)[1])
    out["configs[1] chord mode"] = {"ms": ms, "Mrays_s": n / ms / 1e3}

# the other border models and the origin-compat hit line (round 5: on the assist-wave pipeline; --surface-pipeline 0 = before)
for name, setup in (("lobe", lambda q: setattr(q, "surface_model", 1)),
                    ("rough", lambda q: (setattr(q, "lambertian", 0), setattr(q, "roughness_rad", 0.5))),
                    ("compat", lambda q: setattr(q, "hit_line_mode", 1))):
    if want(name):
        c = isx.default_config()
        setup(c)
        row = {}
        for sp in ((1, 0) if ARGS.only == "all" else (1,)):
            isx.set_option("surface_pipeline", sp)
            ms, st = best(lambda: isx.fluxmap(c, n, 0x5EED0001)[1], reps=2)
            row["pipeline" if sp else "fused kernel of round 1 (before)"] = {"ms": ms, "Mrays_s": n / ms / 1e3, "kinds_ms": isx.last_kernel_ms()}
        isx.set_option("surface_pipeline", 1)
        row["wall_hits_per_ray"] = st.wall_hits / n
        out[{"lobe": "cos^2-lobe border (nonLambertianFlux copy.C:31-70), rho 0.99, 5e7", "rough": "rough-specular border, sigma 0.5, rho 0.99, 5e7",
             "compat": "origin-compat hit line (fluxAtObserverFast.C:1181-1201), 5e7"}[name]] = row

if want("brdf"):
    c = isx.default_config()
    c.source_model = isx.SOURCE_BRDF; c.brdf[0], c.brdf[1], c.brdf[2] = 0.3, 0.4, 0.6
    c.roughness_rad = 0.5; c.reflectance = 1.0; c.max_points = 10000; c.box_half = 200.0
    ms, st = best(lambda: isx.fluxmap(c, n, 0x5EED0001)[1], reps=2)
    out["configs[2] nonLambertianFlux source 5e7"] = {"ms": ms, "Mrays_s": n / ms / 1e3, "wall_hits_per_ray": st.wall_hits / n}


def disc_positions():
    """rootMacros::detectorDiskPlacement (integratingSphereDetectorSweep.C:145-172): centre at 200 cm from the origin, tube
    axis (sin rotTheta, 0, cos rotTheta) for every phi (TGeoRotation::RotateY left-multiplies)."""
    discs = []
    for th in np.arange(-45.0, 45.0 + 1e-9, 0.5):
        for ph in (0.0, 180.0):
            t, p = math.radians(th), math.radians(ph)
            x, y, z = 200 * math.sin(t) * math.cos(p), 200 * math.sin(t) * math.sin(p), -200 * math.cos(t)
            dx, dy, dz = 0 - x, 0 - y, -100 - z
            rot = -math.atan2(math.sqrt(dx * dx + dy * dy), dz)
            discs.append([x, y, z, math.sin(rot), 0.0, math.cos(rot)])
    return np.array(discs)


if want("discs"):
    c = isx.default_config()
    c.r_out = 105.0; c.reflectance = 1.0; c.roughness_rad = 0.0; c.max_points = 10000; c.box_half = 200.0
    c.src[2] = -80.0
    discs = disc_positions()
    n4 = 10_000_000
    ms, st = best(lambda: isx.disc_sweep(c, discs, 5.0, 0.1, n4, 7)[1], reps=2)
    out["configs[3] disc sweep, 362 discs share 1e7 rays"] = {"ms": ms, "Mrays_s": n4 / ms / 1e3}
    rpp = 1_000_000
    ms, st = best(lambda: isx.disc_sweep_per_position(c, discs, 5.0, 0.1, rpp, 7)[1], reps=2)
    out["configs[3] per-position disc sweep, 362 x 1e6 rays, one launch"] = {"ms": ms, "Mrays_s": rpp * len(discs) / ms / 1e3}

if want("perpos"):
    c = isx.default_config()
    ms, st = best(lambda: isx.fluxmap_per_position(c, 50_000, 0x5EED0001)[1], reps=2)
    out["per-position map 5e4 x 16200 (reference: 12 524 s)"] = {"ms": ms, "Mrays_s": 50_000 * 16200 / ms / 1e3}
print(json.dumps(out, indent=1))
