#!/usr/bin/env python3
"""Kernel times of the configurations that are NOT the lean Lambertian border (GPU box): the cos^2-lobe surface of
"nonLambertianFlux copy.C":31-70,188-221 (the de-facto CustomMirror), ROBAST's rough-specular border (EnableLambertian(false) +
SetGaussianRoughness), the origin-compat hit line of fluxAtObserverFast.C:1181-1201 -- and the cost of SMALL calls (5e4 rays:
the reference's own call size, fluxAtObserverOptimize.C:568).

    python tools/bench_surfaces.py [--rays N] [--only lobe|rough|compat|small] [--pipeline 0|1]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import altair_raytracing_amd as isx  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rays", type=int, default=50_000_000)
ap.add_argument("--only", default="all")
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--surface-pipeline", type=int, default=-1,
                help="isx_set_option('surface_pipeline'): 0 = round 1's fused kernel for these surfaces, 1 = the assist-wave pipeline")
A = ap.parse_args()
isx.load(); isx.init(0)
if A.surface_pipeline >= 0:
    isx.set_option("surface_pipeline", A.surface_pipeline)
out = {"device": isx.device_info()[0], "rays": A.rays}


def want(name):
    return A.only in ("all", name)


def timed(cfg, n):
    best, st, kinds = None, None, None
    for _ in range(A.reps):
        t0 = time.perf_counter()
        _, s = isx.fluxmap(cfg, n, 0x5EED0001)
        wall = (time.perf_counter() - t0) * 1e3
        if best is None or s.t_kernel_ms < best:
            best, st, kinds, w = s.t_kernel_ms, s, isx.last_kernel_ms(), wall
    return {"kernel_ms": best, "wall_ms": w, "Mrays_s": n / best / 1e3, "single_trace_bin_ms": kinds,
            "wall_hits_per_ray": st.wall_hits / n, "counted_per_ray": st.counted_below_z / n,
            "increments_per_ray": st.bin_increments / n, "suspended": st.suspended}


n = A.rays
if want("lean"):
    out["headline (Lambertian border)"] = timed(isx.default_config(), n)
if want("lobe"):
    c = isx.default_config(); c.surface_model = 1
    out["lobe surface, rho 0.99"] = timed(c, n)
if want("rough"):
    c = isx.default_config(); c.lambertian = 0; c.roughness_rad = 0.5
    out["rough specular, sigma 0.5, rho 0.99"] = timed(c, n)
    c = isx.default_config(); c.lambertian = 0; c.roughness_rad = 0.1
    out["rough specular, sigma 0.1, rho 0.99"] = timed(c, n)
if want("compat"):
    c = isx.default_config(); c.hit_line_mode = 1
    out["origin-compat hit line (Lambertian border)"] = timed(c, n)
if want("small"):
    c = isx.default_config()
    det = isx.detector_table(c)[90 * 90 + 45]
    small = {}
    for n_small in (50_000, 500_000):
        isx.fluxmap(c, n_small, 1)
        t0 = time.perf_counter()
        ks = []
        for k in range(50):
            _, s = isx.fluxmap(c, n_small, 1, k * n_small)
            ks.append(s.t_kernel_ms)
        small[f"isx_fluxmap {n_small}"] = {"wall_ms_per_call": (time.perf_counter() - t0) * 1e3 / 50, "kernel_ms": float(np.mean(ks))}
        isx.trace_rays_detector(c, det, c.det_diameter, n_small, 1)
        t0 = time.perf_counter()
        ks = []
        for k in range(50):
            _, s = isx.trace_rays_detector(c, det, c.det_diameter, n_small, 1, k * n_small)
            ks.append(s.t_kernel_ms)
        small[f"isx_trace_rays_detector {n_small}"] = {"wall_ms_per_call": (time.perf_counter() - t0) * 1e3 / 50, "kernel_ms": float(np.mean(ks))}
    out["small calls"] = small
print(json.dumps(out, indent=1))
