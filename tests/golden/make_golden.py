#!/usr/bin/env python3
"""Derive tests/golden/reference_fixtures.json from the reference's committed RESULT files.

Run in the build container (needs /root/reference); the GPU box only sees the JSON.
Only numbers are taken (data, not source): exit counts from CSV footers, theta profiles
(phi-means) and totals of flux maps, the 100-bin exit-dz histogram, and the physical-disc
sweep profile.  The reference has no seeded tests, so these pin DISTRIBUTIONS.
"""
import glob
import json
import os
import re
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
FAO = os.path.join(REF, "flux_at_observer")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_fixtures.json")


def parse_csv(path):
    """Same contract as flux_analysis.py:11-57: '#'-lines are 'key: value' metadata, one header, rows."""
    meta, rows = {}, []
    with open(path) as f:
        for line in f:
            if line.startswith("#"):
                if ":" in line:
                    k, v = line[1:].strip().split(":", 1)
                    meta[k.strip()] = v.strip()
            elif line.startswith("theta"):
                continue
            elif line.strip():
                rows.append([float(x) for x in line.strip().split(",")])
    return meta, np.array(rows)


def exit_counts(folder):
    out = []
    for p in sorted(glob.glob(os.path.join(FAO, folder, "*.csv"))):
        meta, _ = parse_csv(p)
        m = re.match(r"(\d+) out of (\d+)", meta.get("Total rays exiting port", ""))
        if m:
            out.append({"file": os.path.relpath(p, REF), "exited": int(m.group(1)), "n": int(m.group(2)),
                        "port_deg": float(meta["Exit port angle"].split()[0]),
                        "trace_s": float(meta["Ray tracing time"].split()[0]),
                        "sweep_s": float(meta["Detector sweep time"].split()[0])})
    return out


def per_position_map(path):
    meta, rows = parse_csv(path)
    nt, nph = int(meta["Theta bins"]), int(meta["Phi bins"])
    if rows.shape[0] != nt * nph:
        return None
    frac = rows[:, 2].reshape(nt, nph)
    m = re.match(r"(\d+) out of (\d+)", meta.get("Total ray hits", ""))
    return {
        "file": os.path.relpath(path, REF),
        "port_deg": float(meta["Exit port angle"].split()[0]),
        "rays_per_position": int(meta["Number of rays per position"]),
        "source_direction": [float(x) for x in meta["Source direction (x,y,z)"].split(",")],
        "n_theta": nt, "n_phi": nph,
        "sum_fraction": float(frac.sum()),
        "theta_profile": [float(x) for x in frac.mean(axis=1)],
        "total_hits": int(m.group(1)) if m else None,
        "total_rays": int(m.group(2)) if m else None,
        "wall_s": float(meta["Total execution time"].split()[0]) if "Total execution time" in meta else None,
    }


fx = {"_generated_by": "tests/golden/make_golden.py", "_source": "bdagnillo/altair-raytracing @ 2025-05-09 result files"}

fx["exit_counts"] = (exit_counts("trace_once_test_04_2-60_0_-75_5") + exit_counts("portAngleSweep_04_03_-60_0_-75_164")
                     + exit_counts("portAngleSweep_04_02_-60_0_-75_160"))

# trace-once maps (affected by the reference's GetPoint(nPoints-2) defect): kept to pin the
# optional hit_line_mode=1 compatibility switch
to = []
for folder in ("trace_once_test_04_2-60_0_-75_5", "portAngleSweep_04_03_-60_0_-75_164"):
    profs, sums, port = [], [], None
    for p in sorted(glob.glob(os.path.join(FAO, folder, "*.csv"))):
        meta, rows = parse_csv(p)
        if rows.shape[0] != 16200:
            continue
        fr = rows[:, 2].reshape(180, 90)
        profs.append(fr.mean(axis=1)); sums.append(float(fr.sum())); port = float(meta["Exit port angle"].split()[0])
    to.append({"folder": folder, "port_deg": port, "n_files": len(profs), "rays": 100000,
               "sum_fraction_mean": float(np.mean(sums)), "sum_fraction_std": float(np.std(sums, ddof=1)),
               "theta_profile_mean": [float(x) for x in np.mean(profs, axis=0)]})
fx["traceonce_maps"] = to

maps = []
for folder in ("results_overnight_03_31-60_0_-75_5", "results_overnight_04_1-60_0_-75_5"):
    for p in sorted(glob.glob(os.path.join(FAO, folder, "*.csv"))):
        m = per_position_map(p)
        if m:
            maps.append(m)
fx["per_position_maps"] = maps

# exit-direction dz histogram (angular_dist.txt: 'bin_center content')
ad = np.loadtxt(os.path.join(REF, "angular_dist.txt"), comments="#")
fx["angular_dist"] = {"bin_centers": [float(x) for x in ad[:, 0]], "content": [int(x) for x in ad[:, 1]]}

# nonLambertianFlux.C map (45x20, 10 cm detector, 1e5 rays/position)
_, rows = parse_csv(os.path.join(FAO, "fluxmap_data.csv"))
fr = rows[:, 2].reshape(45, 20)
fx["nonlambertian_map"] = {"n_theta": 45, "n_phi": 20, "sum_fraction": float(fr.sum()),
                           "theta_profile": [float(x) for x in fr.mean(axis=1)]}

# physical disc sweep detector_sweep.txt (theta x 360 phi, 1000 rays)
ds = np.loadtxt(os.path.join(REF, "detector_sweep.txt"), skiprows=1)
thetas = sorted(set(ds[:, 0]))
fx["disc_sweep"] = {"theta_deg": [float(t) for t in thetas],
                    "phi_mean_fraction": [float(ds[ds[:, 0] == t, 2].mean()) for t in thetas],
                    "n_phi": int((ds[:, 0] == thetas[0]).sum())}

# first line of one golden CSV header + a few rows: the file-format contract for the writers
with open(os.path.join(FAO, "results_overnight_03_31-60_0_-75_5", "fluxmap_50000rays_180x90_src-60_0_-75.csv")) as f:
    lines = f.read().splitlines()
fx["csv_format_sample"] = {"header": lines[:15], "first_rows": lines[15:18], "footer": lines[-3:]}
with open(os.path.join(FAO, "trace_once_test_04_2-60_0_-75_5",
                       "fluxmap_traceonce_100000rays_180x90_src-60_0_-75.csv")) as f:
    lines = f.read().splitlines()
fx["csv_format_sample_traceonce"] = {"header": lines[:16], "first_rows": lines[16:19], "footer": lines[-5:]}

# ---------------------------------------------------------------------------------------------- round 5: the files not used so far
RES = os.path.join(FAO, "results")

# (a) the one complete trace-once map at the OLDER source position (src z = -80): header = rays + detector size only.  Assumed, and
# stated here: direction (5,2,0) as in the sibling twofold / per-position files of the same evening (results/fluxmap_twofold_50000rays
# _180x90_src-60_0_-80.csv header), port 170, the production surface (rho .99, sigma .01, Lambertian).  Pins hit_line_mode = 1.
meta, rows = parse_csv(os.path.join(RES, "fluxmap_traceonce_50000rays_180x90_src-60_0_-80.csv"))
fr = rows[:, 2].reshape(180, 90)
fx["traceonce_src_m80"] = {
    "file": "flux_at_observer/results/fluxmap_traceonce_50000rays_180x90_src-60_0_-80.csv", "rays": int(meta["Number of rays"]),
    "detector_cm": 40.0, "assumed": {"source_position": [-60.0, 0.0, -80.0], "source_direction": [5.0, 2.0, 0.0], "port_deg": 170.0},
    "sum_fraction": float(fr.sum()), "theta_profile": [float(x) for x in fr.mean(axis=1)],
    "row0": [float(x) for x in fr[0]]}

# (b) "Detector Data" files of an older revision of fluxAtObserver.C (2025-03-23/25): sigma 0.75, "20cm x 20cm", rho .99, Lambertian;
# neither the source nor the detector model of that revision is recorded (tests/golden/README.md: what was tried)
meta, rows = parse_csv(os.path.join(RES, "detector_data_50000rays.csv"))
fr = rows[:, 2].reshape(180, 90)
dd = {"file": "flux_at_observer/results/detector_data_50000rays.csv", "rays_per_position": 50000, "roughness": float(meta["Gaussian roughness"]),
      "detector_dimensions": meta["Detector dimensions"], "sum_fraction": float(fr.sum()), "theta_profile": [float(x) for x in fr.mean(axis=1)],
      "coarse": []}
for p in sorted(glob.glob(os.path.join(RES, "detector_data_50000rays_4050points*.csv"))):
    m2, r2 = parse_csv(p)
    f2 = r2[:, 2].reshape(45, 90)
    dd["coarse"].append({"file": os.path.relpath(p, REF), "generated": m2.get("Detector Data - Generated"), "sum_fraction": float(f2.sum()),
                         "theta_profile": [float(x) for x in f2.mean(axis=1)]})
fx["detector_data_sigma075"] = dd

# (c) 3dRayLog.txt: 100 000 un-binned exit directions of distributionSphereDetectorSweep.C (rho 1, no roughness, src z -80, ONE
# thread) as histograms -- dz x azimuth 10 x 12 for a 2-d chi2, dz in 200 and azimuth in 180 bins for the 1-d distributions
d3 = np.loadtxt(os.path.join(REF, "3dRayLog.txt"), comments="#")
az = np.arctan2(d3[:, 1], d3[:, 0])
H2, _, _ = np.histogram2d(d3[:, 2], az, bins=[np.linspace(-1.0, 0.0, 11), np.linspace(-np.pi, np.pi, 13)])
fx["ray_log_3d"] = {"file": "3dRayLog.txt", "n": int(len(d3)), "max_norm_error": float(np.abs((d3 ** 2).sum(1) - 1).max()),
                    "dz_edges": [-1.0, 0.0, 200], "az_edges_pi": [-1.0, 1.0, 180],
                    "dz_hist": [int(x) for x in np.histogram(d3[:, 2], bins=np.linspace(-1.0, 0.0, 201))[0]],
                    "az_hist": [int(x) for x in np.histogram(az, bins=np.linspace(-np.pi, np.pi, 181))[0]],
                    "dz_az_10x12": [[int(x) for x in row] for row in H2], "upward": int((d3[:, 2] >= 0).sum())}

# (d) detector_sweep2.txt: the physical-disc sweep at 1 deg x 1 deg, 1000 rays per position, cut short at theta = +1 deg (the last
# line is half written): the 46 complete theta rows, phi-mean
rows2 = []
with open(os.path.join(REF, "detector_sweep2.txt")) as f:
    for ln in f.read().splitlines()[1:]:
        parts = ln.split()
        if len(parts) == 3:
            rows2.append([float(x) for x in parts])
ds2 = np.array(rows2)
full = [t for t in sorted(set(ds2[:, 0])) if (ds2[:, 0] == t).sum() == 360]
fx["disc_sweep2"] = {"file": "detector_sweep2.txt", "theta_deg": [float(t) for t in full], "n_phi": 360, "rays_per_position": 1000,
                     "phi_mean_fraction": [float(ds2[ds2[:, 0] == t, 2].mean()) for t in full],
                     "rows_total": int(len(ds2)), "rows_dropped": int(len(ds2) - 360 * len(full)),
                     # every position: hits of its 1000 rays, [theta][phi = 0..359 deg] (the phi structure pins addDetectorDisk's rotation quirk)
                     "hits": [[int(round(x * 1000)) for x in ds2[ds2[:, 0] == t][np.argsort(ds2[ds2[:, 0] == t, 1]), 2]] for t in full]}
fx["disc_sweep"]["hits"] = [[int(round(x * 1000)) for x in ds[ds[:, 0] == t][np.argsort(ds[ds[:, 0] == t, 1]), 2]] for t in thetas]

# (e) small and cut-short files whose revision CAN be identified from the file name / header: rows present as integer hit counts
small = []


def small_file(rel, rays, cfg, note):
    meta, rows = parse_csv(os.path.join(FAO, rel))
    k = np.rint(rows[:, 2] * rays)
    assert np.abs(k - rows[:, 2] * rays).max() < 1e-6
    small.append({"file": "flux_at_observer/" + rel, "rays_per_position": rays, "config": cfg, "note": note,
                  "theta": [float(x) for x in rows[:, 0]], "phi": [float(x) for x in rows[:, 1]], "hits": [int(x) for x in k]})


FAO_CFG = {"macro": "fluxAtObserver.C", "src": [-60, 0, -80], "dir": [5, 2, 0], "reflectance": 1.0, "roughness": 0.5, "max_points": 10000, "box_half": 200.0}
OPT_CFG = {"macro": "fluxAtObserverOptimize.C / fluxAtObserverFast.C", "src": [-60, 0, -80], "dir": [5, 2, 0], "reflectance": 0.99, "roughness": 0.01,
           "max_points": 50000, "box_half": 300.0}
small_file("results/fluxmap_data_1000rays_300points.csv", 1000, dict(FAO_CFG, n_theta=15, n_phi=20, det_diameter=10.0), "complete 15x20 map")
small_file("results/fluxmap_data_10000rays_300points.csv", 10000, dict(FAO_CFG, n_theta=15, n_phi=20, det_diameter=10.0), "cut short")
small_file("results/fluxmap_data_50000rays_16200points.csv", 50000, dict(FAO_CFG, n_theta=180, n_phi=90, det_diameter=10.0), "cut short; header '# y direction: 2'")
small_file("results/fluxmap_50000rays_180x90_src-60_0_-80.csv", 50000, dict(OPT_CFG, n_theta=180, n_phi=90, det_diameter=40.0), "cut short")
small_file("results/fluxmap_twofold_50000rays_180x90_src-60_0_-80.csv", 50000, dict(OPT_CFG, n_theta=180, n_phi=90, det_diameter=40.0, fold=2),
           "cut short; twofold: rows alternate (i,j), (i,j+45) and share their rays")
small_file("results/fluxmap_twofold_50000rays_180x90_src-60_0_-80_1.csv", 50000, dict(OPT_CFG, n_theta=180, n_phi=90, det_diameter=40.0, fold=2),
           "cut short; twofold")
fx["small_files"] = small

with open(OUT, "w") as f:
    json.dump(fx, f, indent=1)
print("wrote", OUT, os.path.getsize(OUT), "bytes")
