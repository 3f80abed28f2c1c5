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

with open(OUT, "w") as f:
    json.dump(fx, f, indent=1)
print("wrote", OUT, os.path.getsize(OUT), "bytes")
