#!/usr/bin/env python3
"""Derive tests/golden/reference_maps.npz from the reference's committed RESULT files (data, not source).

Run in the build container (needs /root/reference); the GPU box only sees the .npz.

What is kept, bin by bin (VERDICT r01 "next" #1: the full 180x90 maps, not only their theta profiles):
  * the seven complete per-position maps of fluxAtObserverOptimize.C::sweepSeries / sweepDetector
    (results_overnight_03_31..., results_overnight_04_1...), 50 000 rays per position, as INTEGER hit
    counts (fraction * rays_per_position is an integer to 1e-9: the CSV prints 6 decimals of k/50000);
  * the rows that exist of the two maps whose run was cut short (same folders, `_3.csv` and `_4.csv`);
  * per folder of trace-once files (trace_once_test_04_2..., portAngleSweep_04_03..., portAngleSweep_04_02...)
    the bin-wise SUM of the integer hit counts of all files of the folder, the number of files, and the sum
    of their `Total rays exiting port` footers (these pin hit_line_mode = 1, the GetPoint(nPoints-2) defect).
Keys: <name>_hits [180,90] int32 (-1 = row missing in a cut-short file), index_json = list of per-map metadata.
"""
import glob
import json
import os
import re
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
FAO = os.path.join(REF, "flux_at_observer")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_maps.npz")


def parse_csv(path):
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
                parts = line.strip().split(",")
                if len(parts) == 3:
                    try:
                        rows.append([float(x) for x in parts])
                    except ValueError:
                        pass  # a row cut in the middle
    return meta, np.array(rows)


def to_counts(frac, n):
    k = np.rint(frac * n)
    assert np.abs(k - frac * n).max() < 1e-2 * max(1, n / 50000), np.abs(k - frac * n).max()
    return k.astype(np.int32)


out = {}
index = []

for folder in ("results_overnight_03_31-60_0_-75_5", "results_overnight_04_1-60_0_-75_5"):
    for p in sorted(glob.glob(os.path.join(FAO, folder, "*.csv"))):
        meta, rows = parse_csv(p)
        nt, nph = int(meta["Theta bins"]), int(meta["Phi bins"])
        n = int(meta["Number of rays per position"])
        nrows = rows.shape[0]
        full = nrows == nt * nph
        hits = np.full(nt * nph, -1, np.int32)
        hits[:nrows] = to_counts(rows[:, 2], n)
        # the rows are theta-major, phi-minor, at the bin centres
        th = (np.arange(nrows) // nph + 0.5) * 90.0 / nt
        ph = (np.arange(nrows) % nph + 0.5) * 360.0 / nph
        assert np.abs(rows[:, 0] - th).max() < 1e-6 and np.abs(rows[:, 1] - ph).max() < 1e-6
        m = re.match(r"(\d+) out of (\d+)", meta.get("Total ray hits", ""))
        # name: pp_<mm_dd of the folder>_<k>, k = the suffix the reference's getUniqueFilename gave the file (none = 0)
        stem = os.path.basename(p)[:-4]
        k = stem.rsplit("_", 1)[1] if stem.rsplit("_", 1)[1].isdigit() and not stem.endswith("-75") else "0"
        name = "pp_" + re.match(r"results_overnight_(\d+_\d+)", folder).group(1) + "_" + k
        info = {
            "name": name, "kind": "per_position", "file": os.path.relpath(p, REF), "complete": bool(full),
            "rows_present": int(nrows), "port_deg": float(meta["Exit port angle"].split()[0]),
            "rays_per_position": n, "n_theta": nt, "n_phi": nph,
            "source_position": [float(x.replace("cm", "")) for x in meta["Source position (x,y,z)"].split(",")],
            "source_direction": [float(x) for x in meta["Source direction (x,y,z)"].split(",")],
            "reflectance": float(meta["Mirror reflectance"]), "roughness": float(meta["Gaussian roughness"]),
            "r_in": float(meta["Sphere inner radius"].replace("cm", "")),
            "r_out": float(meta["Sphere outer radius"].replace("cm", "")),
            "total_hits_footer": int(m.group(1)) if m else None,
        }
        if full:
            assert info["total_hits_footer"] == int(hits.sum()), (info["total_hits_footer"], hits.sum())
        out[name + "_hits"] = hits.reshape(nt, nph)
        index.append(info)

for folder in ("trace_once_test_04_2-60_0_-75_5", "portAngleSweep_04_03_-60_0_-75_164",
               "portAngleSweep_04_02_-60_0_-75_160"):
    tot = np.zeros(16200, np.int64)
    nf, exited, port, nr = 0, 0, None, None
    per_file_sum = []
    for p in sorted(glob.glob(os.path.join(FAO, folder, "*.csv"))):
        meta, rows = parse_csv(p)
        if rows.shape[0] != 16200:
            continue
        m = re.match(r"(\d+) out of (\d+)", meta.get("Total rays exiting port", ""))
        nr = int(m.group(2))
        k = to_counts(rows[:, 2], nr)
        tot += k
        per_file_sum.append(int(k.sum()))
        exited += int(m.group(1))
        port = float(meta["Exit port angle"].split()[0])
        nf += 1
    name = "to_" + str(int(port))
    out[name + "_hits"] = tot.reshape(180, 90).astype(np.int32)
    index.append({"name": name, "kind": "trace_once_sum", "folder": folder, "n_files": nf, "rays_per_file": nr,
                  "port_deg": port, "exited_sum": exited, "per_file_total_hits": per_file_sum,
                  "source_position": [-60.0, 0.0, -75.0], "source_direction": [5.0, 0.0, 0.0]})

out["index_json"] = np.array(json.dumps(index, indent=1))
np.savez_compressed(OUT, **out)
print("wrote", OUT, os.path.getsize(OUT), "bytes")
for i in index:
    print(i["name"], i["kind"], i.get("port_deg"), i.get("source_direction"), i.get("complete"), i.get("rows_present"),
          i.get("total_hits_footer"), i.get("n_files"), i.get("exited_sum"))
