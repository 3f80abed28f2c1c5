#!/usr/bin/env python3
"""Condense gpurun_out/prof_* (written by tools/profile.sh on the GPU box) into
profiles/<tag>_summary.md, profiles/<tag>_kernel_stats.csv and profiles/pmc_summary.json."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
note = sys.argv[2] if len(sys.argv) > 2 else ""
KERNEL = "isx_trace_bin_kernel"
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)


def one(pattern):
    g = sorted(glob.glob(os.path.join(OUT, pattern)), key=os.path.getmtime)
    return g[-1] if g else None   # newest: gpurun merges successive runs into the same directories


lines = [f"# rocprofv3 summary `{tag}` — {KERNEL}", "", note, ""]
kt = one("prof_kt/*/*_kernel_trace.csv")
big = []
if kt:
    rows = [r for r in csv.DictReader(open(kt)) if KERNEL in r["Kernel_Name"]]
    durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
    mx = max(durs)
    big = [d for d in durs if d > 0.5 * mx]
    r = rows[durs.index(mx)]
    lines += ["## kernel trace (`rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --cpu-rays 0`)", "",
              f"* full-size launches (5e7 rays): {len(big)}, average {sum(big)/len(big):.3f} ms, min {min(big):.3f}, max {max(big):.3f}",
              f"* grid {r['Grid_Size_X']} threads = {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])} workgroups x {r['Workgroup_Size_X']}, "
              f"VGPR_Count {r['VGPR_Count']}, SGPR_Count {r['SGPR_Count']}, scratch {r['Scratch_Size']}, "
              f"LDS_Block_Size {r['LDS_Block_Size']} (dynamic LDS is not shown by the trace)", ""]
    ks = one("prof_kt/*/*_kernel_stats.csv")
    if ks:
        shutil.copy(ks, os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
        lines += ["`--stats` table (all kernels of the process; the two ~3 ms calls are bench.py's 4096-ray reduce-path probe):", "", "```"]
        lines += open(ks).read().splitlines()[:6]
        lines += ["```", ""]

pmc = {}
for d in sorted(glob.glob(os.path.join(OUT, "prof_pmc_*"))):
    if not os.path.isdir(d):
        continue
    f = sorted(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")), key=os.path.getmtime)
    if not f:
        continue
    by = collections.defaultdict(dict)
    for r in csv.DictReader(open(f[-1])):
        if KERNEL in r["Kernel_Name"]:
            by[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    # keep full-size dispatches: those whose first counter is within 2x of the max
    if not by:
        continue
    key = next(iter(next(iter(by.values()))))
    mx = max(v[key] for v in by.values())
    sel = [v for v in by.values() if v[key] > 0.5 * mx]
    for k in sel[0]:
        pmc[k] = sum(v[k] for v in sel) / len(sel)

if pmc:
    lines += ["## PMC counters, per full-size launch (separate `--pmc` passes, averages)", "", "| counter | value |", "|---|---|"]
    for k in sorted(pmc):
        lines.append(f"| {k} | {pmc[k]:.6g} |")
    lines.append("")
    summ = {"tag": tag}
    if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        fetch_b, write_b = pmc["FETCH_SIZE"] * 1024, pmc["WRITE_SIZE"] * 1024
        summ.update(fetch_bytes=fetch_b, write_bytes=write_b, hbm_bytes_per_launch=fetch_b + write_b,
                    note="FETCH_SIZE/WRITE_SIZE are KiB. gfx950 FETCH_SIZE under-counts wide coalesced streams by 2x "
                         "(MI355X_MICROARCH.md §HBM); this kernel's reads are a few KB of scalar/L2 traffic so no "
                         "correction is applied; WRITE_SIZE is exact for the 8-byte atomics of the histogram flush.")
        lines += [f"HBM bytes per launch: fetch {fetch_b/1e6:.3f} MB + write {write_b/1e6:.3f} MB = {(fetch_b+write_b)/1e6:.3f} MB "
                  f"(algorithmic: 0.1296 MB).", ""]
    if big and "GRBM_GUI_ACTIVE" in pmc:
        clk = pmc["GRBM_GUI_ACTIVE"] / 8 / (sum(big) / len(big) * 1e-3) / 1e9
        summ["clock_ghz"] = clk
        lines.append(f"* effective clock = GRBM_GUI_ACTIVE/8/t = {clk:.3f} GHz")
    if "SQ_ACTIVE_INST_VALU" in pmc and "GRBM_GUI_ACTIVE" in pmc:
        simd_cycles = 1024 * pmc["GRBM_GUI_ACTIVE"] / 8
        busy = 4 * pmc["SQ_ACTIVE_INST_VALU"] / simd_cycles
        summ["valu_busy"] = busy
        lines.append(f"* VALU busy = 4*SQ_ACTIVE_INST_VALU / (1024 SIMD x cycles) = {busy:.3f}")
    if "SQ_THREAD_CYCLES_VALU" in pmc and "SQ_ACTIVE_INST_VALU" in pmc:
        util = pmc["SQ_THREAD_CYCLES_VALU"] / (64 * pmc["SQ_ACTIVE_INST_VALU"])
        summ["valu_lane_utilization"] = util
        lines.append(f"* VALU lane utilisation = SQ_THREAD_CYCLES_VALU / (64*SQ_ACTIVE_INST_VALU) = {util:.3f}")
    if "SQ_INSTS_VALU" in pmc:
        summ["valu_wave_insts_per_ray"] = pmc["SQ_INSTS_VALU"] / 5e7
        lines.append(f"* VALU wave-instructions per ray = {pmc['SQ_INSTS_VALU']/5e7:.1f}; SALU {pmc.get('SQ_INSTS_SALU',0)/5e7:.1f}; "
                     f"LDS {pmc.get('SQ_INSTS_LDS',0)/5e7:.2f}; VMEM {pmc.get('SQ_INSTS_VMEM',0)/5e7:.5f}")
    summ["counters"] = pmc
    json.dump(summ, open(os.path.join(ROOT, "profiles", "pmc_summary.json"), "w"), indent=1)
    json.dump(summ, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.json"), "w"), indent=1)

open(os.path.join(ROOT, "profiles", f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
