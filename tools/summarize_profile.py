#!/usr/bin/env python3
"""Condense gpurun_out/prof_<NAME>_* (written by tools/profile.sh on the GPU box) into profiles/<tag>_summary.md,
profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc_summary.json -- and, for the headline kernel, profiles/pmc_summary.json
(what bench.py reads for its executed-instruction rooflines, with the fingerprint of the kernel sources it was measured on).

usage: summarize_profile.py <tag> [note] [--name flux] [--kernel isx_trace_bin_kernel] [--rays 5e7] [--headline]
"""
import argparse
import collections
import csv
import glob
import hashlib
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
ap = argparse.ArgumentParser()
ap.add_argument("tag")
ap.add_argument("note", nargs="?", default="")
ap.add_argument("--name", default="flux")
ap.add_argument("--kernel", default="isx_trace_bin_kernel",
                help="kernel name; a comma-separated list (the two kernels of the flux-map pipeline) writes one summary per "
                     "kernel (<tag>_<kernel>) and, with --headline, a combined profiles/pmc_summary.json")
ap.add_argument("--rays", type=float, default=5e7, help="rays per full-size launch (per-ray figures)")
ap.add_argument("--headline", action="store_true", help="also write profiles/pmc_summary.json (bench.py's source)")
ap.add_argument("--section", default="", help="with a kernel list: merge the kernels into profiles/pmc_summary.json under sections[NAME] (what bench.py's "
                                              "extra legs -- configs2 / configs3 / surfaces -- price their kernels with); same source fingerprint required")
ap.add_argument("--command", default="python3 bench.py --steps 3 --warmup 1 --cpu-rays 0")
a = ap.parse_args()
if "," in a.kernel or a.section:   # several kernels of one profiled command: one pass per kernel, then the combined headline file
    import subprocess, sys
    parts = {}
    for kname in a.kernel.split(","):
        short = kname.replace("isx_", "").replace("_kernel", "")
        subprocess.check_call([sys.executable, os.path.abspath(__file__), f"{a.tag}_{short}", a.note, "--name", a.name, "--kernel", kname,
                               "--rays", str(a.rays), "--command", a.command])   # (a single kernel with --section works the same way)
        parts[kname] = json.load(open(os.path.join(ROOT, "profiles", f"{a.tag}_{short}_pmc_summary.json")))
    if a.headline:
        comb = {"tag": a.tag, "kernel": a.kernel, "rays_per_launch": a.rays,
                "kernel_source_sha": next(iter(parts.values()))["kernel_source_sha"], "kernels": parts,
                "kernel_ms": sum(p.get("kernel_ms", 0.0) for p in parts.values()),
                "valu_wave_insts_per_ray": sum(p.get("valu_wave_insts_per_ray", 0.0) for p in parts.values()),
                "hbm_bytes_per_launch": sum(p.get("hbm_bytes_per_launch", 0.0) for p in parts.values())}
        f64 = [p.get("fp64_executed") for p in parts.values()]
        if all(f64):
            comb["fp64_executed"] = {"wave_insts_per_ray": sum(x["wave_insts_per_ray"] for x in f64),
                                     "lane_flop_per_ray": sum(x["lane_flop_per_ray"] for x in f64),
                                     "share_of_valu": sum(x["wave_insts_per_ray"] for x in f64) / comb["valu_wave_insts_per_ray"]}
        keep = {}
        try:   # (sections measured on the same sources survive a new headline pass)
            old = json.load(open(os.path.join(ROOT, "profiles", "pmc_summary.json")))
            if old.get("kernel_source_sha") == comb["kernel_source_sha"]:
                keep = old.get("sections", {})
        except Exception:
            pass
        comb["sections"] = keep
        json.dump(comb, open(os.path.join(ROOT, "profiles", "pmc_summary.json"), "w"), indent=1)
        json.dump(comb, open(os.path.join(ROOT, "profiles", f"{a.tag}_pmc_summary.json"), "w"), indent=1)
    if a.section:
        head = json.load(open(os.path.join(ROOT, "profiles", "pmc_summary.json")))
        sha = next(iter(parts.values()))["kernel_source_sha"]
        if head.get("kernel_source_sha") != sha:
            raise SystemExit(f"profiles/pmc_summary.json is for sources {head.get('kernel_source_sha')}, this pass for {sha}: run the headline pass first")
        head.setdefault("sections", {})[a.section] = {"tag": a.tag, "rays_per_launch": a.rays, "command": a.command, "kernels": parts}
        json.dump(head, open(os.path.join(ROOT, "profiles", "pmc_summary.json"), "w"), indent=1)
    raise SystemExit(0)
tag, note, KERNEL, RAYS = a.tag, a.note, a.kernel, a.rays
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)


def kernel_source_sha():
    h = hashlib.sha256()
    for name in ("isx_device.hpp", "isx_kernels.hpp", "isx_api.hip"):
        with open(os.path.join(ROOT, "altair-raytracing_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def one(pattern):
    g = sorted(glob.glob(os.path.join(OUT, pattern)), key=os.path.getmtime)
    return g[-1] if g else None   # newest: gpurun merges successive runs into the same directories


def is_kernel(name):   # exact kernel, not a longer name that contains it (isx_trace_bin_kernel vs isx_trace_bin_brdf_kernel)
    return name.split("(")[0].strip().strip('"') == KERNEL


P = f"prof_{a.name}"
lines = [f"# rocprofv3 summary `{tag}` — {KERNEL}", "", note, ""]
kt = one(f"{P}_kt/*/*_kernel_trace.csv")
big = []
if kt:
    rows = [r for r in csv.DictReader(open(kt)) if is_kernel(r["Kernel_Name"])]
    durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
    mx = max(durs)
    order = sorted(range(len(rows)), key=lambda i: int(rows[i]["Start_Timestamp"]))
    big_all = [durs[i] for i in order if durs[i] > 0.5 * mx]
    # the first full-size launch of the process is the warm-up step (cold instruction cache, page faults of the workspace:
    # 13.0 against 11.2 ms for the trace kernel): it is listed, not averaged
    big = big_all[1:] if len(big_all) > 1 else big_all
    r = rows[durs.index(mx)]
    lines += [f"## kernel trace (`rocprofv3 --kernel-trace --stats -- {a.command}`)", "",
              f"* full-size launches ({RAYS:.3g} rays): {len(big_all)}, the first (warm-up) {big_all[0]:.3f} ms; the other {len(big)}: average "
              f"{sum(big)/len(big):.3f} ms, min {min(big):.3f}, max {max(big):.3f} -> {RAYS / (sum(big)/len(big)) / 1e3:.1f} Mrays/s",
              f"* grid {r['Grid_Size_X']} threads = {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])} workgroups x {r['Workgroup_Size_X']}, "
              f"VGPR_Count {r['VGPR_Count']}, SGPR_Count {r['SGPR_Count']}, scratch {r['Scratch_Size']}, "
              f"LDS_Block_Size {r['LDS_Block_Size']} (dynamic LDS is not shown by the trace)", ""]
    ks = one(f"{P}_kt/*/*_kernel_stats.csv")
    if ks:
        shutil.copy(ks, os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
        lines += ["`--stats` table (all kernels of the process):", "", "```"]
        lines += [ln[:400] for ln in open(ks).read().splitlines()[:6]]
        lines += ["```", ""]

pmc = {}
for d in sorted(glob.glob(os.path.join(OUT, f"{P}_pmc_*"))):
    if not os.path.isdir(d):
        continue
    f = sorted(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")), key=os.path.getmtime)
    if not f:
        continue
    by = collections.defaultdict(dict)
    for r in csv.DictReader(open(f[-1])):
        if is_kernel(r["Kernel_Name"]):
            by[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    if not by:
        continue
    # keep full-size dispatches: those whose first counter is within 2x of the max
    key = next(iter(next(iter(by.values()))))
    mx = max(v[key] for v in by.values())
    sel = [v for v in by.values() if v[key] > 0.5 * mx] if mx > 0 else list(by.values())
    for k in sel[0]:
        pmc[k] = sum(v[k] for v in sel) / len(sel)

summ = {"tag": tag, "kernel": KERNEL, "rays_per_launch": RAYS, "kernel_source_sha": kernel_source_sha()}
if big:
    summ["kernel_ms"] = sum(big) / len(big)
if pmc:
    lines += ["## PMC counters, per full-size launch (separate `--pmc` passes, averages)", "", "| counter | value |", "|---|---|"]
    for k in sorted(pmc):
        lines.append(f"| {k} | {pmc[k]:.6g} |")
    lines.append("")
    if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        # MI355X_MICROARCH.md, HBM section: on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming
        # read (16 B per lane) -- the binning kernel reads its 48-byte exit lines as three 16-byte loads per lane, so its fetch
        # figure is doubled; the other kernels read a few KB of scalar/L2 traffic (no correction).  WRITE_SIZE is exact for
        # 16-B-per-lane streaming stores (the trace kernel's exit lines) and for the 8-byte atomics of the histogram flush.
        fx = 2.0 if KERNEL in ("isx_bin_lines_kernel", "isx_bin_slots_kernel", "isx_bin_cols_kernel") else 1.0
        fetch_b, write_b = pmc["FETCH_SIZE"] * 1024 * fx, pmc["WRITE_SIZE"] * 1024
        summ.update(fetch_bytes=fetch_b, write_bytes=write_b, hbm_bytes_per_launch=fetch_b + write_b, fetch_correction=fx,
                    note="FETCH_SIZE/WRITE_SIZE are KiB; gfx950 correction of MI355X_MICROARCH.md (HBM section): FETCH_SIZE x2 for "
                         "wide coalesced streaming reads (applied to the binning kernels only), WRITE_SIZE exact.")
        lines += [f"HBM bytes per launch: fetch {fetch_b/1e6:.3f} MB" + (" (FETCH_SIZE x 2, gfx950 streaming-read correction)" if fx != 1.0 else "") +
                  f" + write {write_b/1e6:.3f} MB = {(fetch_b+write_b)/1e6:.3f} MB.", ""]
    if big and "GRBM_GUI_ACTIVE" in pmc:
        clk = pmc["GRBM_GUI_ACTIVE"] / 8 / (sum(big) / len(big) * 1e-3) / 1e9
        summ["clock_ghz"] = clk
        lines.append(f"* effective clock = GRBM_GUI_ACTIVE/8/t = {clk:.3f} GHz")
    if "SQ_ACTIVE_INST_VALU" in pmc and "GRBM_GUI_ACTIVE" in pmc:
        simd_cycles = 1024 * pmc["GRBM_GUI_ACTIVE"] / 8
        busy = 4 * pmc["SQ_ACTIVE_INST_VALU"] / simd_cycles
        # (4 x SQ_ACTIVE_INST_VALU counts every SIMD-cycle in which a VALU instruction of ANY wave is in flight; instructions of
        #  different waves overlap in the pipeline, so the ratio exceeds 1 on a saturated kernel: it says "no idle VALU cycle by this
        #  counter", it is not a utilisation)
        summ["valu_busy_counter_ratio"] = busy
        summ["valu_idle"] = "none by this counter (ratio >= 1)" if busy >= 1.0 else f"{1.0 - busy:.3f} of the SIMD cycles"
        lines.append(f"* 4*SQ_ACTIVE_INST_VALU / (1024 SIMD x cycles) = {busy:.3f}" +
                     (" (>= 1: no idle VALU cycle by this counter; overlapping issue is counted more than once, not a fraction)" if busy >= 1.0 else " = VALU busy"))
    if "SQ_THREAD_CYCLES_VALU" in pmc and "SQ_ACTIVE_INST_VALU" in pmc:
        util = pmc["SQ_THREAD_CYCLES_VALU"] / (64 * pmc["SQ_ACTIVE_INST_VALU"])
        summ["valu_lane_utilization"] = util
        lines.append(f"* VALU lane utilisation = SQ_THREAD_CYCLES_VALU / (64*SQ_ACTIVE_INST_VALU) = {util:.3f}")
    if "SQ_INSTS_VALU" in pmc:
        summ["valu_wave_insts_per_ray"] = pmc["SQ_INSTS_VALU"] / RAYS
        lines.append(f"* VALU wave-instructions per ray = {pmc['SQ_INSTS_VALU']/RAYS:.1f}; SALU {pmc.get('SQ_INSTS_SALU',0)/RAYS:.1f}; "
                     f"LDS {pmc.get('SQ_INSTS_LDS',0)/RAYS:.2f}; VMEM {pmc.get('SQ_INSTS_VMEM',0)/RAYS:.5f}")
        if big:
            rate = pmc["SQ_INSTS_VALU"] / (sum(big) / len(big) * 1e-3) / 1e9
            summ["valu_issue_rate_g"] = rate
            lines.append(f"* VALU issue rate = {rate:.1f} G wave-instr/s of 614.4 (256 CU x 4 SIMD x 2.4 GHz / 4) = {rate/614.4:.3f}")
    f64 = [pmc.get(f"SQ_INSTS_VALU_{k}_F64") for k in ("ADD", "MUL", "FMA", "TRANS")]
    if all(v is not None for v in f64) and "SQ_INSTS_VALU" in pmc:
        add, mul, fma, trans = f64
        wave_insts = add + mul + fma + trans
        util = summ.get("valu_lane_utilization", 1.0)
        lane_flop = (add + mul + 2 * fma + trans) * 64 * util
        summ["fp64_executed"] = {"add": add, "mul": mul, "fma": fma, "trans": trans, "wave_insts_per_ray": wave_insts / RAYS,
                                 "share_of_valu": wave_insts / pmc["SQ_INSTS_VALU"], "lane_flop_per_ray": lane_flop / RAYS,
                                 "lane_utilization_used": util}
        lines += ["", "### executed instruction mix (wave-instructions per ray)", "",
                  f"* f64: add {add/RAYS:.1f}, mul {mul/RAYS:.1f}, fma {fma/RAYS:.1f}, trans {trans/RAYS:.2f} = {wave_insts/RAYS:.1f} "
                  f"({100*wave_insts/pmc['SQ_INSTS_VALU']:.1f} % of the VALU stream)"]
        f32 = [pmc.get(f"SQ_INSTS_VALU_{k}_F32", 0.0) for k in ("ADD", "MUL", "FMA", "TRANS")]
        lines.append(f"* f32: add {f32[0]/RAYS:.1f}, mul {f32[1]/RAYS:.1f}, fma {f32[2]/RAYS:.1f}, trans {f32[3]/RAYS:.2f}")
        lines.append(f"* int32 {pmc.get('SQ_INSTS_VALU_INT32',0)/RAYS:.1f}, int64 {pmc.get('SQ_INSTS_VALU_INT64',0)/RAYS:.1f}, "
                     f"cvt {pmc.get('SQ_INSTS_VALU_CVT',0)/RAYS:.1f}, LDS atomics {pmc.get('SQ_INSTS_LDS_ATOMIC',0)/RAYS:.2f}")
        # Mix-aware issue bound: issue cycles per ray = sum over instruction classes of (count x cycles per wave64 instruction).
        # Prices (cycles per wave-instruction per SIMD with 4 waves per SIMD), with where each comes from:
        #   f64 add / mul / fma     4.6   tools/ubench/inst_rate.hip on MI355X (4.36-4.69; MI355X_MICROARCH.md's 16 lanes/clk gives 4)
        #   f64 rcp / rsq          16     inst_rate.hip (v_rcp_f64 + one add: 20.5)
        #   v_mad_u64_u32           7     inst_rate.hip (multiply + xor: 9.05)
        #   f32 add / mul / fma     4.6 in the binning kernels, whose f32 arithmetic is PACKED (v_pk_fma_f32 4.75, v_pk_mul_f32 4.48:
        #                           inst_rate.hip, round 4 -- a packed instruction does two lanes' worth of work per lane at the f64
        #                           rate; rounds 2-3 priced it like a scalar one, at 2, which halved these kernels' fraction);
        #                           2.35 elsewhere (scalar v_fma_f32: inst_rate.hip 2.35; the guide's cycle constants say 2)
        #   f32 transcendental      4     MI355X_MICROARCH.md (cycle constants: 8 for one wave alone, 4 shared)
        #   conversions             4     MI355X_MICROARCH.md (v_cvt_pk 4-5)
        #   everything else         2     32-bit integer / compare / select / move / permute: MI355X_MICROARCH.md (v_add/v_fma_f32 2 cycles with
        #                                 several waves per SIMD); inst_rate.hip: 1.7 (compare + select + add) to 2.3
        f32n = f32[0] + f32[1] + f32[2]
        i64, cvt, i32 = pmc.get("SQ_INSTS_VALU_INT64", 0.0), pmc.get("SQ_INSTS_VALU_CVT", 0.0), pmc.get("SQ_INSTS_VALU_INT32", 0.0)
        other = max(0.0, pmc["SQ_INSTS_VALU"] - (wave_insts + f32n + f32[3] + i64 + cvt + i32))
        packed = KERNEL.startswith("isx_bin_")
        price = {"f64": 4.6, "f64_trans": 16.0, "f32_arith": 4.6 if packed else 2.35, "f32_trans": 4.0, "int64_mad": 7.0, "cvt": 4.0, "other_32bit": 2.0}
        mix_cycles = (price["f64"] * (add + mul + fma) + price["f64_trans"] * trans + price["f32_arith"] * f32n + price["f32_trans"] * f32[3] +
                      price["int64_mad"] * i64 + price["cvt"] * cvt + price["other_32bit"] * (i32 + other))
        # the same mix at the guide's prices ("spec": 78.6 TFLOP/s vector FP64 = one f64 wave-instruction per 4 cycles; packed f32 likewise)
        spec = dict(price, f64=4.0, f32_arith=4.0 if packed else 2.0)
        spec_cycles = (spec["f64"] * (add + mul + fma) + spec["f64_trans"] * trans + spec["f32_arith"] * f32n + spec["f32_trans"] * f32[3] +
                       spec["int64_mad"] * i64 + spec["cvt"] * cvt + spec["other_32bit"] * (i32 + other))
        summ["issue_mix"] = {"cycles_per_ray_spec": spec_cycles / RAYS, "cycles_spec": spec, "cycles_per_ray": mix_cycles / RAYS, "uniform_4_cycles_per_ray": 4.0 * pmc["SQ_INSTS_VALU"] / RAYS,
                             "unclassified_valu_per_ray": other / RAYS, "cycles": price, "f32_arithmetic_is_packed": packed,
                             "price_sources": {"f64": "tools/ubench/inst_rate.hip (4.36-4.69)", "f64_trans": "inst_rate.hip", "int64_mad": "inst_rate.hip",
                                               "f32_arith": "inst_rate.hip (v_pk_fma_f32 4.75 / v_pk_mul_f32 4.48 packed, v_fma_f32 2.35 scalar)",
                                               "f32_trans": "MI355X_MICROARCH.md cycle constants", "cvt": "MI355X_MICROARCH.md cycle constants",
                                               "other_32bit": "MI355X_MICROARCH.md cycle constants (2); inst_rate.hip 1.7-2.3"}}
        lines.append(f"* mix-aware issue cycles per ray = {mix_cycles/RAYS:.1f} (f64 {price['f64']}, f64 rcp/rsq 16, v_mad_u64_u32 7, f32 arithmetic "
                     f"{price['f32_arith']}{' (packed)' if packed else ''}, cvt 4, f32 trans 4, every other VALU instruction 2; {other/RAYS:.1f} "
                     f"unclassified per ray) against {4.0*pmc['SQ_INSTS_VALU']/RAYS:.1f} with 4 cycles for all")
        if big:
            t = sum(big) / len(big) * 1e-3
            for label, clk in (("2.4 GHz", 2.4e9), ("the measured clock", summ.get("clock_ghz", 2.4) * 1e9)):
                lines.append(f"  * fraction of the mix-aware issue peak at {label}: {mix_cycles / (1024 * clk * t):.3f}")
            summ["issue_mix"]["frac_at_2p4GHz"] = mix_cycles / (1024 * 2.4e9 * t)
            summ["issue_mix"]["frac_spec_at_2p4GHz"] = spec_cycles / (1024 * 2.4e9 * t)
            lines.append(f"  * at the guide's prices (f64 and packed f32 4 cycles): {spec_cycles/RAYS:.1f} cycles per ray, fraction {spec_cycles / (1024 * 2.4e9 * t):.3f} at 2.4 GHz")
            summ["issue_mix"]["frac_at_measured_clock"] = mix_cycles / (1024 * summ.get("clock_ghz", 2.4) * 1e9 * t)
        if big:
            tf = lane_flop / (sum(big) / len(big) * 1e-3) / 1e12
            summ["fp64_executed"]["tflops"] = tf
            lines.append(f"* executed FP64 = {lane_flop/RAYS:.0f} lane-flop per ray (fma = 2, x64 lanes x lane utilisation {util:.3f}) "
                         f"= {tf:.2f} TFLOP/s = {tf/78.6:.3f} of the 78.6 TFLOP/s vector FP64 peak")
    summ["counters"] = pmc
    json.dump(summ, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.json"), "w"), indent=1)
    if a.headline:
        json.dump(summ, open(os.path.join(ROOT, "profiles", "pmc_summary.json"), "w"), indent=1)

open(os.path.join(ROOT, "profiles", f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
