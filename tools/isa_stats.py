#!/usr/bin/env python3
"""Per-kernel ISA census of libisx (no GPU needed): VGPRs, scratch, and how every kernel reaches memory.

usage: tools/isa_stats.py [--check] [--keep DIR] [extra -D flags for hipcc]

Compiles csrc/isx_api.hip for gfx950 with --save-temps and counts, per kernel, flat_ / global_ / scratch_ / ds_
instructions.  --check fails (exit 1) on the traps DESIGN.md section 5 describes:
  * a `flat_*` access in ANY kernel (a pointer whose address space hipcc could not infer: flat operations complete
    out of order and force s_waitcnt vmcnt(0) lgkmcnt(0) together, so every line fetch also drains the wave's LDS
    queue);
  * any `flat_* ... sc0 sc1` (a volatile access through a generic pointer: system scope);
  * scratch in a kernel of NO_SCRATCH (the binning kernels and the trace kernels of the BASELINE configurations).
Scratch of every other kernel is printed as a report ("SCRATCH ..."), so that a spill inside a loop is seen at build time.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "altair-raytracing_amd", "csrc", "isx_api.hip")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-function"]
# kernels that must not spill: the binning kernels and the trace kernels the BASELINE configurations run
NO_SCRATCH = ("isx_bin_cols_kernel", "isx_bin_slots_kernel", "isx_bin_lines_kernel", "isx_bin_discs_kernel",
              "isx_trace_assist_kernel", "isx_trace_assist_brdf_kernel", "isx_trace_assist_chord_kernel",
              "isx_trace_assist_perpos_kernel", "isx_trace_assist_lobe_kernel", "isx_trace_assist_rough_kernel")


def compile_isa(extra, keep=None):
    d = keep or tempfile.mkdtemp(prefix="isx_isa_")
    os.makedirs(d, exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, *extra, "--save-temps", "-c", "-o", os.path.join(d, "isx_api.o"), SRC], cwd=d,
                          stderr=subprocess.DEVNULL)
    return os.path.join(d, "isx_api-hip-amdgcn-amd-amdhsa-gfx950.s")


def kernels(path):
    text = open(path).read()
    for part in re.split(r"\n(?=\w+:\s*; @)", text):
        m = re.match(r"(\w+):", part)
        if m and m.group(1).startswith("isx_"):
            yield m.group(1), part


def census(body):
    def n(pat):
        return len(re.findall(pat, body, re.M))

    def meta(key):
        m = re.search(r"; %s: (\d+)" % key, body)
        return int(m.group(1)) if m else -1

    return {
        "vgpr": meta("NumVgprs"), "sgpr": meta("NumSgprs"), "scratch": meta("ScratchSize"), "occ": meta("Occupancy"),
        "flat_ld": n(r"^\s*flat_load"), "flat_st": n(r"^\s*flat_store"), "flat_at": n(r"^\s*flat_atomic"),
        "flat_sys": n(r"^\s*flat_.*sc0 sc1"),
        "glob_ld": n(r"^\s*global_load"), "glob_st": n(r"^\s*global_store"), "glob_at": n(r"^\s*global_atomic"),
        "scr_ld": n(r"^\s*scratch_load"), "scr_st": n(r"^\s*scratch_store"),
        "ds": n(r"^\s*ds_"), "lines": body.count("\n"),
    }


# kernels that may keep scratch OUTSIDE their tracer waves' bounce steps only (the assist wave carries Ray::prev through the generic search)
SCRATCH_OUTSIDE_STEPS = ("isx_trace_assist_disc_kernel", "isx_trace_assist_discpos_kernel")


def scratch_in_steps(body):
    """scratch instructions between the first and the last Philox block of the TRACER loop (the last four runs of >= 15
    v_mad_u64_u32: the bounce steps of a trip; the run before them belongs to the assist wave)"""
    lines = body.splitlines()
    mads = [i for i, l in enumerate(lines) if "v_mad_u64_u32" in l]
    blocks, cur = [], mads[:1]
    for m in mads[1:]:
        if m - cur[-1] < 12:
            cur.append(m)
        else:
            if len(cur) >= 15:
                blocks.append((cur[0], cur[-1]))
            cur = [m]
    if len(cur) >= 15:
        blocks.append((cur[0], cur[-1]))
    if len(blocks) < 5:
        return -1
    a, b = blocks[-4][0] - 120, blocks[-1][1] + 300      # (one step before the first block's Philox, one step after the last)
    return sum(1 for l in lines[a:b] if re.match(r"\s*scratch_", l))


def main():
    args = sys.argv[1:]
    check = "--check" in args
    keep = None
    if "--keep" in args:
        keep = args[args.index("--keep") + 1]
        args = [a for i, a in enumerate(args) if a != "--keep" and (i == 0 or args[i - 1] != "--keep")]
    extra = [a for a in args if a != "--check"]
    path = compile_isa(extra, keep)
    bad, report = [], []
    print("%-36s %4s %4s %7s %3s | %7s %7s %7s %4s | %7s %7s %7s | %6s %6s | %5s" % (
        "kernel", "vgpr", "sgpr", "scratch", "occ", "flat_ld", "flat_st", "flat_at", "sys", "glob_ld", "glob_st", "glob_at", "scr_ld", "scr_st", "ds"))
    for name, body in kernels(path):
        c = census(body)
        print("%-36s %4d %4d %7d %3d | %7d %7d %7d %4d | %7d %7d %7d | %6d %6d | %5d" % (
            name, c["vgpr"], c["sgpr"], c["scratch"], c["occ"], c["flat_ld"], c["flat_st"], c["flat_at"], c["flat_sys"],
            c["glob_ld"], c["glob_st"], c["glob_at"], c["scr_ld"], c["scr_st"], c["ds"]))
        if c["flat_sys"]:
            bad.append("%s: %d system-scope flat accesses" % (name, c["flat_sys"]))
        if c["flat_ld"] + c["flat_st"] + c["flat_at"]:
            bad.append("%s: %d flat_ accesses" % (name, c["flat_ld"] + c["flat_st"] + c["flat_at"]))
        if name in SCRATCH_OUTSIDE_STEPS and c["scratch"] > 0:
            inside = scratch_in_steps(body)
            if inside != 0:
                bad.append("%s: %s scratch instructions inside the tracers' bounce steps" % (name, "unlocatable" if inside < 0 else inside))
            else:
                report.append("SCRATCH %s: none of its %d scratch instructions lies in the tracers' bounce steps" % (name, c["scr_ld"] + c["scr_st"]))
        if c["scratch"] > 0:
            if name in NO_SCRATCH:
                bad.append("%s: %d B of scratch (%d loads, %d stores)" % (name, c["scratch"], c["scr_ld"], c["scr_st"]))
            else:
                report.append("SCRATCH %s: %d B (%d loads, %d stores)" % (name, c["scratch"], c["scr_ld"], c["scr_st"]))
    if report:
        print("\n".join(report))
    if check and bad:
        print("\n".join("TRAP " + b for b in bad))
        sys.exit(1)


if __name__ == "__main__":
    main()
