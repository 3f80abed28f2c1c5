#!/usr/bin/env python3
"""bench.py — the reference's headline metric on MI355X: Mrays/s for the 180x90 flux map at
src(-60,0,-75), port 170 deg (BASELINE.json: metric / configs[1]: 5e7 rays per GPU).

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one batch: every rank traces `--rays` rays (its own
contiguous slice of the global ray-index stream of that step, so results do not depend on N),
bins them into the 180x90 detector histogram on its GPU, and the histograms are summed with
ONE all-reduce over RCCL (torch.distributed backend "nccl").  Weak scaling: per-GPU work is
fixed -- 5e7 rays per GPU per step for EVERY N, so the N = 1 point of a scaling curve is the
single-GPU bench.  At N = 8 two more timed loops run BASELINE configs[4] (1e9 rays per step over
the node: "configs4") and configs[3] (integratingSphereDetectorSweep.C's 362 disc positions share
1e7 rays per GPU and step, ray-sharded: "configs3"); extra keys only.  Prints one JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X vector FP64 peak
VALU_CLOCK_GHZ = 2.4           # MI355X peak engine clock (78.6 TFLOP/s = 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz)
# Work model frozen in BASELINE.md §2 / DESIGN.md §5 (brute-force binning convention):
F_BOUNCE, N_BOUNCE, P_EXIT, F_DISC = 200.0, 57.5, 0.4235, 40.0


def kernel_source_sha():
    """Fingerprint of the kernel sources libisx.so is built from: a committed PMC summary is only used for the roofline if
    it was measured on THIS code (tools/summarize_profile.py stores the same fingerprint)."""
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, "altair-raytracing_amd", "csrc")
    for name in ("isx_device.hpp", "isx_kernels.hpp", "isx_api.hip"):
        with open(os.path.join(base, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rays", type=int, default=0,
                    help="rays per GPU per step; default 5e7 (BASELINE configs[1]) for every --gpus")
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EED0001)
    ap.add_argument("--cpu-rays", type=int, default=-1, help="oracle sample size for cpu_baseline (0: skip, -1: auto)")
    ap.add_argument("--trace-mode", choices=["explicit", "chord"], default="explicit",
                    help="explicit: cosine direction + ray-sphere intersection per bounce (default, the reference-shaped "
                         "algorithm); chord: ISX_TRACE_CHORD, next wall point sampled directly (same distribution)")
    ap.add_argument("--reduce", choices=["auto", "device", "host"], default="auto",
                    help="where the histogram lives for the all-reduce")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra timed legs of the N = 1 line (configs0 / configs2 / configs3 / perpos_8p1e8 / size_sweep / surfaces)")
    return ap.parse_args()


def default_rays_per_gpu(world):
    """5e7 rays per GPU per step (BASELINE configs[1]) whatever the number of GPUs: the points of a scaling curve must be
    the same per-GPU workload (tests/test_host_driver.py asserts it)."""
    del world
    return 50_000_000


# the reference's own number for this metric's workload (per-position 180x90 map, 8.1e8 rays in
# 12 523.9 s on <= 4 ROBAST threads of an unknown CPU) -- quoted in the bench line so that it is self-contained
REFERENCE_PUBLISHED = {"value": 0.0647, "unit": "Mrays/s", "what": "fluxAtObserverOptimize.C sweepDetector, 8.1e8 rays in 12523.9 s, <= 4 ROBAST "
                       "threads, CPU unknown (flux_at_observer/results_overnight_03_31-60_0_-75_5/fluxmap_50000rays_180x90_src-60_0_-75.csv:16217)"}


def cpu_baseline(seed, n_req):
    """Oracle (CPU restatement, kind 'port') timed on this host's cores on a bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    cfg = oracle.default_config()
    # the GPU box gives one GPU's job a 16-core share of the host: never spawn more workers than that
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(avail, oracle.lib().isxo_max_threads(), int(os.environ.get("ISX_CPU_THREADS", "16"))))
    if n_req < 0:
        # calibrate: ~15 s of CPU work
        t0 = time.time()
        oracle.fluxmap(cfg, 100_000, seed, 0, threads)
        dt = max(time.time() - t0, 1e-3)
        n_req = int(min(max(100_000 * 15.0 / dt, 200_000), 20_000_000))
    t0 = time.time()
    _, st = oracle.fluxmap(cfg, n_req, seed, 0, threads)
    dt = time.time() - t0
    return {"value": n_req / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port", "reference_published": REFERENCE_PUBLISHED,
            "sample": f"{n_req} rays of the same workload (180x90 map, src(-60,0,-75), port 170deg), "
                      f"oracle/libisx_oracle.so, OpenMP {threads} threads, {dt:.1f} s"}


def load_pmc():
    """profiles/pmc_summary.json (tools/summarize_profile.py) if it was measured on THIS kernel code -> (dict, note)"""
    pj, note = {}, "no profiles/pmc_summary.json"
    pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")
    if os.path.exists(pmc):
        try:
            pj = json.load(open(pmc))
            sha = kernel_source_sha()
            if pj.get("kernel_source_sha") != sha:
                note = (f"STALE: profiles/pmc_summary.json (tag {pj.get('tag')}) was measured on kernel sources "
                        f"{pj.get('kernel_source_sha')}, this build is {sha}: executed-instruction rooflines omitted")
                pj = {}
            else:
                note = f"profiles/pmc_summary.json tag {pj.get('tag')}, kernel sources {sha}"
        except Exception as e:  # a broken file is reported, not fatal
            pj, note = {}, f"unreadable profiles/pmc_summary.json: {e}"
    return pj, note


def issue_block(kernel, live_ms, pk, n, cus):
    """VALU-issue roofline of one kernel: executed SQ_INSTS_VALU per ray (its PMC pass) x rays / its LIVE time.
    `peak` / `frac` price the executed mix by class (`cycle_prices`, each with its source in `cycle_price_sources`: the
    guide's cycle constants or tools/ubench/inst_rate.hip measured on MI355X; a PACKED f32 instruction costs what an f64 one
    does) against 1024 SIMDs x 2.4 GHz -- `peak` is the wave-instruction rate this mix could issue at best, `frac` = achieved /
    peak (= `frac_mix`); `frac_spec` prices f64 and packed f32 at the guide's 4 cycles (78.6 TFLOP/s vector FP64) instead of the
    4.6 measured; `frac_4cycle` / `peak_4cycle` charge every instruction 4 cycles (the convention of rounds 1-2: a
    mixed stream can exceed it); `valu_idle` says whether the VALU had idle cycles at all (`valu_busy_counter_ratio` is
    4 x SQ_ACTIVE_INST_VALU / SIMD-cycles: overlapping issue is counted more than once, so it exceeds 1 on a saturated
    kernel -- a counter ratio, not a fraction)."""
    peak_issue = cus * 4 * VALU_CLOCK_GHZ / 4.0
    blk = {"bound": "valu_issue", "kernel": kernel, "kernel_ms": live_ms, "peak": peak_issue, "unit": "G wave-instr/s",
           "achieved": None, "frac": None, "traffic": (pk or {}).get("hbm_bytes_per_launch")}
    if pk and pk.get("valu_wave_insts_per_ray") and live_ms > 0:
        ach = pk["valu_wave_insts_per_ray"] * n / (live_ms * 1e-3) / 1e9
        blk.update(achieved=ach, frac=ach / peak_issue, valu_wave_insts_per_ray=pk["valu_wave_insts_per_ray"],
                   valu_idle=pk.get("valu_idle"), valu_busy_counter_ratio=pk.get("valu_busy_counter_ratio"),
                   valu_lane_utilization=pk.get("valu_lane_utilization"),
                   profiled_kernel_ms=pk.get("kernel_ms"), profiled_clock_ghz=pk.get("clock_ghz"))
        mix = pk.get("issue_mix")
        if mix:
            # the roofline proper: what THIS instruction mix can issue per second (peak), against what it did (achieved)
            simd_cycles = cus * 4 * VALU_CLOCK_GHZ * 1e9 * live_ms * 1e-3
            fm = mix["cycles_per_ray"] * n / simd_cycles
            blk.update(peak_mix_cycles_per_ray=mix["cycles_per_ray"], frac_mix=fm, frac_4cycle=ach / peak_issue,
                       peak_4cycle=peak_issue, peak=ach / fm, frac=fm, cycle_prices=mix.get("cycles"),
                       cycle_price_sources=mix.get("price_sources"))
            if mix.get("cycles_per_ray_spec"):
                blk.update(frac_spec=mix["cycles_per_ray_spec"] * n / simd_cycles, spec_cycles_per_ray=mix["cycles_per_ray_spec"],
                           spec_prices=mix.get("cycles_spec"))
    return blk


def disc_positions(np):
    """rootMacros::detectorDiskPlacement (integratingSphereDetectorSweep.C:145-172): 181 x 2 discs at 200 cm from the origin, tube
    axis (sin rotTheta, 0, cos rotTheta) for every phi (TGeoRotation::RotateY left-multiplies: DESIGN.md section 2.4)."""
    import math
    discs = []
    for th in np.arange(-45.0, 45.0 + 1e-9, 0.5):
        for ph in (0.0, 180.0):
            t_, p_ = math.radians(th), math.radians(ph)
            x, y, z = 200 * math.sin(t_) * math.cos(p_), 200 * math.sin(t_) * math.sin(p_), -200 * math.cos(t_)
            rot = -math.atan2(math.sqrt(x * x + y * y), -100 - z)
            discs.append([x, y, z, math.sin(rot), 0.0, math.cos(rot)])
    return np.array(discs)


def extra_legs(isx, np, seed, cus, pj):
    """The other BASELINE configurations as timed legs of the N = 1 line (VERDICT r04 'next' 2): every number is wall clock around
    blocking calls of the C ABI (host synchronisation included), next to the library's HIP-event kernel times.  Extra keys only:
    the headline loop above is untouched and runs first."""
    sections = pj.get("sections", {}) if pj else {}

    def timed(fn, reps, warm=1):
        for _ in range(warm):
            fn()
        isx.sync()
        t0 = time.perf_counter()
        ks, kinds, st = [], [], None
        for _ in range(reps):
            st = fn()
            ks.append(st.t_kernel_ms)
            kinds.append(isx.last_kernel_ms())
        wall = (time.perf_counter() - t0) * 1e3 / reps
        return wall, float(np.mean(ks)), [float(np.mean([k[i] for k in kinds])) for i in range(3)], st

    def blocks(section, kinds, n):
        sec = sections.get(section, {}).get("kernels", {})
        out = {}
        for kname, pk in sec.items():
            live = kinds[1] if "trace" in kname else kinds[2]
            out[kname] = issue_block(kname, live, pk, n, cus)
        return out or None

    legs = {}
    # ---- configs[0]: the reference's own call size (fluxAtObserverOptimize.C:568: 50 000 rays per call, 16 200 calls per map)
    c = isx.default_config()
    det = isx.detector_table(c)[90 * c.n_phi + 45]
    n0 = 50_000
    k = [0]

    def small_flux():
        k[0] += 1
        return isx.fluxmap(c, n0, seed, k[0] * n0)[1]

    def small_det():
        k[0] += 1
        return isx.trace_rays_detector(c, det, c.det_diameter, n0, seed, k[0] * n0)[1]

    w1, k1, kk1, _ = timed(small_flux, 50, 3)
    w2, k2, kk2, _ = timed(small_det, 50, 3)
    legs["configs0"] = {"workload": "BASELINE configs[0]: 50 000 rays per call, src(-60,0,-75), 180x90 grid (fluxAtObserver.C / fluxAtObserverOptimize.C:568)",
                        "isx_fluxmap": {"ms_per_call": w1, "kernel_ms": k1, "trace_ms": kk1[1], "bin_ms": kk1[2], "Mrays_s": n0 / w1 / 1e3},
                        "isx_trace_rays_detector": {"ms_per_call": w2, "kernel_ms": k2, "Mrays_s": n0 / w2 / 1e3,
                                                    "map_of_16200_calls_s": 16200 * w2 / 1e3}}
    # ---- configs[2]: nonLambertianFlux.C source model (BRDF re-scatter), 5e7 rays
    c2 = isx.default_config()
    c2.source_model = isx.SOURCE_BRDF; c2.brdf[0], c2.brdf[1], c2.brdf[2] = 0.3, 0.4, 0.6
    c2.roughness_rad = 0.5; c2.reflectance = 1.0; c2.max_points = 10000; c2.box_half = 200.0
    n2 = 50_000_000
    w, km, kk, st = timed(lambda: isx.fluxmap(c2, n2, seed)[1], 3)
    legs["configs2"] = {"workload": "BASELINE configs[2]: nonLambertianFlux.C:235-304 source model (primary trace, BRDF re-scatter, second trace), 5e7 rays, "
                                    "same geometry and grid", "rays": n2, "value": n2 / w / 1e3, "unit": "Mrays/s", "ms_per_call": w, "kernel_ms": km,
                        "trace_ms": kk[1], "bin_ms": kk[2], "wall_hits_per_ray": st.wall_hits / n2, "exit_lines_per_ray": st.counted_below_z / n2,
                        "issue_blocks": blocks("brdf", kk, n2)}
    # ---- configs[3]: integratingSphereDetectorSweep.C:31-105, 362 disc positions -- sharing 1e7 rays, and the macro's own loop
    c3 = isx.default_config()
    c3.r_out = 105.0; c3.reflectance = 1.0; c3.roughness_rad = 0.0; c3.max_points = 10000; c3.box_half = 200.0
    c3.src[2] = -80.0
    discs = disc_positions(np)
    n3 = 10_000_000
    w, km, kk, st = timed(lambda: isx.disc_sweep(c3, discs, 5.0, 0.1, n3, 7)[1], 3)
    rpp = 1_000_000
    wp, kmp, kkp, stp = timed(lambda: isx.disc_sweep_per_position(c3, discs, 5.0, 0.1, rpp, 7)[1], 2)
    legs["configs3"] = {"workload": "BASELINE configs[3]: integratingSphereDetectorSweep.C, 362 disc positions (r 5 cm, 200 cm from the origin), shell 100.1-105 cm, rho 1",
                        "shared_rays": {"rays": n3, "value": n3 / w / 1e3, "unit": "Mrays/s", "ms_per_call": w, "kernel_ms": km, "trace_ms": kk[1],
                                        "bin_ms": kk[2], "wall_hits_per_ray": st.wall_hits / n3, "issue_blocks": blocks("discs", kk, n3)},
                        "per_position_loop": {"rays": rpp * len(discs), "rays_per_position": rpp, "value": rpp * len(discs) / wp / 1e3, "unit": "Mrays/s",
                                              "ms_per_call": wp, "kernel_ms": kmp}}
    # ---- the reference's 12 524 s run: 50 000 fresh rays per detector position, 16 200 positions (fluxAtObserverOptimize.C:542-579)
    c = isx.default_config()
    w, km, kk, st = timed(lambda: isx.fluxmap_per_position(c, 50_000, seed)[1], 2)
    legs["perpos_8p1e8"] = {"workload": "fluxAtObserverOptimize.C:542-579: 50 000 fresh rays per position x 16 200 positions (the reference: 12 523.9 s)",
                            "rays": 50_000 * 16200, "value": 50_000 * 16200 / w / 1e3, "unit": "Mrays/s", "ms_per_call": w, "kernel_ms": km,
                            "speedup_vs_reference_published": 12523.937 / (w / 1e3)}
    # ---- size sweep of the headline map (north_star: "50k -> 1e9-ray runs"): one blocking isx_fluxmap call per size
    sweep = []
    for n, reps in ((50_000, 30), (500_000, 20), (5_000_000, 8), (50_000_000, 3), (1_000_000_000, 1)):
        w, km, kk, st = timed(lambda: isx.fluxmap(c, n, seed)[1], reps, 1 if n < 10 ** 9 else 0)
        sweep.append({"rays": n, "Mrays_s": n / w / 1e3, "ms_per_call": w, "kernel_ms": km, "trace_ms": kk[1], "bin_ms": kk[2]})
    legs["size_sweep"] = sweep
    # ---- the other border models / hit line on the same pipeline ("nonLambertianFlux copy.C":31-70,188-221)
    surf = {}
    for name, setup, sec in (("lobe_rho0p99", lambda q: setattr(q, "surface_model", 1), "lobe"),
                             ("rough_specular_sigma0p5", lambda q: (setattr(q, "lambertian", 0), setattr(q, "roughness_rad", 0.5)), "rough"),
                             ("origin_compat_hit_line", lambda q: setattr(q, "hit_line_mode", 1), None)):
        q = isx.default_config()
        setup(q)
        ns = 50_000_000
        w, km, kk, st = timed(lambda: isx.fluxmap(q, ns, seed)[1], 2)
        surf[name] = {"rays": ns, "value": ns / w / 1e3, "unit": "Mrays/s", "ms_per_call": w, "kernel_ms": km, "trace_ms": kk[1], "bin_ms": kk[2],
                      "wall_hits_per_ray": st.wall_hits / ns, "bin_increments_per_ray": st.bin_increments / ns,
                      "issue_blocks": blocks(sec, kk, ns) if sec else None}
    # "nonLambertianFlux copy.C":306-345 as the macro runs it: 45 x 20 detector positions x 1e5 fresh rays with the lobe border (rho 1,
    # limit 10 000, box 200, src z -80, 10 cm detector) -- one launch of the per-position lobe kernel
    q = isx.default_config()
    q.surface_model = 1; q.lambertian = 0; q.roughness_rad = 0.5; q.reflectance = 1.0; q.max_points = 10000; q.box_half = 200.0
    q.src[2] = -80.0; q.n_theta, q.n_phi, q.det_diameter = 45, 20, 10.0
    w, km, kk, st = timed(lambda: isx.fluxmap_per_position(q, 100_000, seed)[1], 1)
    surf["nlcopy_macro_sweep"] = {"workload": "nonLambertianFluxCopy::sweepDetector: 900 positions x 1e5 rays, lobe border, rho 1", "rays": 90_000_000,
                                  "value": 90_000_000 / w / 1e3, "unit": "Mrays/s", "ms_per_call": w, "kernel_ms": km, "wall_hits_per_ray": st.wall_hits / 9e7}
    legs["surfaces"] = surf
    # ---- the optional chord mode of the headline configuration (ISX_TRACE_CHORD: the integrating-sphere identity samples the next wall
    # point directly; same distribution, the oracle has the twin; the headline above keeps the reference-shaped explicit bounce)
    q = isx.default_config()
    q.trace_mode = 1
    w, km, kk, st = timed(lambda: isx.fluxmap(q, 50_000_000, seed)[1], 3)
    legs["chord_mode"] = {"rays": 50_000_000, "value": 50_000_000 / w / 1e3, "unit": "Mrays/s", "ms_per_call": w, "kernel_ms": km,
                          "trace_ms": kk[1], "bin_ms": kk[2]}
    return legs


def main():
    a = parse()
    # RCCL (NCCL_DEBUG=VERSION on some boxes) and HIP print banners on fd 1; the contract is ONE JSON line on
    # stdout, so everything else is routed to stderr and the real stdout is kept for the final print.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:   # (before any GPU call: a bare `bench.py --gpus 4` would silently measure one GPU)
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch through python -m torch.distributed.run "
                         f"--nnodes=1 --nproc-per-node {a.gpus} --master-addr 127.0.0.1 bench.py --gpus {a.gpus} ...")

    import numpy as np
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists in the product path)")
    # one GPU per rank; if the launcher restricted this rank's visible devices, fall back to what is visible
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # ISX_FORCE_DIST=1 exercises the RCCL code path even with one rank (rehearsal on a 1-GPU box)
    use_dist = world > 1 or os.environ.get("ISX_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        # ISX_BENCH_BACKEND=gloo: rehearsal of the N > 1 path with several ranks on ONE GPU (RCCL refuses two ranks on a device);
        # the driver's launches never set it
        backend = os.environ.get("ISX_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import altair_raytracing_amd as isx
    isx.load()
    isx.init(dev_index)
    devname, cus = isx.device_info()
    cfg = isx.default_config()
    cfg.trace_mode = 1 if a.trace_mode == "chord" else 0
    nb = cfg.n_theta * cfg.n_phi
    n = a.rays if a.rays > 0 else default_rays_per_gpu(world)
    which = "BASELINE configs[1]"

    hist_dev = torch.zeros(nb, dtype=torch.int64, device=dev)

    def barrier():
        if use_dist:
            if dist.get_backend() == "nccl":
                dist.barrier(device_ids=[dev_index])
            else:
                dist.barrier()
        torch.cuda.synchronize()

    # --- ONE HIP runtime per process: torch's wheel bundles libamdhip64.so with the SONAME libisx.so asks for
    # (libamdhip64.so.7), and torch is imported (and has initialised the device) BEFORE libisx is loaded, so the dynamic
    # linker hands libisx the runtime that is already mapped.  Checked, not assumed: count the distinct libamdhip64 images
    # in this process.  With one runtime a torch tensor's data_ptr() is a device pointer libisx can write to, so the
    # histogram stays on the device for the all-reduce; with two (never seen) it would bounce 130 KB through the host.
    with open("/proc/self/maps") as f:
        hip_images = sorted({ln.split()[-1] for ln in f if "libamdhip64" in ln})
    mode = a.reduce
    if mode == "auto" and len(hip_images) != 1:
        mode = "host"
    if mode == "auto":
        try:
            probe = torch.zeros(nb, dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            isx.fluxmap_device(cfg, 4096, a.seed, 0, probe.data_ptr())
            isx.sync()
            isx.take_stats()
            want, _ = isx.fluxmap(cfg, 4096, a.seed, 0)
            torch.cuda.synchronize()
            mode = "device" if np.array_equal(probe.cpu().numpy().astype(np.uint64), want.reshape(-1)) else "host"
        except Exception:
            mode = "host"
    if use_dist:
        flag = torch.tensor([1 if mode == "device" else 0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        mode = "device" if int(flag.item()) == 1 else "host"

    # the library's own stream as a torch stream: zeroing the histogram and the all-reduce are enqueued ON it, between the
    # kernels of consecutive steps, so that a timed loop needs no host synchronisation between its steps (the host round trip
    # per step -- zero, synchronise, launch, synchronise, read the census -- left the GPU idle for 0.25 of 17.9 ms)
    lib_stream = torch.cuda.ExternalStream(isx.load().isx_stream(), device=dev) if mode == "device" else None

    def timed_loop(n_rays, first_step, warmup, steps):
        """`warmup` untimed + `steps` timed steps of n_rays rays per GPU -> (seconds = max over ranks, per-step records).
        Device path: the steps of the timed region are enqueued back to back on the library's stream (zero the histogram, trace,
        bin, all-reduce), the census is read twice -- after the last step but one (the sum of the steps so far) and after the
        last one (that step's own) -- and the kernel times come from the library's HIP events around every launch."""
        rec = {"kernel_ms": [], "kind_ms": [], "census": [], "allreduce_ms": []}
        ar_events = []

        def enqueue(s):
            """trace + bin this rank's slice of step s, then all-reduce the 180x90 histogram: enqueued, not waited for (device path)"""
            first, _ = isx.step_slice(s, rank, world, n_rays)
            if mode == "device":
                with torch.cuda.stream(lib_stream):
                    hist_dev.zero_()
                isx.fluxmap_device(cfg, n_rays, a.seed, first, hist_dev.data_ptr())
                if use_dist:
                    with torch.cuda.stream(lib_stream):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        dist.all_reduce(hist_dev, op=dist.ReduceOp.SUM)
                        e1.record()
                    ar_events.append((e0, e1))
                return None
            h, st = isx.fluxmap(cfg, n_rays, a.seed, first)
            hist_dev.copy_(torch.from_numpy(h.reshape(-1).astype(np.int64)))
            if use_dist:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                dist.all_reduce(hist_dev, op=dist.ReduceOp.SUM)
                e1.record()
                ar_events.append((e0, e1))
                torch.cuda.synchronize()
            return st

        def collect(n_steps, st_host):
            """census + kernel times of the n_steps steps enqueued since the last collect (host path: st_host = their census)"""
            if mode == "device":
                st = isx.take_stats()          # synchronises the library's stream
            else:
                st = st_host
            kinds = isx.last_kernel_ms()
            rec["kind_ms"] += [tuple(k / n_steps for k in kinds)] * n_steps
            rec["kernel_ms"] += [st.t_kernel_ms / n_steps] * n_steps
            rec["census"].append((n_steps, st))

        for s in range(first_step, first_step + warmup):
            st = enqueue(s)
            collect(1, st)
        torch.cuda.synchronize()
        for v in rec.values():
            v.clear()
        ar_events.clear()
        barrier()
        t0 = time.perf_counter()
        last = first_step + warmup + steps - 1
        if mode == "device":
            for s in range(first_step + warmup, last):
                enqueue(s)
            if steps > 1:
                collect(steps - 1, None)
            enqueue(last)
            collect(1, None)
        else:
            for s in range(first_step + warmup, last + 1):
                st = enqueue(s)
                collect(1, st)
        barrier()
        dt = time.perf_counter() - t0
        rec["allreduce_ms"] = [e0.elapsed_time(e1) for e0, e1 in ar_events]
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()), rec

    def over_ranks(values):
        """[min, max] over the ranks of this rank's mean of `values` (what a bad scaling point would be diagnosed from)"""
        v = torch.tensor([float(np.mean(values)) if len(values) else 0.0], dtype=torch.float64, device=dev)
        if not use_dist:
            return [float(v.item())] * 2
        lo, hi = v.clone(), v.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        return [float(lo.item()), float(hi.item())]

    dt, rec = timed_loop(n, 0, a.warmup, a.steps)
    kernel_ms, census, kind_ms = rec["kernel_ms"], rec["census"], rec["kind_ms"]
    per_rank = {"kernel_ms": over_ranks(kernel_ms), "trace_ms": over_ranks([k[1] for k in kind_ms]),
                "bin_ms": over_ranks([k[2] for k in kind_ms]), "allreduce_ms": over_ranks(rec["allreduce_ms"])}
    total_hits = int(hist_dev.sum().item())
    # BASELINE configs[4] (1e9 rays per step over the 8 GPUs of a node) as a second timed loop at N = 8; extra keys only
    configs4 = None
    if world == 8 and a.rays <= 0:
        n4 = 125_000_000
        dt4, rec4 = timed_loop(n4, a.warmup + a.steps, 1, a.steps)
        configs4 = {"workload": "BASELINE configs[4]: 1e9 rays per step over 8 GPUs (1.25e8 per GPU), same map", "rays_per_gpu_per_step": n4,
                    "value": n4 * world * a.steps / dt4 / 1e6, "unit": "Mrays/s", "ms_per_step": dt4 / a.steps * 1e3,
                    "kernel_ms_min_max_over_ranks": over_ranks(rec4["kernel_ms"]),
                    "allreduce_ms_min_max_over_ranks": over_ranks(rec4["allreduce_ms"])}
    # BASELINE configs[3] (integratingSphereDetectorSweep.C:31-105, ray-sharded): the 181 x 2 disc positions of the macro share the
    # rays -- every rank traces its slice of the step's rays against ALL discs, ONE all-reduce of the 362 counts.  At N = 8
    # (ISX_BENCH_CONFIGS3=1: at any N, rehearsal); extra keys only.
    configs3 = None
    if (world == 8 and a.rays <= 0) or os.environ.get("ISX_BENCH_CONFIGS3") == "1":
        import math
        discs = []
        for th in np.arange(-45.0, 45.0 + 1e-9, 0.5):   # rootMacros::detectorDiskPlacement (integratingSphereDetectorSweep.C:145-172)
            for ph in (0.0, 180.0):
                t_, p_ = math.radians(th), math.radians(ph)
                x, y, z = 200 * math.sin(t_) * math.cos(p_), 200 * math.sin(t_) * math.sin(p_), -200 * math.cos(t_)
                rot = -math.atan2(math.sqrt(x * x + y * y), -100 - z)
                discs.append([x, y, z, math.sin(rot), 0.0, math.cos(rot)])
        discs = np.array(discs)
        c3 = isx.default_config()
        c3.r_out = 105.0; c3.reflectance = 1.0; c3.roughness_rad = 0.0; c3.max_points = 10000; c3.box_half = 200.0
        c3.src[2] = -80.0
        n3 = int(os.environ.get("ISX_BENCH_CONFIGS3_RAYS", "10000000"))
        disc_dev = torch.zeros(len(discs), dtype=torch.int64, device=dev)
        k3, ar3 = [], []

        def step3(s):
            first, _ = isx.step_slice(s, rank, world, n3)
            h, st3 = isx.disc_sweep(c3, discs, 5.0, 0.1, n3, 7, first)
            k3.append(st3.t_kernel_ms)
            disc_dev.copy_(torch.from_numpy(h.reshape(-1).astype(np.int64)))
            if use_dist:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                dist.all_reduce(disc_dev, op=dist.ReduceOp.SUM)
                e1.record()
                torch.cuda.synchronize()
                ar3.append(e0.elapsed_time(e1))

        step3(0)
        k3.clear(); ar3.clear()
        barrier()
        t0 = time.perf_counter()
        for s in range(1, 1 + a.steps):
            step3(s)
        barrier()
        t3 = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(t3, op=dist.ReduceOp.MAX)
        dt3 = float(t3.item())
        configs3 = {"workload": "BASELINE configs[3]: integratingSphereDetectorSweep.C, 362 disc positions (r 5 cm, 200 cm from the origin) share "
                                f"{n3} rays per GPU and step, shell 100.1-105 cm, rho 1; one all-reduce of the 362 counts",
                    "rays_per_gpu_per_step": n3, "n_discs": int(len(discs)), "value": n3 * world * a.steps / dt3 / 1e6, "unit": "Mrays/s",
                    "ms_per_step": dt3 / a.steps * 1e3, "kernel_ms_min_max_over_ranks": over_ranks(k3),
                    "allreduce_ms_min_max_over_ranks": over_ranks(ar3), "disc_hits_last_step": int(disc_dev.sum().item())}
    # the other BASELINE configurations as timed legs of the N = 1 line (extra keys only; the headline loop above ran first)
    legs = None
    if world == 1 and a.rays <= 0 and not a.no_extras:
        try:
            legs = extra_legs(isx, np, a.seed, cus, load_pmc()[0])
        except Exception as e:   # an extra leg must never cost the headline line: report it and go on
            import traceback
            legs = {"error": f"{type(e).__name__}: {e}", "traceback": traceback.format_exc()[-1500:]}
    if rank == 0:
        rays_total = n * world * a.steps
        value = rays_total / dt / 1e6
        k_ms = float(np.mean(kernel_ms)) if kernel_ms else float("nan")
        # --- rooflines, per launch.  The headline map runs as TWO kernels (trace kernel -> exit lines in HBM -> binning
        # kernel); libisx times them separately with HIP events on its stream (isx_last_kernel_ms).
        t_single = float(np.mean([k[0] for k in kind_ms])) if kind_ms else 0.0
        t_trace = float(np.mean([k[1] for k in kind_ms])) if kind_ms else 0.0
        t_bin = float(np.mean([k[2] for k in kind_ms])) if kind_ms else 0.0
        pipeline = t_trace > 0.0 and t_bin > 0.0
        st = census[-1][1]                                                 # (the last step's own census)
        lines = sum(c.counted_below_z for _, c in census) / float(sum(k for k, _ in census))   # exit lines per launch (48 B each, written once, read once)
        alg_bytes_trace, alg_bytes_bin = 48.0 * lines, 48.0 * lines + nb * 8.0
        alg_bytes = (alg_bytes_trace + alg_bytes_bin) if pipeline else nb * 8.0
        hbm_gbs = alg_bytes / (k_ms * 1e-3) / 1e9
        f_ray = N_BOUNCE * F_BOUNCE + P_EXIT * nb * F_DISC
        fp64_tflops = n * f_ray / (k_ms * 1e-3) / 1e12
        # executed-instruction counters come from the committed rocprofv3 PMC passes (profiles/pmc_summary.json,
        # written by tools/summarize_profile.py).  They are used only if they were measured on THIS kernel code.
        pj, pmc_note = load_pmc()
        traffic, fp64 = pj.get("hbm_bytes_per_launch"), pj.get("fp64_executed")
        peak_issue = cus * 4 * VALU_CLOCK_GHZ / 4.0
        del peak_issue

        kern = pj.get("kernels", {})
        k_trace = next((k for k in kern if "trace" in k), "isx_trace_assist_kernel")
        k_bin = next((k for k in kern if "bin" in k), "isx_bin_cols_kernel")
        if pipeline:
            b_trace = issue_block(k_trace, t_trace, kern.get(k_trace), n, cus)
            b_bin = issue_block(k_bin, t_bin, kern.get(k_bin), n, cus)
            b_trace["algorithmic_hbm_bytes"], b_bin["algorithmic_hbm_bytes"] = alg_bytes_trace, alg_bytes_bin
            for b in (b_trace, b_bin):   # HBM bytes the counters saw per launch / the bytes the algorithm needs (1 = no wasted traffic)
                b["traffic_ratio"] = (b["traffic"] / b["algorithmic_hbm_bytes"]) if b.get("traffic") else None
            dominant, other = (b_bin, b_trace) if t_bin >= t_trace else (b_trace, b_bin)
        else:
            dominant, other = issue_block("isx_trace_bin_kernel", t_single or k_ms, pj if not kern else None, n, cus), None
        dominant["note"] = (f"binding resource: VALU issue (no MFMA, HBM idle).  achieved = executed SQ_INSTS_VALU per ray from "
                            f"{pmc_note} x rays / the kernel's live time (HIP events on the library's stream); peak = the rate at which "
                            f"{cus} CUs x 4 SIMDs x {VALU_CLOCK_GHZ} GHz can issue THIS kernel's executed instruction mix (cycles per class in "
                            f"tools/summarize_profile.py; peak_4cycle: 4 cycles for every instruction); traffic = HBM bytes per launch "
                            f"(FETCH_SIZE + WRITE_SIZE passes)")
        out = {
            "metric": "Mrays/sec whole-node, 180x90 fluxmap src(-60,0,-75); achieved HBM GB/s vs peak",
            "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{n} rays/GPU/step, pencil source src(-60,0,-75) dir(5,0,0) -> Lambertian wall "
                                   f"(rho .99, port 170deg), 180x90 detector map, 40 cm disc at 100 cm "
                                   f"({which})",
                       "rays_per_gpu_per_step": n, "grid": [cfg.n_theta, cfg.n_phi], "seed": hex(a.seed),
                       "parallelism": f"rays sharded over {world} GPU(s), one RCCL all-reduce of the histogram",
                       "reduce_path": mode, "trace_mode": a.trace_mode, "device": devname,
                       "hip_runtime_images": hip_images,
                       "rccl_world_size": (dist.get_world_size() if use_dist else 1),
                       "torch_backend": (dist.get_backend() if use_dist else None)},
            # The binding resource is VALU issue (no MFMA, HBM nearly idle).  One wave64 instruction occupies a
            # SIMD's VALU for >= 4 cycles, so a GPU issues at most n_simd * clock / 4 wave-instructions per second by this
            # convention.  `roofline` is the dominant (longer) kernel of the launch, `roofline_other_kernel` the second one.
            "roofline": dominant,
            "roofline_other_kernel": other,
            # the whole step against the issue roofline: (sum over the step's kernels of their mix-priced issue cycles at 2.4 GHz) /
            # ms_per_step -- what the step would take if both kernels issued without a gap, over what it takes with launches,
            # zeroing, synchronisation and the all-reduce
            "roofline_step": (None if not (pipeline and dominant.get("peak_mix_cycles_per_ray") and other and other.get("peak_mix_cycles_per_ray")) else {
                "bound": "valu_issue", "unit": "ms per step",
                "peak": (dominant["peak_mix_cycles_per_ray"] + other["peak_mix_cycles_per_ray"]) * n / (cus * 4 * VALU_CLOCK_GHZ * 1e9) * 1e3,
                "achieved": dt / a.steps * 1e3,
                "frac": (dominant["peak_mix_cycles_per_ray"] + other["peak_mix_cycles_per_ray"]) * n / (cus * 4 * VALU_CLOCK_GHZ * 1e9) / (dt / a.steps)}),
            "pipeline": ({"kernels": [k_trace, k_bin], "trace_ms": t_trace, "bin_ms": t_bin,
                          "exit_lines_per_launch": lines} if pipeline else None),
            # per-rank means of the timed steps, [min, max] over the ranks: kernel time, its two halves, the one all-reduce
            "per_rank_ms_min_max": per_rank,
            "configs4": configs4,
            "configs3": (legs or {}).get("configs3", configs3),
            "configs0": (legs or {}).get("configs0"),
            "configs2": (legs or {}).get("configs2"),
            "perpos_8p1e8": (legs or {}).get("perpos_8p1e8"),
            "size_sweep": (legs or {}).get("size_sweep"),
            "surfaces": (legs or {}).get("surfaces"),
            "chord_mode": (legs or {}).get("chord_mode"),
            "extra_legs_error": (legs or {}).get("error"),
            # the HBM figure the north star asks for: algorithmic bytes (exit lines written once and read once, 48 B each, plus
            # one 129.6 KB histogram) / kernel time against 8 TB/s
            "roofline_hbm": {"bound": "hbm", "achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": hbm_gbs / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes,
                             "traffic_ratio": (traffic / alg_bytes) if traffic else None,
                             "note": "HBM is not the binding resource of this path (SURVEY.md 8d): ~2 GB of exit lines per 5e7 "
                                     "rays cross HBM once in each direction; no traffic is faked to raise the fraction"},
            # reference-algorithm flop (brute-force convention of SURVEY.md 8d) -- NOT a utilisation: the kernel culls
            "roofline_fp64_model": {"bound": "valu_fp64", "achieved": fp64_tflops, "peak": FP64_VALU_PEAK_TFLOPS,
                                    "unit": "TFLOP/s", "frac": fp64_tflops / FP64_VALU_PEAK_TFLOPS,
                                    "flop_per_ray_model": f_ray,
                                    "note": "algorithmic FP64 flop of the reference algorithm (16200 disc tests per exiting "
                                            "ray, BASELINE.md section 2) / kernel time; exceeds 1 because the kernel culls"},
            # EXECUTED f64 operations (rocprofv3 SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64; fma = 2 flop), lane-level:
            # wave-instructions x 64 x measured lane utilisation
            "roofline_fp64_executed": (None if not fp64 else {
                "bound": "valu_fp64", "unit": "TFLOP/s", "peak": FP64_VALU_PEAK_TFLOPS,
                "achieved": fp64["lane_flop_per_ray"] * n / (k_ms * 1e-3) / 1e12,
                "frac": fp64["lane_flop_per_ray"] * n / (k_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                "f64_wave_insts_per_ray": fp64["wave_insts_per_ray"], "share_of_valu_insts": fp64["share_of_valu"],
                "note": "executed FP64 from the PMC pass; the rest of the VALU stream is Philox (32-bit integer), f32 cull "
                        "arithmetic, conversions, compares and moves"}),
            "census_last_step": {"launched": st.launched, "counted_below_z": st.counted_below_z,
                                 "wall_hits": st.wall_hits, "bin_increments": st.bin_increments},
            "hist_sum_last_step": total_hits,
        }
        if world == 1 and a.cpu_rays != 0:
            out["cpu_baseline"] = cpu_baseline(a.seed, a.cpu_rays)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    isx.shutdown()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
