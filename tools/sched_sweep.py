"""Sweep of the generic-boundary-search batching (options sched_mask, sched_min) per configuration (GPU box):
   python tools/sched_sweep.py [flux|chord|brdf|perpos] ...      -> trace / binning kernel ms per setting"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import altair_raytracing_amd as isx
isx.load(); isx.init(0)
def cfg(which):
    c = isx.default_config()
    if which == "chord":
        c.trace_mode = 1
    elif which == "brdf":
        c.source_model = isx.SOURCE_BRDF; c.brdf[0], c.brdf[1], c.brdf[2] = 0.3, 0.4, 0.6
        c.roughness_rad = 0.5; c.reflectance = 1.0; c.max_points = 10000; c.box_half = 200.0
    return c
for which in (sys.argv[1:] or ["flux", "chord", "brdf"]):
    c = cfg(which)
    n = 20_000_000 if which == "brdf" else 50_000_000
    ref = None
    print(f"== {which}, {n} rays: total ms (trace ms) by sched_mask x sched_min", flush=True)
    mins = (1, 2, 4, 8, 12, 16, 24, 65)
    print("mask\\min " + " ".join(f"{m:>14d}" for m in mins))
    for mask in (0, 1, 3, 7, 15):
        row = []
        for mn in mins:
            isx.set_option("sched_mask", mask); isx.set_option("sched_min", mn)
            best = (1e9, 0)
            for _ in range(2):
                if which == "perpos":
                    h, st = isx.fluxmap_per_position(c, 10000, 5)
                else:
                    h, st = isx.fluxmap(c, n, 5)
                k = isx.last_kernel_ms()
                best = min(best, (st.t_kernel_ms, k[1]))
            if ref is None: ref = h.copy()
            assert (h == ref).all(), "the schedule changed a result"
            row.append(f"{best[0]:7.2f}({best[1]:5.2f})")
        print(f"{mask:8d} " + " ".join(row), flush=True)
isx.set_option("sched_mask", 3); isx.set_option("sched_min", 12)
