"""Where the trace loop's time and lanes go (GPU box; needs variants/libisx_diag.so = libisx built with -DISX_DIAG):
   ISX_LIB_PATH=variants/libisx_diag.so python tools/diag_trace.py [flux|brdf|perpos] ...
Wave cycles (s_memtime) per region of a loop trip of persistent_body, and the lane census at the top of a trip."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import altair_raytracing_amd as isx
L = isx.load(); isx.init(0)
L.isx_diag_read.argtypes = [C.POINTER(C.c_uint64)]
def diag():
    a = (C.c_uint64 * 32)()
    assert L.isx_diag_read(a) == 0
    return np.array(a[16:], dtype=np.float64)
REGIONS = ["refill", "step0_search", "generic_flush", "step0_interact", "steps_1..N-1", "census_rescatter", "sink"]
def run_assist(name, fn, n_rays):
    """the trace kernels with an assist wave (assist_body): tracer waves only"""
    isx.set_option("bin_mode", 2)
    diag(); st = fn(); d = diag()
    isx.set_option("bin_mode", 1)
    cyc = d[:4]; trips = d[7]
    out = {"rays": n_rays, "kernel_ms": st.t_kernel_ms, "wall_hits_per_ray": st.wall_hits / n_rays,
           "region_share_of_tracer_wave_cycles": {r: round(c / cyc.sum(), 4) for r, c in zip(["refill", "bounce_steps", "census", "hand_over"], cyc)},
           "trips_per_ray": trips / n_rays, "bounces_per_trip_per_wave": st.wall_hits / trips,
           "lanes_running_at_trip_start": d[8] / trips, "lanes_waiting_to_be_handed_over_at_trip_start": d[9] / trips,
           "hand_overs_refused_per_trip": d[10] / trips, "rays_taken_back_per_trip": d[11] / trips, "rays_ended_in_tracers_per_trip": d[12] / trips,
           "end_of_launch_waits_per_wave": d[13] / max(1, 1)}
    print(name + " (assist)", json.dumps(out, indent=1), flush=True)
def run(name, fn, n_rays):
    isx.set_option("bin_mode", 2)      # trace only (the binning kernel is not launched)
    diag(); st = fn(); d = diag()
    isx.set_option("bin_mode", 1)
    cyc = d[:7]; trips = d[7]
    out = {"rays": n_rays, "kernel_ms": st.t_kernel_ms, "wall_hits_per_ray": st.wall_hits / n_rays,
           "region_share_of_wave_cycles": {r: round(c / cyc.sum(), 4) for r, c in zip(REGIONS, cyc)},
           "trips_per_ray": trips / n_rays, "bounces_per_trip_per_wave": st.wall_hits / trips,
           "lanes_running_at_trip_start": d[8] / trips, "lanes_parked_at_trip_start": d[9] / trips,
           "drain_trip_share": d[10] / trips, "lanes_running_in_drain_trips": d[11] / max(d[10], 1),
           "lanes_running_outside_drain": (d[8] - d[11]) / max(trips - d[10], 1),
           "rays_ended_per_trip": d[12] / trips, "flushes_per_trip": d[13] / trips, "lanes_per_flush": d[14] / max(d[13], 1),
           "lanes_refilled_per_trip": d[15] / trips}
    print(name, json.dumps(out, indent=1), flush=True)
for which in (sys.argv[1:] or ["flux"]):
    c = isx.default_config()
    if which == "brdf":
        c.source_model = isx.SOURCE_BRDF; c.roughness_rad = 0.5; c.reflectance = 1.0; c.max_points = 10000; c.box_half = 200.0
    if which == "perpos":
        run(which, lambda: isx.fluxmap_per_position(c, 2000, 5)[1], 2000 * 16200)
    else:
        n = 20_000_000
        isx.set_option("assist", 0)
        run(which, lambda: isx.fluxmap(c, n, 5)[1], n)
        isx.set_option("assist", 1)
        run_assist(which, lambda: isx.fluxmap(c, n, 5)[1], n)
