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
    a = (C.c_uint64 * 48)()
    assert L.isx_diag_read(a) == 0
    return np.array(a[16:], dtype=np.float64)   # [0..15] the tracer waves' (persistent_body's) slots, [16..31] the assist wave's
REGIONS = ["refill", "step0_search", "generic_flush", "step0_interact", "steps_1..N-1", "census_rescatter", "sink"]
def run_assist(name, fn, n_rays, steps=8):
    """the trace kernels with an assist wave (assist_body): where the lanes of the launch go.
    lane-cycles of the kernel = (tracer + assist) wave cycles x 64; useful = a lane with a live ray inside a bounce step."""
    if name != "discs":
        isx.set_option("bin_mode", 2)
        diag()
    st = fn(); d = diag()
    isx.set_option("bin_mode", 1)
    cyc = d[:4]; trips = d[7]; a = d[16:]
    live_steps = d[4]                      # lanes that attempted a bounce, summed over the steps of all trips
    slots = trips * 64 * steps
    tracer_cyc, assist_cyc = cyc.sum(), a[0] + a[1]
    out = {"rays": n_rays, "kernel_ms": st.t_kernel_ms, "wall_hits_per_ray": st.wall_hits / n_rays, "steps_per_trip": steps,
           "tracer_wave_cycle_share": {r: round(c / cyc.sum(), 4) for r, c in zip(["refill", "bounce_steps", "census", "hand_over"], cyc)},
           "trips_per_ray": trips / n_rays,
           "lanes_running_at_trip_start": d[8] / trips, "lanes_waiting_to_be_handed_over_at_trip_start": d[9] / trips,
           "lane_fill_inside_the_bounce_steps": live_steps / slots,
           "  lost to rays that ended or left earlier in the trip": (d[8] * steps - live_steps) / slots,
           "  lost to lanes without a ray at the top of the trip": (trips * 64 - d[8]) * steps / slots,
           "drain_trip_share (queue dry)": d[14] / trips, "lanes_running_in_drain_trips": d[15] / max(d[14], 1),
           "lanes_running_outside_drain": (d[8] - d[15]) / max(trips - d[14], 1),
           "hand_overs_refused_per_trip": d[10] / trips, "rays_taken_back_per_trip": d[11] / trips, "rays_ended_in_tracers_per_trip": d[12] / trips,
           "end_of_launch_wait_polls": d[13], "waves_that_gave_their_last_rays_away": d[6], "rays_given_away_per_such_wave": d[5] / max(d[6], 1),
           "assist_wave": {"share_of_all_wave_cycles": assist_cyc / (assist_cyc + tracer_cyc), "at_work_share_of_its_cycles": a[1] / max(assist_cyc, 1),
                           "batches_per_ray": a[2] / n_rays, "rays_per_batch (of 64 lanes)": a[3] / max(a[2], 1),
                           "hand_overs_per_ray": a[3] / n_rays, "sent_to_the_back_of_the_queue_per_ray": a[4] / n_rays,
                           "returned_to_tracers_per_ray": a[5] / n_rays, "ended_here_per_ray": a[6] / n_rays,
                           "cycles_per_batch": a[1] / max(a[2], 1)}}
    lane_cyc = (tracer_cyc + assist_cyc) * 64
    fill = a[3] / max(a[2], 1)
    out["lane_cycle_budget"] = {
        "tracer bounce steps, live lanes": cyc[1] * 64 * (live_steps / slots) / lane_cyc,
        "tracer bounce steps, dead lanes": cyc[1] * 64 * (1 - live_steps / slots) / lane_cyc,
        "tracer refill + census + hand-over (all lanes)": (cyc[0] + cyc[2] + cyc[3]) * 64 / lane_cyc,
        "assist wave at work, live lanes": a[1] * fill / lane_cyc,
        "assist wave at work, idle lanes": a[1] * (64 - fill) / lane_cyc,
        "assist wave waiting": a[0] * 64 / lane_cyc}
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
    if which == "lobe":      # "nonLambertianFlux copy.C":31-70: one rejection try per step (a "bounce step" of the tables = one try)
        c.surface_model = 1
    if which == "rough":
        c.lambertian = 0; c.roughness_rad = 0.5
    if which == "brdf":
        c.source_model = isx.SOURCE_BRDF; c.roughness_rad = 0.5; c.reflectance = 1.0; c.max_points = 10000; c.box_half = 200.0
    if which == "discs":   # BASELINE configs[3]: 362 disc positions share 1e7 rays (shell 100.1-105, reflectance 1)
        import math
        discs = []
        for th in np.arange(-45.0, 45.0 + 1e-9, 0.5):
            for ph in (0.0, 180.0):
                t_, p_ = math.radians(th), math.radians(ph)
                x, y, z = 200 * math.sin(t_) * math.cos(p_), 200 * math.sin(t_) * math.sin(p_), -200 * math.cos(t_)
                rot = -math.atan2(math.sqrt(x * x + y * y), -100 - z)
                discs.append([x, y, z, math.sin(rot), 0.0, math.cos(rot)])
        c.r_out = 105.0; c.reflectance = 1.0; c.roughness_rad = 0.0; c.max_points = 10000; c.box_half = 200.0; c.src[2] = -80.0
        n = 10_000_000
        diag()
        st = isx.disc_sweep(c, np.array(discs), 5.0, 0.1, n, 7)[1]
        class _S: pass
        run_assist(which, lambda: st, n, int(os.environ.get("ISX_STEPS", "8")))
        continue
    if which == "perpos":
        run(which, lambda: isx.fluxmap_per_position(c, 2000, 5)[1], 2000 * 16200)
    else:
        n = 20_000_000
        if os.environ.get("ISX_DIAG_OLD"):
            isx.set_option("assist", 0)
            run(which, lambda: isx.fluxmap(c, n, 5)[1], n)
        isx.set_option("assist", 1)
        run_assist(which, lambda: isx.fluxmap(c, n, 5)[1], n, int(os.environ.get("ISX_STEPS", "8")))
