"""Where the binning work goes (GPU box; needs variants/libisx_diag.so = libisx built with -DISX_DIAG):
   ISX_LIB_PATH=variants/libisx_diag.so python tools/diag_binning.py"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import altair_raytracing_amd as isx
L = isx.load(); isx.init(0)
if os.environ.get('ISX_PIPELINE'): isx.set_option('pipeline', int(os.environ['ISX_PIPELINE']))
L.isx_diag_read.argtypes = [C.POINTER(C.c_uint64)]
def diag():
    a = (C.c_uint64 * 48)()
    assert L.isx_diag_read(a) == 0
    return np.array(a[:32], dtype=np.uint64)
def run(name, c, n):
    diag()
    h, st = isx.fluxmap(c, n, 5)
    d = diag().astype(float)
    lines = st.counted_below_z
    out = {"lines": lines, "hits_per_line": st.bin_increments / lines,
           "path_share": {"fast": d[0] / lines, "caps": d[1] / lines, "fallback": d[2] / lines, "miss": d[3] / lines},
           "col_iterations_per_line": {"fast": d[4] / max(d[0], 1), "caps": d[5] / max(d[1], 1), "fallback": d[6] / max(d[2], 1)},
           "candidates_per_line": {"fast": d[7] / max(d[0], 1), "caps": d[8] / max(d[1], 1), "fallback": d[9] / max(d[2], 1)},
           "row_passes_per_line": d[11] / lines, "split_passes_per_line": d[10] / lines,
           "tier2_per_candidate": d[12] / max(d[7] + d[8] + d[9], 1), "tier3_per_candidate": d[13] / max(d[7] + d[8] + d[9], 1),
           "WRONG_DECISIONS": int(d[14]), "candidates_total": int(d[7] + d[8] + d[9]),
           "lane_fill": {"fast": d[7] / max(d[4] * 64, 1), "caps": d[8] / max(d[5] * 64, 1), "fallback": d[9] / max(d[6] * 64, 1)}}
    if d[16:22].sum() > 0:   # binning kernel with slot queues: share of the waves' cycles per region (ISX_BD_MARK)
        # isx_bin_slots_kernel (BRDF source) / isx_bin_cols_kernel (pencil source): what the six marks bracket
        names = (["batch_prep", "producers", "push", "pop_fetch_t0", "coefficients_and_walk", "unit_bookkeeping"] if name == "brdf" else
                 ["batch_prep", "producers", "push", "pop", "slot_prologue", "walk"])
        if name != "brdf":   # finer marks inside the column producer
            names += ["owner_search", "cap_through_bpermute", "cap_rows", "producer_loop_and_drain_check"]
        tot = d[16:16 + len(names)].sum()
        out["slot_kernel_wave_cycle_share"] = {n: round(float(v / tot), 4) for n, v in zip(names, d[16:16 + len(names)])}
        out["slot_kernel_wave_cycles_per_pass"] = float(d[16:22].sum() / max(d[11], 1))
    print(name, json.dumps(out, indent=1))
c = isx.default_config()
run("headline", c, 20_000_000)
c.source_model = 1; c.roughness_rad = 0.5; c.reflectance = 1.0; c.max_points = 10000; c.box_half = 200.0
run("brdf", c, 10_000_000)
c = isx.default_config(); c.theta_max_deg = 160.0; c.dir[1] = 2.0
run("port160", c, 10_000_000)
c = isx.default_config(); c.n_theta, c.n_phi, c.det_diameter = 200, 180, 4.0
run("grid200x180_4cm", c, 5_000_000)
c = isx.default_config(); c.n_theta, c.n_phi, c.det_diameter, c.det_distance = 45, 20, 10.0, 100.0
run("grid45x20_10cm", c, 10_000_000)
