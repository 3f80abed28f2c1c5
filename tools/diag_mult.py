"""How often do lanes of one ds_add_u32 of the column walk name the same bin?  (GPU box; needs variants/libisx_mult.so = libisx built
with -DISX_DIAG -DISX_DIAG_MULT -DISX_DIAG_TIMING_ONLY:  ISX_LIB_PATH=variants/libisx_mult.so python tools/diag_mult.py)
north_star asks for "wavefront ballot/shuffle to coalesce same-bin writes" (the reference's hitCount++, fluxAtObserverFast.C:1291-1293);
this measures what such a pass could merge before anybody builds it."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import altair_raytracing_amd as isx
L = isx.load(); isx.init(0)
L.isx_diag_read.argtypes = [C.POINTER(C.c_uint64)]
def diag():
    a = (C.c_uint64 * 48)()
    assert L.isx_diag_read(a) == 0
    return np.array(a[:48], dtype=np.float64)
out = {}
# (self-check of the instrument: on a 1 x 1 grid every lane of an instruction names THE bin; on 2 x 2 most do)
for name, n, port, grid in (("self-check, 1 x 1 grid", 200_000, 170.0, (1, 1)), ("self-check, 2 x 2 grid", 200_000, 170.0, (2, 2)),
                            ("headline (port 170)", 5_000_000, 170.0, (180, 90)), ("port 160", 5_000_000, 160.0, (180, 90))):
    c = isx.default_config(); c.theta_max_deg = port; c.n_theta, c.n_phi = grid
    diag(); h, st = isx.fluxmap(c, n, 5); d = diag()
    out[name] = {"rays": n, "bin_increments": int(st.bin_increments), "lane_adds_of_1_in_the_walk": int(d[8]),
                 "of_them_not_the_first_lane_on_their_bin": int(d[9]),
                 "mean_multiplicity (adds per distinct bin of a wave-instruction)": d[8] / max(d[8] - d[9], 1)}
print(json.dumps(out, indent=1))
