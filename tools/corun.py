"""Can the trace kernel of chunk k+1 share the CUs with the binning kernel of chunk k?  The flux-map pipeline's `overlap` option
(chunk k binned on a second stream while chunk k+1 is traced) with grids of ONE workgroup per CU for both kernels, so that a CU can
hold a trace workgroup (12 waves x 80 VGPRs) next to a binning workgroup -- which needs a library built with -DISX_BLOCK=512
(8 binning waves x 128 VGPRs; the default 1024-thread binning workgroup fills the register file by itself).
GPU box: ISX_LIB_PATH=variants/libisx_b512.so python tools/corun.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import altair_raytracing_amd as isx
isx.load(); isx.init(0)
c = isx.default_config()
n = 50_000_000
ref = None
for ov, ts, tb in ((0, 1, 0), (4, 1, 1), (8, 1, 1), (8, 2, 1), (16, 1, 1), (8, 1, 0), (0, 1, 1), (0, 1, 0)):
    isx.set_option("overlap", ov); isx.set_option("overlap_trace_streams", ts); isx.set_option("trace_blocks_per_cu", tb)
    isx.fluxmap(c, 1_000_000, 3)
    best = 1e9
    for _ in range(4):
        h, st = isx.fluxmap(c, n, 5)
        best = min(best, st.t_kernel_ms)
    if ref is None: ref = h
    print(f"overlap {ov:2d} trace streams {ts} trace workgroups per CU {tb or 'resident'}: {best:.3f} ms  {n / best / 1e3:.1f} Mrays/s  same {np.array_equal(h, ref)}", flush=True)
