"""Trace-only and pipeline timings for builds with another workgroup size (variants/libisx_block<N>.so) and blocks per CU (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import altair_raytracing_amd as isx
isx.load(); isx.init(0)
c = isx.default_config()
n = 50_000_000
for bpc in (1, 2, 3, 4):
    isx.set_option("blocks_per_cu", bpc)
    out = []
    for pipe, bm in ((1, 2), (1, 1), (0, 1)):
        isx.set_option("pipeline", pipe); isx.set_option("bin_mode", bm)
        try:
            isx.fluxmap(c, 100000, 1)
            t = min(isx.fluxmap(c, n, 5)[1].t_kernel_ms for _ in range(3))
        except Exception as e:
            t = float("nan")
        out.append(t)
    t2 = min(isx.fluxmap_per_position(c, 50_000, 5)[1].t_kernel_ms for _ in range(2))
    print(os.environ.get("ISX_LIB_PATH", "default"), f"blocks_per_cu={bpc}: trace-only(rec) {out[0]:.2f} ms, pipeline {out[1]:.2f} ms, fused {out[2]:.2f} ms, perpos 8.1e8: {t2:.1f} ms", flush=True)
isx.set_option("bin_mode", 1); isx.set_option("pipeline", 0); isx.set_option("blocks_per_cu", 1)
