import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import altair_raytracing_amd as isx
isx.load(); isx.init(0)
c = isx.default_config(); c.trace_mode = int(os.environ.get("TM", "0"))
isx.set_option("bin_mode", int(os.environ.get("BM", "1")))
for rep in range(3):
    h, st = isx.fluxmap(c, 20_000_000, 5 + rep)
print(st.t_kernel_ms)
