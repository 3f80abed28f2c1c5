"""Per-position map (8.1e8 rays) and exit-dz histogram times by workgroups per CU of the trace-only kernels (GPU box)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import altair_raytracing_amd as isx
isx.load(); isx.init(0)
c = isx.default_config()
for bpc in (8, 12, 16, 24):
    isx.set_option("trace_blocks_per_cu", bpc)
    isx.fluxmap_per_position(c, 100, 5)
    t = min(isx.fluxmap_per_position(c, 50000, 5)[1].t_kernel_ms for _ in range(2))
    h, st = isx.exit_dz_hist(c, 50_000_000, 5) if hasattr(isx, "exit_dz_hist") else (None, None)
    print(f"bpc {bpc}: perpos 8.1e8 rays {t:.1f} ms = {8.1e8/t/1e3:.0f} Mrays/s", (f"dz 5e7: {st.t_kernel_ms:.2f} ms" if st else ""), flush=True)
