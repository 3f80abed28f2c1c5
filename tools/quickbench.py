#!/usr/bin/env python3
"""Tuning helper (GPU box): time isx.fluxmap for library variants / options.
usage: quickbench.py [n_rays] ; variants listed in VARIANTS or env ISX_VARIANTS="path:bpc,path:bpc"."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000

if os.environ.get("ISX_QB_CHILD"):
    sys.path.insert(0, ROOT)
    import altair_raytracing_amd as isx
    isx.load(); isx.init(0)
    bpc = int(os.environ["ISX_QB_BPC"])
    isx.set_option("blocks_per_cu", bpc)
    cfg = isx.default_config()
    cfg.trace_mode = int(os.environ.get("ISX_QB_TRACE_MODE", "0"))
    out = {}
    for k in ("sched_mask", "sched_min"):
        if os.environ.get("ISX_QB_" + k.upper()):
            isx.set_option(k, int(os.environ["ISX_QB_" + k.upper()]))
    for mode in (1, 2):
        isx.set_option("bin_mode", mode)
        isx.fluxmap(cfg, 1_000_000, 1)
        ts = []
        for rep in range(3):
            h, st = isx.fluxmap(cfg, n, 0x5EED0001 + rep)
            ts.append(st.t_kernel_ms)
        out[f"mode{mode}_ms"] = min(ts)
        out[f"mode{mode}_Mrays"] = n / min(ts) / 1e3
        out[f"mode{mode}_ns_per_wallhit"] = min(ts) * 1e6 / max(st.wall_hits, 1)
        if mode == 1:
            out["hist_sum"] = int(h.sum())
            out["wall_hits_per_ray"] = st.wall_hits / n
            out["p_exit"] = st.counted_below_z / n
    print(json.dumps(out))
    sys.exit(0)

variants = os.environ.get("ISX_VARIANTS")
if variants:
    variants = [v.split(":") for v in variants.split(",")]
else:
    variants = [("altair-raytracing_amd/csrc/libisx.so", "1")]
for v in variants:
    path, bpc = v[0], v[1]
    env = dict(os.environ, ISX_QB_CHILD="1", ISX_QB_BPC=bpc, ISX_LIB_PATH=os.path.join(ROOT, path))
    for extra in v[2:]:
        k, val = extra.split("=")
        env["ISX_QB_" + k.upper()] = val
    r = subprocess.run([sys.executable, __file__, str(n)], env=env, capture_output=True, text=True, timeout=600)
    print(":".join(v), r.stdout.strip() or r.stderr[-400:], flush=True)
