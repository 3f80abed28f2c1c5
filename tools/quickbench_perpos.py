#!/usr/bin/env python3
"""Occupancy experiment on the trace-dominated per-position kernel: variants "path:blocks_per_cu"."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("ISX_QB_CHILD"):
    sys.path.insert(0, ROOT)
    import altair_raytracing_amd as isx
    isx.load(); isx.init(0)
    isx.set_option("blocks_per_cu", int(os.environ["ISX_QB_BPC"]))
    c = isx.default_config()
    isx.fluxmap_per_position(c, 200, 1)
    ts = []
    for rep in range(3):
        h, st = isx.fluxmap_per_position(c, 2000, 7 + rep)   # 3.24e7 rays
        ts.append(st.t_kernel_ms)
    print(json.dumps({"ms": min(ts), "Mrays": 16200 * 2000 / min(ts) / 1e3, "sum": int(h.sum())}))
    sys.exit(0)
for v in os.environ["ISX_VARIANTS"].split(","):
    path, bpc = v.split(":")
    env = dict(os.environ, ISX_QB_CHILD="1", ISX_QB_BPC=bpc, ISX_LIB_PATH=os.path.join(ROOT, path))
    r = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True, timeout=600)
    print(v, r.stdout.strip() or r.stderr[-300:], flush=True)
