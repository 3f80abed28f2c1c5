"""Compare total ray hits of the reference's seven 8.1e8-ray per-position maps with this build (GPU)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import altair_raytracing_amd as isx
isx.load(); isx.init(0)
fx = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "reference_fixtures.json")))
extra = {}
for a in sys.argv[1:]:
    k, v = a.split("="); extra[k] = float(v)
for m in fx["per_position_maps"]:
    c = isx.default_config()
    c.theta_max_deg = m["port_deg"]
    for k in range(3):
        c.dir[k] = m["source_direction"][k]
    for k, v in extra.items():
        setattr(c, k, type(getattr(c, k))(v))
    tot, exits = 0, 0
    for rep in range(2):
        h, st = isx.fluxmap_per_position(c, 50000, 1234 + rep, 1)
        tot += int(h.sum()); exits += st.counted_below_z
    prof = (h / 50000.0).mean(axis=1)
    gold = np.array(m["theta_profile"])
    k = [0, 30, 60, 90, 120, 150]
    print(f"port {m['port_deg']:5.1f} dir {m['source_direction']} ref hits {m['total_hits']} ours {tot/2:.0f} ratio {tot/2/m['total_hits']:.5f} "
          f"exit frac {exits/2/8.1e8:.5f} profile ratio {np.round(prof[k]/gold[k],3)}")
