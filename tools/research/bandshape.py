"""Which hypothesis has the SHAPE of the residual?  For each variant of tools/research/hyp.c: the ratio variant / base of the
detector-angle profile in 5-degree bands (same seed: common random numbers), for the 170 and 163 degree ports, against the
measured reference / build factor of profiles/r02_overdispersion.json; least-squares amplitude a of (ratio - 1) and what is
left.  CPU, ~1 min per run.   python tools/research/bandshape.py [rays]  > profiles/r02_bandshape.json"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import hyp

ROOT = hyp.ROOT
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
tgt = {m["map"]: m for m in json.load(open(os.path.join(ROOT, "profiles", "r02_overdispersion.json")))["maps"]}
ports = {170.0: "pp_03_31_0", 163.0: "pp_04_1_0"}
variants = [("rim=1", dict(rim=1)), ("rim=2", dict(rim=2)), ("rim=3", dict(rim=3)), ("rho_rim=0.5", dict(rho_rim=0.5)),
            ("law=4,pow=1.03", dict(law=4, law_pow=1.03)), ("law=4,pow=0.97", dict(law=4, law_pow=0.97)),
            ("law=8,5% specular", dict(law=8, law_pow=0.05)), ("rough_lambert sigma=0.1", dict(rough_lambert=1, sigma=0.1)),
            ("first_specular", dict(first_specular=1)), ("det_distance=101", dict(det_distance=101.0)),
            ("replay theta 5%", dict(replay_q=0.05, replay_what=1)), ("replay theta+phi 5%", dict(replay_q=0.05, replay_what=3))]


def bands(h):
    return np.array([h[10 * b:10 * b + 10].sum() for b in range(18)], float)


base = {}
for port in ports:
    h, st, _, _ = hyp.run(hyp.default_cfg(theta_max_deg=port), N, 11)
    base[port] = bands(h)
    print("base", port, st.counted, file=sys.stderr, flush=True)
out = {"rays": N,
       "target_minus_1_percent": {str(p): [round((x - 1) * 100, 3) for x in tgt[ports[p]]["band_ratio_ref_over_build_5deg"]] for p in ports},
       "variants": []}
use = slice(0, 15)     # bands up to 75 degrees (beyond: few hits)
for name, kw in variants:
    rec = {"variant": name}
    num = den = 0.0
    prof = {}
    for port in ports:
        h, st, _, _ = hyp.run(hyp.default_cfg(theta_max_deg=port, **kw), N, 11)
        r = bands(h) / base[port] - 1.0
        t = np.array(tgt[ports[port]]["band_ratio_ref_over_build_5deg"]) - 1.0
        w = 1.0 / np.array(tgt[ports[port]]["band_sigma_5deg"]) ** 2
        num += float((w[use] * r[use] * t[use]).sum())
        den += float((w[use] * r[use] * r[use]).sum())
        prof[port] = (r, t, w)
        rec[f"ratio_minus_1_percent_{int(port)}"] = [round(x * 100, 3) for x in r]
    a = num / den if den > 0 else 0.0
    chi0 = sum(float((w[use] * t[use] ** 2).sum()) for (r, t, w) in prof.values())
    chi1 = sum(float((w[use] * (t[use] - a * r[use]) ** 2).sum()) for (r, t, w) in prof.values())
    rec.update({"best_amplitude": round(a, 3), "chi2_of_target_alone": round(chi0, 1),
                "chi2_after_subtracting_a_times_variant": round(chi1, 1), "bands_used": 30})
    out["variants"].append(rec)
    print(json.dumps(rec), file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
