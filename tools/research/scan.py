"""Hypothesis scan (RESEARCH): each named variant of the inferred ROBAST behaviour against reference maps.
usage: python scan.py N name=val,name=val ...   (each argument one hypothesis; 'base' = defaults)"""
import sys, json, time
import numpy as np
import hyp

def run_one(spec, N, maps=("pp_03_31_0",), seed=1):
    kw = {}
    if spec != "base":
        for kv in spec.split(","):
            k, v = kv.split("=")
            kw[k] = float(v) if ("." in v or "e" in v) else int(v)
    R = hyp.ref_maps()
    out = {}
    for name in maps:
        info, ref = R[name]
        c = hyp.default_cfg(theta_max_deg=info["port_deg"], dir=info["source_direction"], **kw)
        h, st, dz, rad = hyp.run(c, N, seed)
        cmp = hyp.compare(ref, info["rays_per_position"], h, N, bands=12)
        cmp["exit_frac"] = st.counted / N
        cmp["rim_per_ray"] = st.rim_hits / N
        out[name] = cmp
    return out

if __name__ == "__main__":
    N = int(float(sys.argv[1]))
    maps = ("pp_03_31_0",)
    specs = sys.argv[2:]
    if specs and specs[0].startswith("maps="):
        maps = tuple(specs[0][5:].split("+")); specs = specs[1:]
    np.set_printoptions(linewidth=220, precision=2, suppress=True)
    for s in specs:
        t = time.time()
        o = run_one(s, N, maps)
        for m, c in o.items():
            print(f"{s:40s} {m:11s} ratio {c['ratio']:.4f} chi2 {c['chi2_dof']:.3f} exit {c['exit_frac']:.4f} rim/ray {c['rim_per_ray']:.4f} bands% {(np.array(c['band_ratio'])-1)*100}", flush=True)
