"""RESEARCH: the exclusion table of profiles/r02_parity_scan.md — every inferred ROBAST behaviour that was varied, against
the reference's 8.1e8-ray maps at port 170 and 163 deg (binomial chi2 over all bins, total-hit ratio, 15-degree band ratios)."""
import sys, json, time
import numpy as np
import hyp, scan
N = int(float(sys.argv[1])); out = sys.argv[2]
H = [
 ("base", "model of round 1: cosine law about the geometric normal, rim = same surface, roughness ignored"),
 ("rim=1", "rim absorbs"), ("rim=2", "no rim (shell has no thickness at the port)"), ("rim=3", "rim reflects specularly"),
 ("rho_rim=0.5", "rim reflectance 0.5"),
 ("law=1", "emission uniform over the hemisphere (theta = acos(u))"), ("law=2", "sin(theta) = u (theta = asin(u))"),
 ("law=3", "n + uniform point of the unit ball"), ("law=4,law_pow=1.03", "cos^1.03 lobe"), ("law=4,law_pow=0.97", "cos^0.97 lobe"),
 ("rough_lambert=1", "normal tilted by N(0, 0.01) before the cosine emission"),
 ("rough_lambert=1,sigma=0.1", "same, sigma 0.1"), ("rough_lambert=1,sigma=0.573", "same, sigma 0.573 (0.01 rad read as degrees x RadToDeg)"),
 ("law=8,law_pow=0.05", "5 % of the reflections specular"), ("law=5", "specular reflection about a cosine-distributed random facet normal"),
 ("step_back=1e-6", "hit point pulled back by ROBAST's 1e-6 cm epsilon"), ("count_absorbed=1", "absorbed rays below the port plane counted too"),
 ("port_test=1", "every ray leaving downwards counted (no z < -100 test on the box)"), ("outer=1", "outer sphere absorbs"),
 ("box_half=200.", "world box 200 cm"), ("det_distance=101.", "detector distance 101 cm"), ("first_specular=1", "first interaction specular"),
]
import os
res = json.load(open(out))["rows"] if os.path.exists(out) else []
done = {r["spec"] for r in res}
for spec, text in H:
    if spec in done:
        continue
    t = time.time()
    o = scan.run_one(spec, N, ("pp_03_31_0", "pp_04_1_0"))
    row = {"spec": spec, "text": text}
    for m, c in o.items():
        b = np.array(c["band_ratio"]).reshape(6, 2)
        row[m] = {"ratio": c["ratio"], "chi2_dof": c["chi2_dof"], "exit": c["exit_frac"],
                  "bands": [float(x) for x in (np.array(c["band_ratio"]) - 1) * 100]}
    res.append(row)
    print(spec, {m: (round(row[m]["ratio"], 4), round(row[m]["chi2_dof"], 3)) for m in o}, round(time.time() - t), "s", flush=True)
    json.dump({"N": N, "rows": res}, open(out, "w"), indent=1)
