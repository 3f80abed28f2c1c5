"""Is the reference's scatter binomial once the smooth theta-only residual is taken out?  (GPU box)

For each of the seven complete 8.1e8-ray maps of the reference: this build's bin probabilities from a 4e9-ray trace-once map
(every ray tested against every bin: p_b known to ~2e-4 relative, i.e. exactly on the scale of one 50 000-ray bin), then
  chi2_0  = sum (k_ref - n p)^2 / (n p (1-p)) / dof                    -- reference vs this build's probabilities
  chi2_row = the same after scaling p by one free factor per theta row (180 parameters: any theta-only systematic)
  chi2_pol = the same after scaling p by a 6th-order polynomial in theta (7 parameters: a smooth theta-only systematic)
chi2_row ~ 1 means: the whole residual is a function of theta alone and the reference's bins scatter binomially around it
(no overdispersion from correlated or duplicated rays at the level 1/sqrt(dof/2) = 1.1 %).
   python tools/research/overdispersion.py [rays_per_map]  > profiles/r02_overdispersion.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import altair_raytracing_amd as isx
isx.load(); isx.init(0)
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 4_000_000_000
z = np.load(os.path.join(ROOT, "tests", "golden", "reference_maps.npz"))
idx = [i for i in json.loads(str(z["index_json"])) if i["kind"] == "per_position" and i["complete"]]
out = []
for k, info in enumerate(idx):
    ref = z[info["name"] + "_hits"].astype(np.int64).reshape(180, 90)
    n = info["rays_per_position"]
    c = isx.default_config()
    c.theta_max_deg = info["port_deg"]
    for a in range(3):
        c.src[a] = info["source_position"][a]; c.dir[a] = info["source_direction"][a]
    h = np.zeros(16200, np.int64); done = 0; t_ms = 0.0
    while done < N:                       # 1e9 rays per call
        m = min(1_000_000_000, N - done)
        hh, st = isx.fluxmap(c, m, 777 + k, done)
        h += hh.astype(np.int64).reshape(-1); done += m; t_ms += st.t_kernel_ms
    p = (h / N).reshape(180, 90)
    use = p * n >= 5
    var = n * p * (1 - p) * (1 + n / N)   # (+ this build's own sampling variance: 1.25e-5 relative)
    def chi2(scale, npar):
        e = n * p * scale
        return float((((ref - e) ** 2 / np.where(var > 0, var * scale, 1))[use]).sum() / (use.sum() - npar))
    c0 = chi2(np.ones_like(p), 0)
    row = np.array([(ref[i][use[i]].sum() / (n * p[i][use[i]]).sum()) if use[i].any() else 1.0 for i in range(180)])
    c_row = chi2(row[:, None] * np.ones_like(p), int(use.any(axis=1).sum()))
    th = (np.arange(180) + 0.5) * 0.5
    w = np.array([(n * p[i][use[i]]).sum() for i in range(180)])
    ok = w > 0
    coef = np.polyfit(th[ok] / 90, row[ok], 6, w=np.sqrt(w[ok]))
    pol = np.polyval(coef, th / 90)
    c_pol = chi2(pol[:, None] * np.ones_like(p), 7)
    # azimuthal structure of what is left after the row scaling: chi2 of 12 phi-sectors x 9 theta-bands
    res = np.where(use, ref - n * p * row[:, None], 0.0); v = np.where(use, var * row[:, None], 0.0)
    sect = [(res[a * 20:(a + 1) * 20, b * 15 // 2:(b + 1) * 15 // 2].sum(), v[a * 20:(a + 1) * 20, b * 15 // 2:(b + 1) * 15 // 2].sum())
            for a in range(9) for b in range(12)]
    c_sect = float(np.mean([r * r / vv for r, vv in sect if vv > 0]))
    rec = {"map": info["name"], "port_deg": info["port_deg"], "direction": info["source_direction"], "rays_this_build": N,
           "kernel_s": round(t_ms * 1e-3, 2), "bins": int(use.sum()), "total_ratio_build_over_ref": float((n * p).sum() / ref.sum()),
           "chi2_per_dof": round(c0, 4), "chi2_after_row_scaling": round(c_row, 4), "chi2_after_6th_order_polynomial_in_theta": round(c_pol, 4),
           "chi2_of_108_sector_sums_after_row_scaling": round(c_sect, 3),
           "band_ratio_ref_over_build_5deg": [float(ref[10 * b:10 * b + 10].sum() / (n * p[10 * b:10 * b + 10]).sum()) for b in range(18)],
           "band_sigma_5deg": [float(1 / np.sqrt(max(ref[10 * b:10 * b + 10].sum(), 1))) for b in range(18)],
           "row_ratio_ref_over_build": {"0-5deg": float(row[:10].mean()), "10-12": float(row[20:24].mean()), "24-28": float(row[48:56].mean()),
                                        "45-49": float(row[90:98].mean()), "55-65": float(row[110:130].mean()), "75-85": float(row[150:170].mean())}}
    out.append(rec)
    print(json.dumps(rec), file=sys.stderr, flush=True)
print(json.dumps({"expected_sigma_of_chi2_per_dof": float(np.sqrt(2 / 15800)), "maps": out}, indent=1))
