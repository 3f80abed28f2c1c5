"""RESEARCH TOOL (dev container only: reads the reference's data files).  VERDICT r02 item 1: the three differences between
the reference's production maps and its single-threaded files -- reflectance (0.99 vs 1), roughness (0.01 vs off / 0.5) and
threads (4 vs 1) -- tabulated separately with the resolution each committed file allows.  Statistic: the amplitude `a` of the
production maps' residual pattern (1 = the full pattern of profiles/r02_parity_scan.md section 6, 0 = this build) projected
on each file.  python tools/research/factorise.py > profiles/r03_factorisation.txt"""
import numpy as np
import hyp

REF = "/root/reference/"
# the production pattern, reference / build - 1 in per cent (profiles/r02_parity_scan.md):
#   by exit-direction angle alpha from -z, 5-degree bands (section 3, port 170)
PAT_ALPHA = np.array([-3.6, -1.56, .34, 1.75, 2.4, 2.34, 1.85, 1.12, .29, -.51, -1.08, -1.33, -1.35, -1.34, -1.44, -1.61, -1.79, -1.98])
#   by detector angle theta, 5-degree bands (section 6, map pp_03_31_0: port 170, dir (5,0,0))
PAT_THETA = np.array([-1.05, -0.50, 0.54, 1.00, 1.46, 1.69, 1.68, 1.19, 1.06, 0.00, -1.02, -1.07, -0.99, -0.98, -0.78, -1.03, -3.88, -0.21])


def amplitude(r, s, pat, offset):
    """weighted least squares r = a pat (+ b): -> a, sigma_a, chi2 without / with the pattern"""
    W = 1.0 / s ** 2
    X = np.vstack([pat, np.ones(len(pat))]).T if offset else pat[:, None]
    cov = np.linalg.inv(X.T @ (W[:, None] * X))
    ab = cov @ (X.T @ (W * r))
    null = r - (np.average(r, weights=W) if offset else 0.0)
    return ab[0], np.sqrt(cov[0, 0]), float((W * null ** 2).sum()), float((W * (r - X @ ab) ** 2).sum())


rows = []
# ---- C, D: exit directions of distributionSphereDetectorSweep.C (rho = 1, no roughness call, single-threaded)
c = hyp.default_cfg(rho=1.0, sigma=0.0, box_half=200.0, src=[-60, 0, -80], max_points=10000, n_theta=2, n_phi=2)
h, st, dz, rad = hyp.run(c, 8_000_000, 5)
edges = np.linspace(-1, 1, 101)
cum = np.concatenate([[0], np.cumsum(dz)]).astype(float); cum /= cum[-1]
a_edges = np.arange(0, 95, 5.0)
model = np.diff(np.interp(-np.cos(np.radians(a_edges)), edges, cum))          # P(alpha band) of this build
d = np.loadtxt(REF + "3dRayLog.txt")
al = np.degrees(np.arccos(np.clip(-d[:, 2], -1, 1)))
hl = np.histogram(al, bins=18, range=(0, 90))[0].astype(float)
exp = model * len(d)
a, sa, c0, c1 = amplitude((hl / exp - 1) * 100, 100 / np.sqrt(exp), PAT_ALPHA, True)
rows.append(("3dRayLog.txt (1e5 exit directions, 18 alpha bands)", "1", "off", "1", a, sa, c0, c1, 17))
ad = np.loadtxt(REF + "angular_dist.txt")
ha = ad[:50, 1]                                                                  # dz in [-1, 0): 50 bins of 0.02
ours = dz[:50].astype(float); ours /= ours.sum()
centres = np.degrees(np.arccos(-(edges[:50] + 0.01)))                            # alpha of the bin centres
pat_dz = np.interp(centres, a_edges[:-1] + 2.5, PAT_ALPHA)
exp = ours * ha.sum()
a, sa, c0, c1 = amplitude((ha / exp - 1) * 100, 100 / np.sqrt(exp), pat_dz, True)
rows.append(("angular_dist.txt (1e5 rays, 50 dz bins)", "1", "off", "1", a, sa, c0, c1, 49))
# ---- F: flux_at_observer/fluxmap_data.csv (nonLambertianFlux.C revision without the re-scatter: rho = 1, roughness 0.5, one ray at a time)
f = np.loadtxt(REF + "flux_at_observer/fluxmap_data.csv", delimiter=",", skiprows=1)
frac = f[:, 2].reshape(45, 20)
n_ref = 100_000
c = hyp.default_cfg(rho=1.0, sigma=0.0, box_half=200.0, src=[-60, 0, -80], max_points=10000, n_theta=45, n_phi=20, det_diameter=10.0)
N = 6_000_000
h, st, dz2, rad = hyp.run(c, N, 7)
p = h / N
row_ref, row_our = frac.sum(1) * n_ref, p.sum(1) * n_ref                        # hits per theta row at the reference's statistics
use = row_our > 50
theta = (np.arange(45) + 0.5) * 2.0
pat_rows = np.interp(theta, np.arange(18) * 5 + 2.5, PAT_THETA)
r = (row_ref[use] / row_our[use] - 1) * 100
s = 100 * np.sqrt(1 / row_our[use] + 1 / (p.sum(1)[use] * N))
a, sa, c0, c1 = amplitude(r, s, pat_rows[use], False)
rows.append(("fluxmap_data.csv (45 theta rows, 1e5 rays per position, 10 cm detector)", "1", "0.5", "1", a, sa, c0, c1, int(use.sum()) - 1))
tot = row_ref.sum() / row_our.sum()
print("fluxmap_data.csv total reference / build = %.4f +- %.4f" % (tot, np.sqrt(1 / row_ref.sum() + 1 / (p.sum() * N))))
print()
print("| file | reflectance | roughness | threads | amplitude a of the production pattern | chi2 without -> with the pattern (dof) |")
print("|---|---|---|---|---|---|")
print("| seven per-position maps (8.1e8 rays each) | 0.99 | 0.01 | 4 | 1 (definition; each map fits the common pattern, chi2/dof 0.98-1.02) | - |")
for name, rho, sig, thr, a, sa, c0, c1, dof in rows:
    print("| %s | %s | %s | %s | %.2f +- %.2f | %.1f -> %.1f (%d) |" % (name, rho, sig, thr, a, sa, c0, c1, dof))
