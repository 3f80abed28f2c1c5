"""Round 5 (GPU): the reference's data files at the statistics only the GPU affords -- the same fixtures as
tests/test_oracle_golden.py (numbers taken from the reference's result files by tests/golden/make_golden.py), with this build's own
sampling noise made negligible, so that every comparison is limited by the REFERENCE's sample alone:
  * the one trace-once map at the older source position against the distribution of 400 independent 5e4-ray maps;
  * 3dRayLog.txt's 100 000 exit directions against 1e7 of this build's;
  * detector_sweep.txt / detector_sweep2.txt position by position (19 / 46 theta x 360 phi x 1000 rays), incl. their phi structure,
    which only addDetectorDisk's rotation quirk (LOG 2.4) reproduces."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = 20250509


def test_traceonce_map_at_the_older_source_position_is_one_draw_of_this_builds_distribution(isx, golden):
    g = golden["traceonce_src_m80"]
    c = isx.default_config()
    for k in range(3):
        c.src[k] = g["assumed"]["source_position"][k]; c.dir[k] = g["assumed"]["source_direction"][k]
    c.hit_line_mode = 1
    n, reps = g["rays"], 400
    sums, bands = [], []
    for s in range(reps):
        h, _ = isx.fluxmap(c, n, SEED + s)
        f = h / n
        sums.append(f.sum())
        bands.append(f.mean(axis=1)[:150].reshape(10, 15).mean(axis=1))
    sums, bands = np.array(sums), np.array(bands)
    z_sum = (g["sum_fraction"] - sums.mean()) / sums.std(ddof=1)
    gold = np.array(g["theta_profile"])[:150].reshape(10, 15).mean(axis=1)
    zb = (gold - bands.mean(axis=0)) / bands.std(axis=0, ddof=1)
    # Mahalanobis distance of the file's ten band means from the cloud of 400 maps (the bands share rays: correlated)
    cov = np.cov(bands.T)
    d2 = float((gold - bands.mean(axis=0)) @ np.linalg.solve(cov, gold - bands.mean(axis=0)))
    from scipy import stats
    print(f"sum: file {g['sum_fraction']:.3f}, this build {sums.mean():.3f} +- {sums.std(ddof=1):.3f} (z {z_sum:+.2f}); bands z {np.round(zb, 2)}; d2 {d2:.1f} / 10")
    assert abs(z_sum) < 3.5, z_sum
    assert np.abs(zb).max() < 4, zb
    assert stats.chi2.sf(d2 * (reps - 10) / (10 * (reps - 1)) * 10, 10) > 1e-4, d2     # (Hotelling's T^2 ~ chi2 for 400 samples)
    # the canonical hit line is excluded by the same file
    c.hit_line_mode = 0
    h, _ = isx.fluxmap(c, 2_000_000, SEED)
    assert (h.sum() / 2_000_000 - g["sum_fraction"]) / sums.std(ddof=1) > 20


def test_exit_log_against_1e7_directions(isx, golden):
    from scipy import stats
    g = golden["ray_log_3d"]
    c = isx.default_config()
    c.src[2] = -80.0; c.reflectance = 1.0; c.roughness_rad = 0.0; c.max_points = 10000; c.box_half = 200.0
    n = 10_000_000
    ids, d, cnt, st = isx.exit_directions(c, n, SEED + 31)
    # rho = 1: every ray leaves the box; a few per million leave it ABOVE the port plane (off the rim or the outer sphere) and are not logged
    assert cnt == st.counted_below_z and st.exited + st.suspended == n and n - cnt < 1000
    az = np.arctan2(d[:, 1], d[:, 0])
    Hr = np.array(g["dz_az_10x12"], dtype=float)
    Ho, _, _ = np.histogram2d(d[:, 2], az, bins=[np.linspace(-1.0, 0.0, 11), np.linspace(-np.pi, np.pi, 13)])
    p = Ho / Ho.sum()                                   # this build's cell probabilities, 1e7 directions: noise 1 % of the log's
    use = p * Hr.sum() >= 20
    exp = p * Hr.sum()
    chi2 = (((Hr - exp) ** 2 / (exp * (1 + Hr.sum() / Ho.sum())))[use]).sum()
    print(f"3dRayLog.txt, dz x azimuth: chi2 {chi2:.1f} for {use.sum() - 1} dof")
    assert stats.chi2.sf(chi2, use.sum() - 1) > 1e-4, (chi2, use.sum())
    for key, vals, lo, hi, nb in (("dz_hist", d[:, 2], -1.0, 0.0, 200), ("az_hist", az, -np.pi, np.pi, 180)):
        hr = np.array(g[key], dtype=float)
        ho = np.histogram(vals, bins=np.linspace(lo, hi, nb + 1))[0].astype(float)
        e = ho / ho.sum() * hr.sum()
        u = e >= 20
        x2 = (((hr - e) ** 2 / (e * (1 + hr.sum() / ho.sum())))[u]).sum()
        print(f"  {key}: chi2 {x2:.1f} for {u.sum() - 1} dof")
        assert stats.chi2.sf(x2, u.sum() - 1) > 1e-4, (key, x2, u.sum())


@pytest.mark.parametrize("which", ["disc_sweep", "disc_sweep2"])
def test_physical_disc_sweeps_position_by_position(isx, golden, which):
    """Every position of the reference's two committed disc sweeps (1000 rays each) against this build's hit probability of that
    position from 2e7 shared rays: theta rows (360 positions each), and theta groups x 30-degree phi sectors -- the phi structure
    exists only because addDetectorDisk's RotateZ / RotateY order leaves the tube axis in the x-z plane for every phi."""
    from scipy import stats
    g = golden[which]
    thetas = np.array(g["theta_deg"])
    ref = np.array(g["hits"], dtype=float)                           # [theta][phi 0..359]
    assert ref.shape == (len(thetas), 360)
    c = isx.default_config()
    c.r_out = 105.0; c.src[2] = -80.0; c.reflectance = 1.0; c.roughness_rad = 0.0; c.max_points = 10000; c.box_half = 200.0
    ca = []
    for t in thetas:
        for p in range(360):
            tr, pr = np.deg2rad(t), np.deg2rad(float(p))
            x, y, z = 200 * np.sin(tr) * np.cos(pr), 200 * np.sin(tr) * np.sin(pr), -200 * np.cos(tr)
            rot_theta = -np.arctan2(np.sqrt(x * x + y * y), -100.0 - z)
            ca.append([x, y, z, np.sin(rot_theta), 0.0, np.cos(rot_theta)])
    n = 20_000_000
    ca = np.array(ca)
    # (at most 8000 discs per call: the disc pipeline keeps the disc list and its cluster table in LDS; every call traces the SAME
    #  rays -- same seed, same indices -- so the chunks are one sweep)
    parts = []
    for a in range(0, len(ca), 7920):
        h, st = isx.disc_sweep(c, ca[a:a + 7920], 5.0, 0.1, n, SEED)
        assert isx.last_kernel_ms()[1] > 0 and isx.last_kernel_ms()[2] > 0      # trace kernel + disc-binning kernel, not the fused fallback
        parts.append(h)
    hits = np.concatenate(parts)
    p = hits.reshape(len(thetas), 360) / n
    rays = g.get("rays_per_position", 1000)
    exp = p * rays
    # theta rows
    z = (ref.sum(axis=1) - exp.sum(axis=1)) / np.sqrt(exp.sum(axis=1) * (1 + 360 * rays / n))
    print(f"{which}: theta rows z max {np.abs(z).max():.2f}, chi2/dof {(z ** 2).mean():.2f}, total ratio {exp.sum() / ref.sum():.4f}")
    assert np.abs(z).max() < 4.5 and (z ** 2).mean() < 2.0, z
    # theta groups x phi sectors
    ng = 4 if len(thetas) >= 40 else 2
    edges = np.linspace(0, len(thetas), ng + 1).astype(int)
    o = np.array([[ref[edges[a]:edges[a + 1], 30 * s:30 * s + 30].sum() for s in range(12)] for a in range(ng)])
    e = np.array([[exp[edges[a]:edges[a + 1], 30 * s:30 * s + 30].sum() for s in range(12)] for a in range(ng)])
    chi2 = ((o - e) ** 2 / e).sum()
    print(f"  {ng} theta groups x 12 phi sectors: chi2 {chi2:.1f} for {o.size} cells; sector sums ref {o.sum(axis=0).astype(int)} / this build {np.round(e.sum(axis=0)).astype(int)}")
    assert e.min() > 20 and stats.chi2.sf(chi2, o.size) > 1e-4, chi2
