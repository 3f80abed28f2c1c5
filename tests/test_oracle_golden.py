"""Pin the oracle against every fixture the reference's own result files offer.

The reference has no seeded tests and its trace arithmetic lives in ROBAST (absent), so these
are distribution-level checks; bit-level parity vs ROOT/ROBAST is "unpinned" (DESIGN.md §2).

Tolerances are the statistical resolution of the two samples being compared (pure binomial / Poisson sigmas, no
additive percentage floors).  Where round 2 measured a residual against the reference's 8.1e8-ray maps (totals 0.3-1.0 %
low, profiles/r02_parity_scan.md) the expected value of the comparison is that measured residual, stated in the test."""
import numpy as np
import pytest

SEED = 20250509


def _exit_fraction(orc, cfg, n):
    st, _, lp, _ = orc.trace_endstates(cfg, n, SEED)
    return float(((st == 1) & (lp[:, 2] < cfg.exit_port_z)).mean())


@pytest.mark.parametrize("port", [170.0, 164.0, 160.0])
def test_exit_counts_vs_traceonce_footers(orc, golden, port):
    """'Total rays exiting port: E out of 100000' (fluxAtObserverFast.C:1380) of 5/5/10 committed runs."""
    runs = [e["exited"] / e["n"] for e in golden["exit_counts"] if e["port_deg"] == port]
    assert len(runs) >= 5
    cfg = orc.default_config()
    cfg.theta_max_deg = port
    n = 400_000
    got = _exit_fraction(orc, cfg, n)
    ref, sd = np.mean(runs), np.std(runs, ddof=1)
    tol = 4 * np.hypot(sd / np.sqrt(len(runs)), np.sqrt(ref * (1 - ref) / n)) + 0.002
    assert abs(got - ref) < tol, (got, ref, tol)
    # closed form f/(1-rho(1-f)) (finitePort/test.py:11): the MC sits ~1 % below it
    f = (1 - np.cos(np.deg2rad(180 - port))) / 2
    assert 0.95 < got / (f / (1 - 0.99 * (1 - f))) < 1.005


def test_per_position_map_170(orc, golden):
    """results_overnight_03_31.../fluxmap_50000rays_180x90_src-60_0_-75.csv (8.1e8 reference rays)."""
    m = [m for m in golden["per_position_maps"] if m["port_deg"] == 170.0 and m["source_direction"] == [5.0, 0.0, 0.0]][0]
    n = 400_000
    h, st = orc.fluxmap(orc.default_config(), n, SEED)
    frac = h / n
    # total: this sample's own noise is 1.16/sqrt(n) = 0.18 % (hits per ray: 0 or ~270); known residual -0.3..-0.6 %
    # (symmetric about the reference's value: a model that closed the gap passes too; the gap's one marker is the strict xfail
    #  tests/test_gpu_round2.py::test_totals_of_the_reference_maps_within_0p15_percent)
    assert abs(frac.sum() / m["sum_fraction"] - 1) < 0.012
    assert st.bin_increments == int(h.sum())
    prof, gold = frac.mean(axis=1), np.array(m["theta_profile"])
    # noise of the reference rows (binomial, 90 x 50000 rays) + ours (correlated: ~n*p_exit/4 rays touch a row)
    sig_ref = np.sqrt(np.maximum(gold, 1e-7) / (m["rays_per_position"] * m["n_phi"]))
    sig_our = gold / np.sqrt(n * 0.42 * 0.2)
    z = np.abs(prof - gold) / np.hypot(sig_ref, sig_our)
    assert z.max() < 5, (z.max(), int(z.argmax()))
    # the documented quirk: detectors are edge-on at theta=90, the map collapses there
    assert prof[179] < 0.02 * prof[0]


def test_every_bin_of_the_170_map(orc):
    """All 16 200 bins of the reference's 8.1e8-ray map (tests/golden/reference_maps.npz) against 4e5 oracle rays, binomial
    sigmas: chi2/dof ~ 1.0-1.06 (the GPU tests do the same with 8.1e8 rays for all seven maps)."""
    import json
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_maps.npz"))
    info = [i for i in json.loads(str(z["index_json"])) if i["name"] == "pp_03_31_0"][0]
    ref = z["pp_03_31_0_hits"].astype(np.float64)
    assert ref.shape == (180, 90) and int(ref.sum()) == info["total_hits_footer"] == 5723365
    n, n_ref = 400_000, info["rays_per_position"]
    h, _ = orc.fluxmap(orc.default_config(), n, SEED + 1)
    p = (ref + h) / (n_ref + n)
    var = p * (1 - p) * (1.0 / n_ref + 1.0 / n)
    use = p * n_ref >= 5
    chi2 = (((ref / n_ref - h / n) ** 2)[use] / var[use]).sum() / use.sum()
    assert use.sum() > 15000 and chi2 < 1.12, chi2


def test_total_hits_vs_port_angle_and_direction(orc, golden):
    """'Total ray hits: H out of 810000000' footers for ports 163/166/169/172 and dirs (5,2,0),(5,4,0)."""
    n = 100_000
    for m in golden["per_position_maps"]:
        if m["port_deg"] == 170.0 and m["source_direction"] == [5.0, 0.0, 0.0]:
            continue
        cfg = orc.default_config()
        cfg.theta_max_deg = m["port_deg"]
        for k in range(3):
            cfg.dir[k] = m["source_direction"][k]
        h, _ = orc.fluxmap(cfg, n, SEED + int(m["port_deg"]))
        assert m["total_hits"] == pytest.approx(m["sum_fraction"] * m["rays_per_position"], rel=1e-4)
        # own noise 1.16/sqrt(n) = 0.37 %; known residual -0.4 % (port 172) .. -1.0 % (port 163)
        assert abs(h.sum() / n / m["sum_fraction"] - 1) < 0.025, (m["port_deg"], m["source_direction"])


def test_exit_direction_histogram(orc, golden):
    """angular_dist.txt: 100-bin histogram of the exit direction's z (distributionSphereDetectorSweep.C geometry)."""
    cfg = orc.default_config()
    cfg.src[2] = -80.0; cfg.reflectance = 1.0; cfg.roughness_rad = 0.0; cfg.max_points = 10000; cfg.box_half = 200.0
    gold = np.array(golden["angular_dist"]["content"], dtype=float)
    assert gold.size == 100 and golden["angular_dist"]["bin_centers"][0] == pytest.approx(-0.99)
    n = 400_000
    hist, st = orc.exit_dz_hist(cfg, n, SEED, 100)
    assert hist.sum() == st.counted_below_z
    p_ref, p_our = gold / gold.sum(), hist / hist.sum()
    assert hist[50:].sum() == 0 and gold[50:].sum() == 0          # nothing leaves upwards
    k = gold > 500
    z = np.abs(p_our - p_ref)[k] / np.sqrt(p_ref[k] / gold.sum() + p_ref[k] / hist.sum())
    assert z.max() < 5, z.max()
    # law ~ |dz| with the rim's collimation: first bins above the ideal-Lambert 0.0396, 0.0388
    assert p_our[0] > 0.0398 and p_our[0] == pytest.approx(p_ref[0], rel=0.04)


def test_sigma_half_map_pins_lambertian_ignores_roughness(orc, golden):
    """flux_at_observer/fluxmap_data.csv (45x20, 10 cm detector, sigma=0.5, rho=1): reproduced with the
    roughness not acting on the Lambertian border — the evidence behind DESIGN.md §2.3."""
    g = golden["nonlambertian_map"]
    cfg = orc.default_config()
    cfg.src[2] = -80.0; cfg.reflectance = 1.0; cfg.roughness_rad = 0.5; cfg.max_points = 10000; cfg.box_half = 200.0
    cfg.n_theta, cfg.n_phi, cfg.det_diameter = 45, 20, 10.0
    n = 600_000
    h, st = orc.fluxmap(cfg, n, SEED)
    assert st.absorbed == 0 and st.counted_below_z + (st.exited - st.counted_below_z) == n
    frac = h / n
    assert abs(frac.sum() / g["sum_fraction"] - 1) < 0.02
    prof, gold = frac.mean(axis=1), np.array(g["theta_profile"])
    sig = np.sqrt(np.maximum(gold, 2e-7) / (100000 * 20)) + gold / np.sqrt(n * 0.2)
    z = np.abs(prof - gold) / sig
    assert z.max() < 5, (z.max(), int(z.argmax()))


def _nl_cfg(orc, source_model):
    cfg = orc.default_config()
    cfg.src[2] = -80.0; cfg.reflectance = 1.0; cfg.roughness_rad = 0.5; cfg.max_points = 10000; cfg.box_half = 200.0
    cfg.n_theta, cfg.n_phi, cfg.det_diameter = 45, 20, 10.0
    cfg.source_model = source_model
    cfg.brdf[0], cfg.brdf[1], cfg.brdf[2] = 0.3, 0.4, 0.6
    return cfg


@pytest.mark.xfail(strict=True, reason="BRDF source model unpinned: the only committed nonLambertianFlux.C output is matched "
                                       "with the BRDF re-scatter OFF (sum ratio 0.997) and missed with it ON (0.81)")
def test_nonlambertian_file_with_the_brdf_rescatter_on(orc, golden):
    """run source_model = 1 (nonLambertianFlux.C:147-208,253-268 as committed) at the macro's own settings
    (45x20, 10 cm detector, src z = -80) against flux_at_observer/fluxmap_data.csv.  Measured in round 2 (4e6 rays):
    sum(fraction) 0.7673 vs the file's 0.9488 (ratio 0.809), chi2 per theta-row 1011; with the re-scatter off 0.9463
    (ratio 0.997), chi2 per row 1.10.  So that file was written by a revision WITHOUT the re-scatter (the committed
    nonLambertianFlux_C.so is older than the source), no reference data constrains the BRDF model, and
    BASELINE configs[2] is benchmarked on a restatement that only the source text pins."""
    g = golden["nonlambertian_map"]
    n = 600_000
    h, st = orc.fluxmap(_nl_cfg(orc, 1), n, SEED)
    assert abs(h.sum() / n / g["sum_fraction"] - 1) < 0.02


def test_physical_disc_sweep(orc, golden):
    """detector_sweep.txt: 5 cm disc at 200 cm (integratingSphereDetectorSweep.C), phi-mean per theta."""
    g = golden["disc_sweep"]
    cfg = orc.default_config()
    cfg.r_out = 105.0; cfg.src[2] = -80.0; cfg.reflectance = 1.0; cfg.roughness_rad = 0.0
    cfg.max_points = 10000; cfg.box_half = 200.0
    thetas = np.array(g["theta_deg"])
    phis = np.arange(0, 360, 15.0)
    ca = []
    for t in thetas:
        for p in phis:
            tr, pr = np.deg2rad(t), np.deg2rad(p)
            x, y, z = 200 * np.sin(tr) * np.cos(pr), 200 * np.sin(tr) * np.sin(pr), -200 * np.cos(tr)
            # addDetectorDisk (integratingSphereDetectorSweep.C:158-168): rot->RotateZ(rotPhi); rot->RotateY(rotTheta).
            # TGeoRotation::RotateY left-multiplies, so the tube axis is RY(rotTheta)*RZ(rotPhi)*ez =
            # (sin rotTheta, 0, cos rotTheta) for EVERY phi: the disc faces the port only at phi=0 — a
            # reference quirk this fixture (360 phi values per theta) is sensitive to.
            dx, dy, dz = 0 - x, 0 - y, -100.0 - z
            rot_theta = -np.arctan2(np.sqrt(dx * dx + dy * dy), dz)
            ca.append([x, y, z, np.sin(rot_theta), 0.0, np.cos(rot_theta)])
    n = 400_000
    hits, st = orc.disc_sweep(cfg, np.array(ca), 5.0, 0.1, n, SEED)
    got = (hits.reshape(len(thetas), len(phis)) / n).mean(axis=1)
    gold = np.array(g["phi_mean_fraction"])
    sig_ref = np.sqrt(np.maximum(gold, 1e-5) / (1000 * g["n_phi"]))     # 1000 rays x 360 phi per reference point
    sig_our = np.sqrt(np.maximum(got, 1e-6) / (n * len(phis)))
    z = np.abs(got - gold) / np.hypot(sig_ref, sig_our)
    assert z.max() < 5, (z.max(), got, gold)
    assert got[len(thetas) // 2] == pytest.approx(gold[len(thetas) // 2], rel=0.12)


def test_oracle_partition_invariance(orc):
    cfg = orc.default_config()
    full, s = orc.fluxmap(cfg, 30000, 5)
    a, sa = orc.fluxmap(cfg, 12345, 5, 0, 3)
    b, sb = orc.fluxmap(cfg, 30000 - 12345, 5, 12345, 1)
    assert np.array_equal(full, a + b)
    assert s.wall_hits == sa.wall_hits + sb.wall_hits


def test_origin_compat_reproduces_old_traceonce_files(orc, golden):
    """The committed fluxmap_traceonce_* maps differ from the per-position maps because GetPoint(nPoints-2, buf)
    never filled buf.  hit_line_mode=1 (start at the origin, direction lastPoint/|lastPoint|)
    reproduces them: on-axis 0.0091 instead of 0.0156, sum 78-79 instead of 114."""
    for t in golden["traceonce_maps"]:
        c = orc.default_config(); c.theta_max_deg = t["port_deg"]; c.hit_line_mode = 1
        n = 300_000
        h, _ = orc.fluxmap(c, n, SEED)
        frac = h / n
        assert abs(frac.sum() / t["sum_fraction_mean"] - 1) < 0.03
        prof, gold = frac.mean(axis=1), np.array(t["theta_profile_mean"])
        k = gold > 2e-4
        assert np.abs(prof[k] / gold[k] - 1).max() < 0.06


def test_lobe_surface_is_a_valid_diffuser(orc):
    """cos^2-lobe surface: unit directions, never into the wall, narrower than Lambert (more rays exit per bounce
    budget is not implied; just sanity + determinism)."""
    c = orc.default_config(); c.surface_model = 1
    st, npts, lp, d = orc.trace_endstates(c, 50000, 9)
    assert np.abs((d ** 2).sum(1) - 1).max() < 1e-14
    frac = ((st == 1) & (lp[:, 2] < -100)).mean()
    assert 0.30 < frac < 0.55
    st2, npts2, lp2, d2 = orc.trace_endstates(c, 50000, 9)
    assert np.array_equal(lp, lp2) and np.array_equal(npts, npts2)


def test_chord_mode_meets_the_same_fixtures(orc, golden):
    """ISX_TRACE_CHORD (integrating-sphere identity) against the reference's exit counts and 8.1e8-ray map."""
    for port in (170.0, 160.0):
        runs = [e["exited"] / e["n"] for e in golden["exit_counts"] if e["port_deg"] == port]
        c = orc.default_config(); c.theta_max_deg = port; c.trace_mode = 1
        got = _exit_fraction(orc, c, 400_000)
        assert abs(got - np.mean(runs)) < 0.004, (port, got)
    m = [m for m in golden["per_position_maps"] if m["port_deg"] == 170.0 and m["source_direction"] == [5.0, 0.0, 0.0]][0]
    c = orc.default_config(); c.trace_mode = 1
    n = 400_000
    h, _ = orc.fluxmap(c, n, SEED)
    frac = h / n
    assert abs(frac.sum() / m["sum_fraction"] - 1) < 0.012          # as in test_per_position_map_170
    prof, gold = frac.mean(axis=1), np.array(m["theta_profile"])
    sig_ref = np.sqrt(np.maximum(gold, 1e-7) / (m["rays_per_position"] * m["n_phi"]))
    sig_our = gold / np.sqrt(n * 0.42 * 0.2)
    z = np.abs(prof - gold) / np.hypot(sig_ref, sig_our)
    assert z.max() < 5, (z.max(), int(z.argmax()))


@pytest.mark.parametrize("mode", [0, 1])
def test_wall_points_uniform_on_sphere(orc, mode):
    """Known answer of the physics (no reference file needed): inside a Lambertian sphere the wall-hit points are
    uniform over the sphere's area, so the z of the points where rays get absorbed is uniform on [z_cut, r_in]
    (Archimedes).  Pins cosine emission + intersection in explicit mode and the direct sampling in chord mode."""
    c = orc.default_config(); c.trace_mode = mode
    st, npts, lp, _ = orc.trace_endstates(c, 300_000, 99)
    r = np.linalg.norm(lp, axis=1)
    sel = (st == 2) & (np.abs(r - c.r_in) < 1e-6) & (npts > 3)     # absorbed on the inner sphere, not at the first hit
    z = lp[sel, 2]
    zcut = c.r_in * np.cos(np.deg2rad(c.theta_max_deg))
    hist, _ = np.histogram(z, bins=40, range=(zcut, c.r_in))
    exp = sel.sum() / 40.0
    chi2 = ((hist - exp) ** 2 / exp).sum()
    assert chi2 < 80, chi2            # 39 dof: mean 39, 99.99 % quantile ~ 80
    # azimuth uniform too
    phi = np.arctan2(lp[sel, 1], lp[sel, 0])
    h2, _ = np.histogram(phi, bins=36, range=(-np.pi, np.pi))
    assert ((h2 - sel.sum() / 36.0) ** 2 / (sel.sum() / 36.0)).sum() < 75


# ------------------------------------------------------------------------------------------------ round 5: the reference's data files
# that no test had used so far ("next" #1; tests/golden/README.md lists every data file of the reference)
def test_traceonce_file_at_the_older_source_position_is_the_origin_compat_pattern(orc, golden):
    """results/fluxmap_traceonce_50000rays_180x90_src-60_0_-80.csv: ONE 50 000-ray trace-once map (all bins share the rays).  Its
    header records rays and detector size only; source (-60,0,-80), direction (5,2,0), port 170 are ASSUMED from the sibling files of
    the same evening (fixture: "assumed").  The file is one draw of a distribution of which the oracle makes independent draws:
    sum and theta profile must lie inside the oracle's own scatter with hit_line_mode = 1 -- and far outside with the canonical line
    (on-axis 0.0091 against 0.0155)."""
    g = golden["traceonce_src_m80"]
    assert g["rays"] == 50000 and g["row0"][0] == pytest.approx(0.00908)
    c = orc.default_config()
    for k in range(3):
        c.src[k] = g["assumed"]["source_position"][k]; c.dir[k] = g["assumed"]["source_direction"][k]
    c.hit_line_mode = 1
    sums, profs = [], []
    for s in range(16):
        h, _ = orc.fluxmap(c, g["rays"], SEED + 100 + s)
        sums.append(h.sum() / g["rays"]); profs.append((h / g["rays"]).mean(axis=1))
    sums, profs = np.array(sums), np.array(profs)
    z_sum = (g["sum_fraction"] - sums.mean()) / sums.std(ddof=1)
    assert abs(z_sum) < 4, (z_sum, g["sum_fraction"], sums.mean())            # measured: +0.8
    gold = np.array(g["theta_profile"])
    # ten bands of 15 rows (theta < 75 deg): each band's mean against the scatter of the oracle's sixteen maps
    zb = []
    for b in range(10):
        mine = profs[:, 15 * b:15 * b + 15].mean(axis=1)
        zb.append((gold[15 * b:15 * b + 15].mean() - mine.mean()) / mine.std(ddof=1))
    assert np.abs(zb).max() < 5, zb
    c.hit_line_mode = 0
    h, _ = orc.fluxmap(c, 200_000, SEED)
    assert (h.sum() / 200_000) / g["sum_fraction"] == pytest.approx(1.43, abs=0.04)   # the canonical line is 28 sigma away


@pytest.mark.parametrize("name", ["pp_03_31_3", "pp_04_1_4"])
def test_cut_short_per_position_maps(orc, name):
    """The two per-position runs that were interrupted (source direction (5,6,0): 12 867 rows; port 175 deg: 2 715 rows; 50 000 fresh
    rays per row, tests/golden/reference_maps.npz): every row present against 4e5 oracle rays, binomial sigmas.  A new direction and
    a new port angle; the known residual (a smooth factor of theta, +1.3 % on axis) shows in the 175-degree file, whose rows stop at
    theta = 15 deg: ratio 0.988 +- 0.004."""
    import json
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_maps.npz"))
    info = [i for i in json.loads(str(z["index_json"])) if i["name"] == name][0]
    ref = z[name + "_hits"].astype(np.int64).reshape(-1)
    have = ref >= 0
    assert not info["complete"] and have.sum() == info["rows_present"] and have[:info["rows_present"]].all()
    c = orc.default_config()
    c.theta_max_deg = info["port_deg"]
    for a in range(3):
        c.src[a] = info["source_position"][a]; c.dir[a] = info["source_direction"][a]
    n, n_ref = 400_000, info["rays_per_position"]
    h, _ = orc.fluxmap(c, n, SEED + 7)
    h = h.reshape(-1).astype(np.int64)
    p = (np.where(have, ref, 0) + h) / (n_ref + n)
    var = p * (1 - p) * (1.0 / n_ref + 1.0 / n)
    use = have & (p * n_ref >= 5)
    chi2 = (((ref / n_ref - h / n) ** 2)[use] / var[use]).sum() / use.sum()
    ratio = (h[have].sum() / n) / (ref[have].sum() / n_ref)
    assert use.sum() > 0.95 * have.sum() and chi2 < 1.15, (name, chi2)
    assert abs(ratio - 1) < 0.02, (name, ratio)


def test_unbinned_exit_log(orc, golden):
    """3dRayLog.txt (distributionSphereDetectorSweep.C:76-100: 100 000 exit directions, rho 1, no roughness, ONE thread): the joint
    distribution of (dz, azimuth) of the oracle's exit directions against the log's -- 2-d chi2 over 10 x 12 cells and the two 1-d
    distributions.  Measured: chi2 114 for 119 dof; KS p = 0.17 (dz), 0.11 (azimuth)."""
    from scipy import stats
    g = golden["ray_log_3d"]
    assert g["n"] == 100000 and g["upward"] == 0 and g["max_norm_error"] < 1e-5
    c = orc.default_config()
    c.src[2] = -80.0; c.reflectance = 1.0; c.roughness_rad = 0.0; c.max_points = 10000; c.box_half = 200.0
    n = 400_000
    ids, d, cnt = orc.exit_directions(c, n, SEED + 31)
    assert cnt == n and (d[:, 2] < 0).all()                                   # rho = 1: every ray leaves, downwards
    az = np.arctan2(d[:, 1], d[:, 0])
    Hr = np.array(g["dz_az_10x12"], dtype=float)
    Ho, _, _ = np.histogram2d(d[:, 2], az, bins=[np.linspace(-1.0, 0.0, 11), np.linspace(-np.pi, np.pi, 13)])
    assert Hr.sum() == g["n"] and Ho.sum() == n
    p = (Hr + Ho) / (Hr.sum() + Ho.sum())
    use = p * Hr.sum() >= 20
    chi2 = (((Hr / Hr.sum() - Ho / Ho.sum()) ** 2)[use] / (p * (1 / Hr.sum() + 1 / Ho.sum()))[use]).sum()
    assert stats.chi2.sf(chi2, use.sum() - 1) > 1e-4, (chi2, use.sum())
    for key, vals, lo, hi, nb in (("dz_hist", d[:, 2], -1.0, 0.0, 200), ("az_hist", az, -np.pi, np.pi, 180)):
        hr = np.array(g[key], dtype=float)
        ho = np.histogram(vals, bins=np.linspace(lo, hi, nb + 1))[0].astype(float)
        pp = (hr + ho) / (hr.sum() + ho.sum())
        u = pp * hr.sum() >= 20
        x2 = (((hr / hr.sum() - ho / ho.sum()) ** 2)[u] / (pp * (1 / hr.sum() + 1 / ho.sum()))[u]).sum()
        assert stats.chi2.sf(x2, u.sum() - 1) > 1e-4, (key, x2, u.sum())


def test_physical_disc_sweep_at_one_degree(orc, golden):
    """detector_sweep2.txt: the physical-disc sweep at 1 deg x 1 deg (46 complete theta rows, 1000 rays per position), phi-mean per
    theta: all rows within 4 sigma (measured: max 2.9, chi2/dof 1.2, total ratio 0.987) -- with the TGeoRotation quirk of DESIGN 2.4."""
    g = golden["disc_sweep2"]
    cfg = orc.default_config()
    cfg.r_out = 105.0; cfg.src[2] = -80.0; cfg.reflectance = 1.0; cfg.roughness_rad = 0.0
    cfg.max_points = 10000; cfg.box_half = 200.0
    thetas, phis = np.array(g["theta_deg"]), np.arange(0, 360, 15.0)
    assert len(thetas) == 46 and g["rows_dropped"] == 219
    ca = []
    for t in thetas:
        for p in phis:
            tr, pr = np.deg2rad(t), np.deg2rad(p)
            x, y, z = 200 * np.sin(tr) * np.cos(pr), 200 * np.sin(tr) * np.sin(pr), -200 * np.cos(tr)
            rot_theta = -np.arctan2(np.sqrt(x * x + y * y), -100.0 - z)
            ca.append([x, y, z, np.sin(rot_theta), 0.0, np.cos(rot_theta)])
    n = 400_000
    hits, _ = orc.disc_sweep(cfg, np.array(ca), 5.0, 0.1, n, SEED)
    got = (hits.reshape(len(thetas), len(phis)) / n).mean(axis=1)
    gold = np.array(g["phi_mean_fraction"])
    sig = np.hypot(np.sqrt(np.maximum(gold, 1e-5) / (g["rays_per_position"] * g["n_phi"])), np.sqrt(np.maximum(got, 1e-6) / (n * len(phis))))
    zz = (got - gold) / sig
    assert np.abs(zz).max() < 4 and (zz ** 2).mean() < 1.8, (np.abs(zz).max(), (zz ** 2).mean())
    assert got.sum() / gold.sum() == pytest.approx(1.0, abs=0.04)


def test_small_and_cut_short_files_of_identified_revisions(orc, golden):
    """Six small files whose macro revision the file name / header identifies (fluxAtObserver.C: src z -80, dir (5,2,0), rho 1,
    sigma 0.5, 10 cm detector; fluxAtObserverOptimize/Fast.C at src z -80): the hits of the rows present against the oracle's bin
    probabilities -- the sum within 4 sigma of its Poisson noise (twofold rows share their rays in pairs: sigma x sqrt 2).
    Measured ratios oracle / file: 1.000, 1.014, 0.993, 1.016, 1.003, 1.026."""
    cache = {}
    for f in golden["small_files"]:
        cf = f["config"]
        key = (cf["macro"], cf["n_theta"], cf["n_phi"], cf["det_diameter"])
        if key not in cache:
            c = orc.default_config()
            for k in range(3):
                c.src[k] = cf["src"][k]; c.dir[k] = cf["dir"][k]
            c.reflectance, c.roughness_rad, c.max_points, c.box_half = cf["reflectance"], cf["roughness"], cf["max_points"], cf["box_half"]
            c.n_theta, c.n_phi, c.det_diameter = cf["n_theta"], cf["n_phi"], cf["det_diameter"]
            n = 300_000
            h, _ = orc.fluxmap(c, n, SEED + len(cache))
            cache[key] = h / n
        p = cache[key]
        th, ph, hits = np.array(f["theta"]), np.array(f["phi"]), np.array(f["hits"], dtype=float)
        i = np.rint(th / (90.0 / cf["n_theta"]) - 0.5).astype(int)
        j = np.rint(ph / (360.0 / cf["n_phi"]) - 0.5).astype(int)
        assert np.abs((i + 0.5) * 90.0 / cf["n_theta"] - th).max() < 1e-4 and np.abs((j + 0.5) * 360.0 / cf["n_phi"] - ph).max() < 1e-4
        expect = p[i, j] * f["rays_per_position"]
        sig = np.sqrt(expect.sum() * (2.0 if cf.get("fold") == 2 else 1.0) * (1.0 + f["rays_per_position"] * len(hits) / 300_000 * 0.01))
        z = (hits.sum() - expect.sum()) / sig
        assert abs(z) < 4, (f["file"], hits.sum(), expect.sum(), z)


@pytest.mark.xfail(strict=True, reason="results/detector_data_50000rays*.csv come from an older revision of fluxAtObserver.C whose source and "
                                       "detector model are not recorded: with today's detector normal the sum ratio is 0.951 (10 cm disc) / 3.80 "
                                       "(20 cm disc) and rows beyond 60 deg are 0.25-0.68 of the file's; with normals facing the port 1.40 "
                                       "(2.5-3.4 beyond 50 deg).  No second check of DESIGN 2.3 can be drawn from them")
def test_detector_data_files_of_the_older_revision(orc, golden):
    """VERDICT r04 'next' 1(b): sigma = 0.75, '20cm x 20cm', Lambertian.  Kept as a strict xfail that states the measured ratios."""
    g = golden["detector_data_sigma075"]
    assert g["roughness"] == 0.75 and len(g["coarse"]) == 4
    c = orc.default_config()
    c.det_diameter = 10.0; c.roughness_rad = 0.75
    n = 300_000
    h, _ = orc.fluxmap(c, n, SEED)
    prof, gold = (h / n).mean(axis=1), np.array(g["theta_profile"])
    assert abs((h.sum() / n) / g["sum_fraction"] - 1) < 0.02
    assert abs(prof[120:].sum() / gold[120:].sum() - 1) < 0.1
