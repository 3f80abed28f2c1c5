"""The f32 helper functions of the binning pre-selection (isx_kernels.hpp: atan2_cull, acos_cull) never decide a
result, but the windows they produce must be conservative by the margins the kernel adds (1e-3 rad on the column
half-width, 2e-3 rad on the row range).  Their polynomials are restated here in numpy float32 and their error
claims checked against numpy's double-precision functions."""
import numpy as np

f32 = np.float32


def atan2_cull(y, x):
    y, x = y.astype(f32), x.astype(f32)
    ax, ay = np.abs(x), np.abs(y)
    mx, mn = np.maximum(ax, ay), np.minimum(ax, ay)
    a = np.where(mx > 0, mn / np.where(mx > 0, mx, f32(1)), f32(0)).astype(f32)
    s = (a * a).astype(f32)
    r = f32(-0.01172120)
    for c in (0.05265332, -0.11643287, 0.19354346, -0.33262347, 0.99997726):
        r = (s * r + f32(c)).astype(f32)
    r = (r * a).astype(f32)
    r = np.where(ay > ax, f32(1.57079637) - r, r).astype(f32)
    r = np.where(x < 0, f32(3.14159274) - r, r).astype(f32)
    return np.where(y < 0, -r, r).astype(f32)


def acos_cull(x):
    x = x.astype(f32)
    ax = np.abs(x)
    p = f32(-0.0187293)
    for c in (0.0742610, -0.2121144, 1.5707288):
        p = (ax * p + f32(c)).astype(f32)
    r = (np.sqrt(np.maximum(f32(0), f32(1) - ax)).astype(f32) * p).astype(f32)
    return np.where(x < 0, f32(3.14159274) - r, r).astype(f32)


def test_atan2_cull_error_below_2e5():
    rng = np.random.default_rng(0)
    y, x = rng.standard_normal(2_000_000) * 100, rng.standard_normal(2_000_000) * 100
    err = np.abs(atan2_cull(y, x).astype(np.float64) - np.arctan2(y.astype(f32).astype(np.float64), x.astype(f32).astype(np.float64)))
    assert err.max() < 2e-5
    # axes and diagonals
    for yy, xx in ((0.0, 1.0), (1.0, 0.0), (0.0, -1.0), (-1.0, 0.0), (1.0, 1.0), (-1.0, -1.0), (1e-30, 1.0)):
        got = float(atan2_cull(np.array([yy]), np.array([xx]))[0])
        assert abs(got - np.arctan2(yy, xx)) < 2e-5 or abs(abs(got - np.arctan2(yy, xx)) - 2 * np.pi) < 2e-5


def test_acos_cull_error_below_1e4():
    x = np.concatenate([np.linspace(-1, 1, 2_000_001), 1 - np.logspace(-8, 0, 2000), -1 + np.logspace(-8, 0, 2000)])
    err = np.abs(acos_cull(x).astype(np.float64) - np.arccos(x.astype(f32).astype(np.float64)))
    assert err.max() < 1e-4


def test_box_windows_contain_every_hit():
    """The per-row box windows of the binning pre-selection (isx_kernels.hpp box_line / box_window, restated in float32 numpy
    in tests/boxwin_np.py) against the exact test on all bins: exit lines of the headline and of the BRDF source model from the
    oracle, random lines including horizontal / vertical / nearly horizontal ones, and two other detector grids.  No hit
    outside its row's windows, no column in two windows of a row -- and the windows are worth having (few candidates)."""
    import boxwin_np as bw
    import oracle as orc
    c, lp, d = bw.lines_for("brdf", 300)
    r = bw.check(c, lp, d)
    assert r["lines"] > 250 and r["missed"] == 0 and r["twice"] == 0
    assert r["candidates"] < 4 * r["hits"]
    c, lp, d = bw.lines_for("headline", 400)
    r = bw.check(c, lp, d)
    assert r["lines"] > 100 and r["missed"] == 0 and r["twice"] == 0
    c = orc.default_config()
    lp, d = bw.random_lines(c, 400)
    r = bw.check(c, lp, d)
    assert r["hits"] > 10000 and r["missed"] == 0 and r["twice"] == 0
    for n_theta, n_phi, diam, dist in ((45, 20, 10.0, 100.0), (60, 120, 4.0, 150.0), (7, 3, 60.0, 80.0)):
        c = orc.default_config()
        c.n_theta, c.n_phi, c.det_diameter, c.det_distance = n_theta, n_phi, diam, dist
        lp, d = bw.random_lines(c, 300, seed=n_theta)
        r = bw.check(c, lp, d)
        assert r["hits"] > 0 and r["missed"] == 0 and r["twice"] == 0, (n_theta, n_phi)


def test_column_slots_contain_every_hit():
    """The COLUMN-slot pre-selection of isx_bin_cols_kernel (caps of the lines that pass near O -- both of them for a line whose
    second cap reaches detector rows, unless they could touch --, cap AND band for grazing lines; restated in float32 numpy in
    tests/bandwin_np.py) against the exact test on all bins: no hit outside the rows its column was given, no bin handed out
    twice for one line.  Exit lines of the headline, of a wide port and of the BRDF source from the oracle, random lines incl. the
    degenerate families, coarse grids (where row ranges are widest), detectors from 2 % to 150 % of their sphere's radius."""
    import bandwin_np as bw
    import boxwin_np as bx
    import oracle as orc
    c, lp, d = bx.lines_for("brdf", 400)
    r = bw.check(c, lp, d)
    assert r["lines"] > 300 and r["missed"] == 0 and r["twice"] == 0
    assert r["by_kind"]["band"]["lines"] > 80 and r["by_kind"]["band"]["candidates"] < 4.5 * r["by_kind"]["band"]["hits"]
    for kind in ("headline", "wide"):
        c, lp, d = bx.lines_for(kind, 400)
        r = bw.check(c, lp, d)
        assert r["lines"] > 100 and r["missed"] == 0 and r["twice"] == 0
        assert r["candidates"] < 1.8 * r["hits"]
    rng = np.random.default_rng(11)
    for n_theta, n_phi, diam, dist in ((180, 90, 40.0, 100.0), (45, 20, 10.0, 100.0), (60, 120, 4.0, 150.0), (7, 3, 60.0, 80.0), (1, 1, 40.0, 100.0),
                                       (2, 3, 30.0, 25.0), (3, 2, 12.0, 8.0), (5, 9, 150.0, 100.0), (17, 4, 90.0, 100.0), (90, 36, 6.0, 12.0)):
        c = orc.default_config()
        c.n_theta, c.n_phi, c.det_diameter, c.det_distance = n_theta, n_phi, diam, dist
        lp, d = bx.random_lines(c, 200, seed=n_theta + n_phi)
        # ... and lines aimed through the neighbourhood of O, where lines have two low caps
        m = 150
        tgt = np.array([0.0, 0.0, c.exit_port_z]) + rng.standard_normal((m, 3)) * dist * 0.4
        V = rng.standard_normal((m, 3)); V /= np.linalg.norm(V, axis=1)[:, None]
        lp = np.concatenate([lp, tgt - 120.0 * V]); d = np.concatenate([d, V])
        r = bw.check(c, lp, d)
        assert r["hits"] > 0 and r["missed"] == 0 and r["twice"] == 0, (n_theta, n_phi, diam, dist)
    # the degenerate branch of prep_band: detectors of 175-195 % of their sphere's radius (no caps: every line is a
    # band line) and lines within 1e-3 R of O, where a stand-in replaces h^ and the band needs the extra h / R
    for n_theta, n_phi, frac in ((180, 90, 1.76), (45, 20, 1.9), (17, 4, 1.95)):
        c = orc.default_config()
        c.n_theta, c.n_phi, c.det_distance = n_theta, n_phi, 100.0
        c.det_diameter = frac * c.det_distance
        m = 300
        off = rng.standard_normal((m, 3)); off /= np.linalg.norm(off, axis=1)[:, None]
        off *= rng.uniform(0.0, 1.5e-3, (m, 1)) * c.det_distance
        V = rng.standard_normal((m, 3)); V /= np.linalg.norm(V, axis=1)[:, None]
        tgt = np.array([0.0, 0.0, c.exit_port_z]) + off
        r = bw.check(c, tgt - 120.0 * V, V)
        assert r["hits"] > 0 and r["missed"] == 0 and r["twice"] == 0, (n_theta, n_phi, frac)
