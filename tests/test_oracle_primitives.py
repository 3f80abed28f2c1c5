"""Oracle primitives: RNG known answers, elementary functions vs libm, detector arithmetic.
(CPU only; pins the oracle before anything is compared with it.)"""
import ctypes as C
import math

import numpy as np


# Random123 kat_vectors for philox4x32 with 10 rounds
PHILOX_KAT = [
    ([0, 0, 0, 0], [0, 0], [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
    ([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
    ([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0],
     [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]),
]


def test_philox_known_answers(orc):
    for ctr, key, want in PHILOX_KAT:
        assert orc.philox(ctr, key) == want


def test_u01_open_interval(orc):
    L = orc.lib()
    assert L.isxo_u01(0) == 2.0 ** -33
    assert L.isxo_u01(0xFFFFFFFF) == 1.0 - 2.0 ** -33
    assert L.isxo_u01(0x80000000) == 0.5 + 2.0 ** -33


def test_log_accuracy(orc):
    L = orc.lib()
    rng = np.random.default_rng(11)
    xs = np.concatenate([rng.random(20000), 2.0 ** -rng.integers(1, 33, 2000) * (1 + rng.random(2000)) / 2])
    worst = 0.0
    for x in xs:
        if x <= 0:
            continue
        ref = math.log(x)
        if ref != 0:
            worst = max(worst, abs(L.isxo_log(float(x)) - ref) / abs(ref))
    assert worst < 4.5e-16


def test_sincos2pi_accuracy(orc):
    rng = np.random.default_rng(12)
    worst = 0.0
    for u in np.concatenate([rng.random(20000), [0.0, 0.125, 0.25, 0.375, 0.5, 0.625, 0.75, 0.875, 1 - 2.0 ** -33]]):
        s, c = orc.sincos2pi(float(u))
        worst = max(worst, abs(s - math.sin(2 * math.pi * u)), abs(c - math.cos(2 * math.pi * u)))
        assert abs(s * s + c * c - 1) < 5e-16
    assert worst < 1.5e-15  # includes the rounding of 2*pi*u in the libm argument


def test_sincos_accuracy(orc):
    rng = np.random.default_rng(13)
    worst = 0.0
    for x in np.concatenate([(rng.random(20000) - 0.5) * 40, [0.0, 1e-9, -1e-9, math.pi / 4, -math.pi / 2]]):
        s, c = orc.sincos(float(x))
        worst = max(worst, abs(s - math.sin(x)), abs(c - math.cos(x)))
    assert worst < 4e-16


def test_detector_table_matches_formula(orc):
    """Detector::setPosition (fluxAtObserver.C:49-68) evaluated independently with numpy/libm."""
    cfg = orc.default_config()
    tab = orc.detector_table(cfg).reshape(180, 90, 6)
    for i, j in [(0, 0), (17, 3), (90, 45), (179, 89), (120, 7)]:
        theta = (i + 0.5) * 90.0 / 180
        phi = (j + 0.5) * 360.0 / 90
        tr, pr = theta * math.pi / 180.0, phi * math.pi / 180.0
        x = 100.0 * math.sin(tr) * math.cos(pr)
        y = 100.0 * math.sin(tr) * math.sin(pr)
        z = -100.0 - 100.0 * math.cos(tr)
        mag = math.sqrt(x * x + y * y + (z + 100.0) * (z + 100.0))
        want = [x, y, z, -y / mag, x / mag, (z + 100.0) / mag]
        assert list(tab[i, j]) == want
    # the quirk the reference has: the normal is the radial direction rotated 90 deg about z
    n = tab[..., 3:6]
    d = tab[..., 0:3] - np.array([0, 0, -100.0])
    assert np.allclose(np.einsum("ijk,ijk->ij", n, d) / 100.0, np.cos(np.deg2rad((np.arange(180) + .5) * .5))[:, None] ** 2,
                       atol=1e-12)


def test_check_intersection_geometry(orc):
    """Hit iff the infinite line meets the detector plane within width/2 of the centre (fluxAtObserver.C:70-107)."""
    L = orc.lib()
    det = (C.c_double * 6)(0.0, 0.0, -200.0, 0.0, 0.0, -1.0)

    def hit(p, d, w=40.0):
        return L.isxo_check_intersection(det, w, (C.c_double * 3)(*p), (C.c_double * 3)(*d))

    assert hit((0, 0, -300), (0, 0, -1)) == 1          # behind the plane: still a hit (no t>0 test)
    assert hit((19.9, 0, -300), (0, 0, -1)) == 1
    assert hit((20.0, 0, -300), (0, 0, -1)) == 1       # boundary: r2 <= (w/2)^2
    assert hit((20.0000001, 0, -300), (0, 0, -1)) == 0
    assert hit((0, 0, -300), (1, 0, 0)) == 0            # parallel: |dot| < 1e-10
    assert hit((0, 0, -300), (1, 0, 1e-11)) == 0
    assert hit((5, 5, -150), (0.1, -0.2, -0.97), 10.0) == 0


def test_circle_point_is_a_uniform_unit_circle_point(orc):
    """isxo_circle_point(u) = (cos 4psi, sin 4psi), psi = (u - 1/2) pi/2: unit and exact to a few 1e-16,
    and therefore uniform on the circle when u is uniform."""
    rng = np.random.default_rng(21)
    w = np.concatenate([rng.integers(0, 2 ** 32, 50000, dtype=np.uint64), [0, 1, 2 ** 31 - 1, 2 ** 31, 2 ** 32 - 1]])
    u = (w.astype(np.float64) + 0.5) * 2.0 ** -32
    cs = np.array([orc.circle_point(float(x)) for x in u])
    assert np.abs((cs ** 2).sum(1) - 1).max() < 3e-15
    ang = 2 * np.pi * (u - 0.5)
    assert np.abs(cs[:, 0] - np.cos(ang)).max() < 2e-15 and np.abs(cs[:, 1] - np.sin(ang)).max() < 2e-15


def test_cosine_emission_is_the_cosine_law_about_the_normal(orc):
    """interact()'s emission n + s (s uniform on the unit sphere, world coordinates; oracle/isx_oracle.c cosine_emission):
    cos^2 of the polar angle about the geometric normal is uniform (the cosine law: P(cos <= x) = x^2), the azimuth about the
    normal is uniform and independent of it, nothing is emitted into the wall; on the inner sphere the vector is r_in s - q
    (un-normalised, |w| <= 2 r_in), elsewhere a unit vector."""
    from scipy import stats
    c = orc.default_config()
    rng = np.random.default_rng(41)
    rho_thr = int(np.ceil(c.reflectance * 2.0 ** 32 - 0.5))
    for kind, radius in ((1, c.r_in), (2, c.r_out)):
        # one surface point with a generic normal; the two words are the random input
        th, ph = 0.7, 2.1
        q = radius * np.array([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)])
        n = orc.surface_normal(c, kind, q)
        assert abs(np.linalg.norm(n) - 1) < 1e-12
        m = 40000
        wa = rng.integers(0, 2 ** 32, m, dtype=np.uint64)
        wb = rng.integers(0, rho_thr, m, dtype=np.uint64)
        w = np.array([orc.cosine_emission(c, kind, q, int(a), int(b)) for a, b in zip(wa, wb)])
        ln = np.linalg.norm(w, axis=1)
        if kind == 1:
            assert ln.max() <= 2 * c.r_in * (1 + 1e-12) and ln.min() > 0
        else:
            assert np.abs(ln - 1).max() < 4e-16
        u = w / ln[:, None]
        ct = u @ n
        assert ct.min() > 0                                            # never into the wall
        assert stats.kstest(ct * ct, "uniform").pvalue > 1e-3           # cosine law
        assert abs(ct.mean() - 2.0 / 3.0) < 4 * np.sqrt(1.0 / 18.0 / m)  # E cos = 2/3, Var cos = 1/18
        e1 = np.cross(n, [0.0, 0.0, 1.0]); e1 /= np.linalg.norm(e1)
        e2 = np.cross(n, e1)
        az = (np.arctan2(u @ e2, u @ e1) + np.pi) / (2 * np.pi)
        assert stats.kstest(az, "uniform").pvalue > 1e-3                # azimuth uniform
        assert abs(np.corrcoef(az, ct * ct)[0, 1]) < 4 / np.sqrt(m)     # ... and independent of the polar angle


def test_exit_directions_are_unit_vectors_in_both_trace_modes(orc):
    """The bounce works with un-normalised directions (rule S1' takes any length); a ray that leaves the rule is given a unit
    direction first, so what leaves the port -- the direction Detector::checkIntersection is handed -- is a unit vector to
    1 ulp, as ROBAST's is.  And the explicit and the chord mode visit the same wall points (to rounding)."""
    out = []
    for mode in (0, 1):
        c = orc.default_config()
        c.trace_mode = mode
        ids, d, cnt = orc.exit_directions(c, 40000, 77)
        assert cnt > 10000 and np.abs(np.linalg.norm(d, axis=1) - 1).max() < 3e-16
        out.append((ids, d))
    common = np.intersect1d(out[0][0], out[1][0])
    assert len(common) > 0.999 * max(len(out[0][0]), len(out[1][0]))   # the same rays leave ...
    a = out[0][1][np.isin(out[0][0], common)]
    b = out[1][1][np.isin(out[1][0], common)]
    assert np.abs(a - b).max() < 1e-9                                  # ... along the same directions
