"""numpy float32 restatement of the COLUMN-slot pre-selection of isx_bin_cols_kernel (isx_kernels.hpp: prep_shared, cap_is_low,
caps_may_touch, prep_cols, cap_rows for lines with caps; prep_band, band_rows for grazing lines -- same formulas, same slack terms),
checked against the brute-force exact test on exit lines from the oracle and on random lines: every hit must lie inside the rows
its column was given, and no bin may be handed out twice for one line (tests/test_cull_math.py).  Also a small tool:

  python tests/bandwin_np.py [brdf|headline|wide|random] [n_rays]      # candidates per line by kind of line

Test infrastructure only (imports the oracle)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import oracle as orc
from boxwin_np import acos_cull, exact_hits, lines_for, random_lines

f32 = np.float32
PI = f32(3.14159274)
HALF_PI = f32(1.57079637)


def atan2_cull(y, x):
    return np.arctan2(f32(y), f32(x)).astype(f32)   # (error 2e-5 in the kernel's polynomial; the slack covers it)


def sqrt_cull(x):
    return np.sqrt(f32(x)).astype(f32)


def cap_rows(fx, fy, a, cosw, c32, s32, inv_dth, n_theta):
    """rows [ilo, ilo + cnt) of every column (arrays c32, s32) inside the cap; vectorised over columns"""
    b = (c32 * fx + s32 * fy).astype(f32)
    rho2 = (a * a + b * b).astype(f32)
    ok = rho2 > f32(1e-12)
    x = (cosw / np.sqrt(np.where(ok, rho2, f32(1)))).astype(f32)
    ok &= ~(x > f32(1))
    ok0 = ok.copy()
    dl = (acos_cull(np.maximum(x, f32(-1))) + f32(2.5e-3)).astype(f32)
    tc = np.arctan2(b, np.full_like(b, a)).astype(f32)
    tlo, thi = (tc - dl).astype(f32), (tc + dl).astype(f32)
    ok &= ~((thi < 0) | (tlo > HALF_PI))
    lo = np.maximum(np.ceil(tlo * inv_dth - f32(0.5) - f32(1e-3)).astype(int), 0)
    hi = np.minimum(np.floor(thi * inv_dth - f32(0.5) + f32(1e-3)).astype(int), n_theta - 1)
    cnt = np.where(ok & (hi >= lo), hi - lo + 1, 0)
    wide = ok0 & (x < 0)                                              # wider than a quarter turn in the meridian's plane: every row
    cnt = np.where(wide, n_theta, cnt)
    return np.where((cnt > 0) & ~wide, lo, 0), cnt


def col_range(fx, fy, cosw, inv_dphi, n_phi, need_cosw_positive):
    """the columns a cap about (fx, fy, .) of opening cos w can reach -> (jlo, ncol)"""
    sinF = sqrt_cull(fx * fx + fy * fy)
    sinw = sqrt_cull(max(f32(0), f32(1) - cosw * cosw))
    if (need_cosw_positive and not cosw > f32(0.05)) or not (sinF > sinw * f32(1.01) + f32(1e-4)):
        return 0, n_phi
    r = min(f32(1), sinw / sinF * f32(1.001))
    dphi = (HALF_PI - acos_cull(np.array([r], f32))[0]) + f32(3e-3)
    phiF = atan2_cull(fy, fx)
    if phiF < 0:
        phiF = f32(phiF + f32(6.28318530718))
    jc = f32(phiF * inv_dphi - f32(0.5)); hw = f32(dphi * inv_dphi + f32(0.02))
    lo, hi = int(np.ceil(jc - hw)), int(np.floor(jc + hw))
    n = hi - lo + 1
    if n >= n_phi:
        return 0, n_phi
    return lo % n_phi, max(n, 0)


def line_slots(P, V, cfg):
    """-> (kind, list of (column j, first row, rows)) for ONE line, as isx_bin_cols_kernel's three passes produce them.
    kind: 'caps', 'band', 'far'."""
    n_phi, n_theta = cfg.n_phi, cfg.n_theta
    R, rho, portz = f32(cfg.det_distance), f32(cfg.det_diameter / 2), f32(cfg.exit_port_z)
    inv_dphi = f32(f32(n_phi) * f32(0.15915494309)); inv_dth = f32(f32(n_theta) * f32(0.63661977237))
    phi = (np.arange(n_phi) + 0.5) * 360.0 / n_phi * np.pi / 180
    c32, s32 = np.cos(phi).astype(f32), np.sin(phi).astype(f32)
    # ---- prep_shared
    wz = P[2] - float(portz)
    wv = P[0] * V[0] + P[1] * V[1] + wz * V[2]
    hx, hy, hz = P[0] - wv * V[0], P[1] - wv * V[1], wz - wv * V[2]
    dO2 = f32(hx * hx + hy * hy + hz * hz)
    R2 = f32(R * R)
    dO = sqrt_cull(dO2)
    a1 = f32(dO + rho)
    kind = -1
    if dO - rho > f32(1.001) * R:
        return "far", []
    sM = iL = ch2 = cosw = f32(0)
    if a1 < f32(0.999) * R:
        # the tube of radius rho about the line meets S(O,R), in the plane of the line and O, between the points
        # A1 = (dO + rho) h^ + smin V and A2 = (dO - rho) h^ + smx V: the cap about their bisector through both holds the whole
        # footprint; the bisector is the direction of the line's point at parameter (smin + smx)/2 from the foot
        am = f32(dO - rho)
        smin = sqrt_cull(R2 - a1 * a1)
        smx = sqrt_cull(R2 - am * am)
        sM = f32(f32(0.5) * f32(smin + smx))
        ds = f32(smx - smin)
        d2 = f32(f32(f32(4.0) * rho * rho + ds * ds) * f32(1.0001) + f32(4e-3))      # |A2 - A1|^2 = 4 R^2 sin^2 w
        s2w = f32(f32(0.25) * d2 / R2)
        cosw = f32(sqrt_cull(max(f32(0), f32(1) - s2w)) - f32(2e-6))
        iL = f32(1) / sqrt_cull(f32(dO2 + sM * sM))
        sinw = sqrt_cull(max(f32(0), f32(1) - cosw * cosw))
        if cosw > f32(0.5) and sM * iL > sinw * f32(1.01) + f32(1e-3):
            kind = 0
            ch2 = f32(f32(2.0) * R2 * f32(f32(1) - cosw))                              # chord^2 from the cap's centre to its rim
            ch = sqrt_cull(ch2)
    iR = f32(1) / R
    slots = []
    if kind == 0:
        def side(s):
            s0 = (float(sM) - wv) if s == 0 else (-float(sM) - wv)
            return s0, f32(f32(f32(wz + s0 * V[2]) * iL) * R)                          # z - portz of the cap's centre ON the sphere
        low2 = not (side(1)[1] - ch > 0)
        sinw = sqrt_cull(max(f32(0), f32(1) - cosw * cosw))
        touch = not (cosw > f32(0.05) and sM * iL > sinw * f32(1.0001) + f32(8e-3))
        if low2 and touch:
            kind = -1                                                   # the whole line as a grazing line
    if kind == 0:
        for s in ((0, 1) if low2 else (0,)):
            s0, Gz = side(s)
            if Gz - ch > 0:
                continue
            fx, fy = f32(f32(P[0] + s0 * V[0]) * iL), f32(f32(P[1] + s0 * V[1]) * iL)
            a = f32(-f32(wz + s0 * V[2]) * iL)
            jlo, ncol = col_range(fx, fy, cosw, inv_dphi, n_phi, False)
            js = (jlo + np.arange(ncol)) % n_phi
            ilo, cnt = cap_rows(fx, fy, a, cosw, c32[js], s32[js], inv_dth, n_theta)
            slots += [(int(j), int(i), int(n)) for j, i, n in zip(js, ilo, cnt) if n > 0]
        return "caps", slots
    # ---- prep_band
    Hx, Hy, Hz = f32(hx), f32(hy), f32(hz)
    Vx, Vy, Vz = f32(V[0]), f32(V[1]), f32(V[2])
    h = sqrt_cull(Hx * Hx + Hy * Hy + Hz * Hz)
    rs = f32(rho * f32(1.001) + f32(2e-3))
    kap_extra = f32(0)
    if h > f32(1e-3) * R:
        ex, ey, ez = f32(Hx / h), f32(Hy / h), f32(Hz / h)
        cosw = max(f32(-1), f32((h - rs) * iR - f32(4e-6)))
    else:
        kap_extra = f32(h * iR * f32(1.0001))   # H.u can reach h when h^ is a stand-in: the band is that much wider
        ax, ay, az = abs(Vx), abs(Vy), abs(Vz)
        t = np.zeros(3, f32)
        t[0 if (ax <= ay and ax <= az) else (1 if ay <= az else 2)] = 1
        tv = f32(t[0] * Vx + t[1] * Vy + t[2] * Vz)
        e = np.array([t[0] - tv * Vx, t[1] - tv * Vy, t[2] - tv * Vz], f32)
        e = (e / np.sqrt((e * e).sum())).astype(f32)
        ex, ey, ez = e
        cosw = f32(-1)
    fx, fy, a = ex, ey, f32(-ez)
    ux, uy, uz = f32(Vy * ez - Vz * ey), f32(Vz * ex - Vx * ez), f32(Vx * ey - Vy * ex)
    kap = f32(rs * iR * f32(1.0001) + f32(4e-6)) + kap_extra
    jlo, ncol = col_range(fx, fy, cosw, inv_dphi, n_phi, True)
    js = (jlo + np.arange(ncol)) % n_phi
    ilo, cnt = cap_rows(fx, fy, a, cosw, c32[js], s32[js], inv_dth, n_theta)
    for j, r_lo, n in zip(js, ilo, cnt):
        if n <= 0:
            continue
        r_hi = r_lo + n - 1
        ga = f32(c32[j] * ux + s32[j] * uy)
        N2 = f32(ga * ga + uz * uz)
        if not N2 > kap * kap * f32(1.03) + f32(1e-12):
            slots.append((int(j), int(r_lo), int(n)))
            continue
        x = f32(kap / np.sqrt(N2))
        al = f32((HALF_PI - acos_cull(np.array([x], f32))[0]) + f32(2.6e-3))
        de = atan2_cull(uz, ga)
        for m in (-1, 0, 1):
            ce = f32(f32(m) * PI + de)
            tlo, thi = f32(ce - al), f32(ce + al)
            if thi < 0 or tlo > HALF_PI:
                continue
            lo = max(int(np.ceil(tlo * inv_dth - f32(0.5) - f32(1e-3))), int(r_lo))
            hi = min(int(np.floor(thi * inv_dth - f32(0.5) + f32(1e-3))), int(r_hi))
            if hi >= lo:
                slots.append((int(j), lo, hi - lo + 1))
    return "band", slots


def check(c, lp, d, verbose=False):
    """-> dict(lines, hits, candidates, missed, twice) + the same per kind of line"""
    tab = orc.detector_table(c)
    out = dict(lines=0, hits=0, candidates=0, missed=0, twice=0, by_kind={})
    for P, V in zip(lp, d):
        h = exact_hits(P, V, tab, c.det_diameter).reshape(c.n_theta, c.n_phi)
        kind, slots = line_slots(P, V, c)
        cov = np.zeros((c.n_theta, c.n_phi), int)
        for j, i, n in slots:
            cov[i:i + n, j] += 1
        miss, twice = int((h & (cov == 0)).sum()), int((cov > 1).sum())
        if (miss or twice) and verbose:
            print("MISSED" if miss else "TWICE", miss, twice, kind, "line", P.tolist(), V.tolist())
        k = out["by_kind"].setdefault(kind, dict(lines=0, hits=0, candidates=0))
        k["lines"] += 1; k["hits"] += int(h.sum()); k["candidates"] += int(cov.sum())
        out["missed"] += miss; out["twice"] += twice; out["candidates"] += int(cov.sum()); out["hits"] += int(h.sum()); out["lines"] += 1
    return out


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "brdf"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    if kind == "random":
        c = orc.default_config()
        lp, d = random_lines(c, n)
    else:
        c, lp, d = lines_for(kind, n)
    r = check(c, lp, d, verbose=True)
    print(f"{kind}: lines {r['lines']}, hits/line {r['hits'] / r['lines']:.1f}, candidates/line {r['candidates'] / r['lines']:.1f}, "
          f"missed {r['missed']}, bins twice {r['twice']}")
    for k, v in r["by_kind"].items():
        print(f"  {k:5s}: {v['lines']} lines, hits/line {v['hits'] / max(v['lines'], 1):.1f}, candidates/line {v['candidates'] / max(v['lines'], 1):.1f}")


if __name__ == "__main__":
    main()
