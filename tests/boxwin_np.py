"""numpy float32 restatement of the per-row "box" column windows of the binning pre-selection (isx_kernels.hpp: box_line,
box_window -- same formulas, same slack terms), checked against the brute-force exact test on exit lines from the oracle:
every hit must lie inside its row's windows and no column may appear twice (tests/test_cull_math.py).  Also a small tool:

  python tests/boxwin_np.py [brdf|headline|wide|random] [n_rays]      # candidates per line, longest windows

Test infrastructure only (imports the oracle)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import oracle as orc

f32 = np.float32
PI = f32(3.14159274)


def acos_cull(x):
    x = x.astype(f32)
    ax = np.abs(x)
    p = f32(-0.0187293)
    for c in (0.0742610, -0.2121144, 1.5707288):
        p = (ax * p + f32(c)).astype(f32)
    r = (np.sqrt(np.maximum(f32(0), f32(1) - ax)).astype(f32) * p).astype(f32)
    return np.where(x < 0, PI - r, r).astype(f32)


def asin_cull(x):
    return (f32(1.57079637) - acos_cull(x)).astype(f32)


def atan2_cull(y, x):
    return np.arctan2(y.astype(f32), x.astype(f32)).astype(f32)   # (error 2e-5 in the kernel's version; slack covers it)


def box_windows(P, V, cfg, zrow, Arow):
    """Returns per row: (jL0, cntL, jU0, cntU) for ONE line; float32 arithmetic after the f64 foot point."""
    n_phi, n_theta = cfg.n_phi, cfg.n_theta
    R, rho, portz = f32(cfg.det_distance), f32(cfg.det_diameter / 2), cfg.exit_port_z
    wz = P[2] - portz
    wv = P[0] * V[0] + P[1] * V[1] + wz * V[2]
    hx, hy, hz = P[0] - wv * V[0], P[1] - wv * V[1], wz - wv * V[2]
    dO2 = f32(hx * hx + hy * hy + hz * hz)
    dO = np.sqrt(dO2)
    empty = (np.zeros(n_theta, int), np.zeros(n_theta, int), np.zeros(n_theta, int), np.zeros(n_theta, int))
    if dO - rho > f32(1.001) * R:
        return empty
    rs = rho * f32(1.001) + f32(2e-3)
    smax = np.sqrt(max(f32(0), (R + rs) * (R + rs) - dO2)) * f32(1.001) + f32(1e-2)
    a1 = dO + rs
    smin = f32(0) if a1 >= R else np.sqrt(R * R - a1 * a1) * f32(0.999) - f32(1e-2)
    Vx, Vy, Vz = f32(V[0]), f32(V[1]), f32(V[2])
    vxy2 = Vx * Vx + Vy * Vy
    if vxy2 > f32(1e-10):
        vxy = np.sqrt(vxy2)
        mx, my = Vx / vxy, Vy / vxy
    else:
        vxy = f32(0); mx, my = f32(1), f32(0)
    avz = abs(Vz)
    nx, ny = -my, mx
    Hx, Hy, Hz = f32(hx), f32(hy), f32(hz)
    dn = Hx * nx + Hy * ny
    sigma = -1.0          # phi = phi_n + sigma * psi for (n, m) with m = n rotated by -90 deg
    if dn < 0:
        nx, ny, dn, sigma = -nx, -ny, -dn, 1.0
    Hm = Hx * mx + Hy * my
    phin = np.arctan2(ny, nx).astype(f32)
    # per row
    z = zrow.astype(f32); A = Arow.astype(f32)
    zrel = z - f32(portz) - Hz
    rz = rs * vxy          # |e_z| <= rho |V_xy|
    if avz > f32(1e-3):
        iv = f32(1) / Vz
        s1 = (zrel - rz) * iv; s2 = (zrel + rz) * iv
        sa = np.minimum(s1, s2); sb = np.maximum(s1, s2)
        pad = f32(1e-4) * (np.abs(sa) + np.abs(sb)) + f32(1e-3)
        sa = sa - pad; sb = sb + pad
        ok = np.ones(n_theta, bool)
    else:
        ok = np.abs(zrel) <= rz + smax * avz + f32(1e-2)
        sa = np.full(n_theta, -smax, f32); sb = np.full(n_theta, smax, f32)
    sa = np.maximum(sa, -smax); sb = np.minimum(sb, smax)
    ok &= sa <= sb
    ok &= ~((sa > -smin) & (sb < smin))
    em = rs * avz          # |e . m| <= rho |V_z|
    ylo = Hm + sa * vxy - em; yhi = Hm + sb * vxy + em
    iA = f32(1) / A
    chi = (dn + rs) * iA; clo = (dn - rs) * iA
    ok &= clo <= 1
    dl = f32(1.5e-3)
    psi1 = np.maximum(f32(0), acos_cull(np.minimum(f32(1), chi)) - dl)
    psi2 = np.minimum(PI, acos_cull(np.maximum(f32(-1), clo)) + dl)
    psi1 = np.where(chi >= 1, f32(0), psi1)
    psi2 = np.where(clo <= -1, PI, psi2)

    def arc(sl, sh):
        """hull of {psi in [psi1, psi2] : sl <= sin psi <= sh}; returns lo, hi, nonempty"""
        ne = (sh >= 0) & (sl <= 1)
        al = np.where(sl <= 0, f32(0), asin_cull(np.clip(sl, 0, 1)) - dl)
        be = np.where(sh >= 1, f32(1.57079637), asin_cull(np.clip(sh, 0, 1)) + dl)
        a_lo, a_hi = np.maximum(al, psi1), np.minimum(be, psi2)
        b_lo, b_hi = np.maximum(PI - be, psi1), np.minimum(PI - al, psi2)
        ha, hb = a_lo <= a_hi, b_lo <= b_hi
        lo = np.where(ha, a_lo, b_lo); hi = np.where(hb, b_hi, a_hi)
        return lo, hi, ne & (ha | hb)
    p_lo, p_hi, p_ne = arc(ylo * iA, yhi * iA)       # psi > 0: m-coordinate +A sin psi
    q_lo, q_hi, q_ne = arc(-yhi * iA, -ylo * iA)     # psi < 0
    p_ne &= ok; q_ne &= ok
    if sigma > 0: u_lo, u_hi, u_ne, l_lo, l_hi, l_ne = p_lo, p_hi, p_ne, q_lo, q_hi, q_ne
    else: u_lo, u_hi, u_ne, l_lo, l_hi, l_ne = q_lo, q_hi, q_ne, p_lo, p_hi, p_ne
    inv = f32(n_phi * 0.15915494309)
    jU0 = np.ceil((phin + u_lo) * inv - f32(0.5) - f32(1e-2)).astype(int); jU1 = np.floor((phin + u_hi) * inv - f32(0.5) + f32(1e-2)).astype(int)
    jL0 = np.ceil((phin - l_hi) * inv - f32(0.5) - f32(1e-2)).astype(int); jL1 = np.floor((phin - l_lo) * inv - f32(0.5) + f32(1e-2)).astype(int)
    cU = np.where(u_ne, np.maximum(0, jU1 - jU0 + 1), 0); cL = np.where(l_ne, np.maximum(0, jL1 - jL0 + 1), 0)
    both = (cU > 0) & (cL > 0)
    # no column twice: trim L where it reaches into U (around psi = 0), trim U where it wraps into L (around psi = pi)
    over = both & (jL0 + cL - 1 >= jU0)
    cL = np.where(over, np.maximum(0, jU0 - jL0), cL)
    both = (cU > 0) & (cL > 0)
    over = both & (jU0 + cU - 1 >= jL0 + n_phi)
    cU = np.where(over, np.maximum(0, jL0 + n_phi - jU0), cU)
    cU = np.minimum(cU, n_phi); cL = np.minimum(cL, n_phi)
    return jL0, cL, jU0, cU


def exact_hits(P, V, tab, w):
    c, n = tab[:, :3], tab[:, 3:]
    dot = n @ V
    dd = P - c
    with np.errstate(divide='ignore', invalid='ignore'):
        t = -(np.einsum('ij,ij->i', dd, n)) / dot
    I = P + np.outer(t, V)
    r = I - c
    u = np.cross(n, r)
    return (np.abs(dot) >= 1e-10) & (np.einsum('ij,ij->i', u, u) <= (w / 2) ** 2)


def lines_for(kind, n):
    c = orc.default_config()
    if kind == "brdf":
        c.source_model = 1; c.brdf[0], c.brdf[1], c.brdf[2] = 0.3, 0.4, 0.6; c.roughness_rad = 0.5; c.reflectance = 1.0; c.max_points = 10000; c.box_half = 200.0
    elif kind == "wide":
        c.theta_max_deg = 160.0; c.dir[1] = 2.0
    st, npts, lp, d = orc.trace_endstates(c, n, 5)
    keep = (st == 1) & (lp[:, 2] < c.exit_port_z)
    return c, lp[keep], d[keep]


def random_lines(c, m, seed=1):
    """Lines in general position plus the degenerate families: horizontal, vertical, nearly horizontal, aimed at the shell."""
    rng = np.random.default_rng(seed)
    P = rng.uniform(-250, 250, (m, 3)); V = rng.standard_normal((m, 3))
    V[: m // 8, 2] = 0; V[m // 8: m // 4, :2] *= 1e-7; V[m // 4: m // 2, 2] *= 1e-3
    V /= np.linalg.norm(V, axis=1)[:, None]
    tgt = rng.standard_normal((m, 3)); tgt /= np.linalg.norm(tgt, axis=1)[:, None]; tgt *= c.det_distance
    tgt[:, 2] = c.exit_port_z - np.abs(tgt[:, 2])
    P[::2] = tgt[::2] - 150 * V[::2]
    return P, V


def check(c, lp, d, verbose=False):
    """-> dict(lines, hits, candidates, missed, twice, longest): the windows of every line against the exact test"""
    tab = orc.detector_table(c)
    th = (np.arange(c.n_theta) + 0.5) * 90.0 / c.n_theta * np.pi / 180
    zrow = c.exit_port_z - c.det_distance * np.cos(th); Arow = c.det_distance * np.sin(th)
    out = dict(lines=0, hits=0, candidates=0, missed=0, twice=0, longest=[])
    for P, V in zip(lp, d):
        h = exact_hits(P, V, tab, c.det_diameter).reshape(c.n_theta, c.n_phi)
        jL0, cL, jU0, cU = box_windows(P, V, c, zrow, Arow)
        cov = np.zeros((c.n_theta, c.n_phi), int)
        for i in range(c.n_theta):
            for j0, cc in ((jL0[i], cL[i]), (jU0[i], cU[i])):
                if cc > 0:
                    np.add.at(cov[i], (j0 + np.arange(cc)) % c.n_phi, 1)
        out["twice"] += int((cov > 1).sum())
        miss = int((h & (cov == 0)).sum())
        if miss and verbose:
            print("MISSED", miss, "hits; line", P.tolist(), V.tolist())
        out["missed"] += miss; out["candidates"] += int(cov.sum()); out["hits"] += int(h.sum()); out["lines"] += 1
        if cov.sum(): out["longest"].append(int((cL + cU).max()))
    return out


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "brdf"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    if kind == "random":
        c = orc.default_config()
        lp, d = random_lines(c, n)
    else:
        c, lp, d = lines_for(kind, n)
    r = check(c, lp, d, verbose=True)
    print(f"{kind}: lines {r['lines']}, hits/line {r['hits'] / r['lines']:.1f}, candidates/line {r['candidates'] / r['lines']:.1f}, "
          f"missed {r['missed']}, columns twice {r['twice']}, mean longest window {np.mean(r['longest']):.1f}")


if __name__ == "__main__":
    main()
