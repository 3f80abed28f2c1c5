"""Reader + replay of a reference-side bounce dump (format of tools/ref_dump/dumpBounces.C).

Test infrastructure.  `replay()` walks every ray of a dump through the CPU oracle one bounce at a time:
  * navigation  -- from track point k along the dumped segment, the oracle's boundary search must land on track point k+1
                   and on the same kind of surface;
  * emission    -- at every mirror point the cosine between the outgoing segment and the oracle's surface normal must be
                   sqrt(1-u) (or sqrt(u)) of ONE of the uniforms the generator handed out for that ray, the same form for all
                   bounces, in log order: the cosine law about the GEOMETRIC normal, pinned draw by draw;
  * final state -- the dumped direction is the last segment's direction, the end status matches where the track stops;
  * detector    -- Detector::checkIntersection's dumped answers equal the oracle's, bit for bit.
`write_synthetic()` writes a file of the same format from the oracle's own trace: the committed
tests/golden/robast_bounces_synthetic.txt keeps the reader and the replay honest while no ROBAST dump exists.  The oracle
draws its cosine-law direction as n + s (s uniform on the unit sphere, two Philox words; oracle/isx_oracle.c interact()),
not from a polar angle, so the "polar uniform" the synthetic file logs for an emission is the one a sqrt(1-u) sampler would
have needed for the same direction, u = 1 - cos^2; navigation, end states and detector answers are the oracle's own.
"""
import numpy as np

import oracle

K_NONE, K_INNER, K_OUTER, K_CONE, K_BOX = 0, 1, 2, 3, 4


def read(path):
    head, dets, rays = {}, [], []
    with open(path) as f:
        for ln in f:
            t = ln.split()
            if not t:
                continue
            if t[0] == "#":
                if len(t) > 1 and t[1] == "detector":
                    v = [float(x) for x in t[8:14]]      # "# detector k theta T phi P : x y z nx ny nz width W"
                    dets.append({"theta": float(t[4]), "phi": float(t[6]), "det": v, "width": float(t[15])})
                elif len(t) > 1 and t[1] == "seed":
                    keys = {"seed": 1, "rays": 1, "reflectance": 1, "roughness": 1, "theta_max": 1, "r_in": 1, "r_out": 1,
                            "box_half": 1, "limit": 1, "port_z": 1, "src": 3, "dir": 3, "cm": 1}
                    i = 1
                    while i < len(t):
                        k = t[i]
                        n = keys[k]
                        head[k] = [float(x) for x in t[i + 1:i + 1 + n]] if n > 1 else float(t[i + 1])
                        i += 1 + n
                elif len(t) > 1 and t[1] == "synthetic":
                    head["synthetic"] = True
                continue
            if t[0] == "R":
                rays.append({"index": int(t[1]), "npoints": int(t[2]), "status": t[3], "points": [], "u": [], "in_gaus": [],
                             "gaus": [], "dir": None, "flags": []})
            elif t[0] == "P":
                rays[-1]["points"].append([float(x) for x in t[2:5]])
            elif t[0] == "U":
                for x in t[1:]:
                    rays[-1]["in_gaus"].append(x.startswith("g"))
                    rays[-1]["u"].append(float(x.lstrip("g")))
            elif t[0] == "G":
                rays[-1]["gaus"] = [float(x) for x in t[1:]]
            elif t[0] == "D":
                rays[-1]["dir"] = [float(x) for x in t[1:4]]
                rays[-1]["flags"] = [int(x) for x in t[4:]]
    return head, dets, rays


def config_of(head):
    c = oracle.default_config()
    cm = head.get("cm", 1.0)
    c.reflectance = head["reflectance"]; c.roughness_rad = head["roughness"]; c.theta_max_deg = head["theta_max"]
    c.r_in = head["r_in"] / cm; c.r_out = head["r_out"] / cm; c.box_half = head["box_half"] / cm
    c.max_points = int(head["limit"]); c.exit_port_z = head["port_z"] / cm; c.lambertian = 1
    for k in range(3):
        c.src[k] = head["src"][k] / cm
        c.dir[k] = head["dir"][k]
    return c


def classify(c, p, tol=1e-6):
    """which modelled surface a track point lies on"""
    r = float(np.linalg.norm(p))
    if max(abs(p[0]), abs(p[1]), abs(p[2])) > c.box_half - tol:
        return K_BOX
    tmax = np.deg2rad(c.theta_max_deg)
    if abs(r - c.r_in) < tol and p[2] >= c.r_in * np.cos(tmax) - tol:
        return K_INNER
    if abs(r - c.r_out) < tol and p[2] >= c.r_out * np.cos(tmax) - tol:
        return K_OUTER
    if c.r_in - tol <= r <= c.r_out + tol and p[2] < 0 and abs(np.hypot(p[0], p[1]) - abs(np.tan(tmax) * p[2])) < tol * 10:
        return K_CONE
    return -1


def replay(path, pos_tol=1e-7, cos_tol=1e-9):
    """-> summary dict; raises AssertionError with the first ray / bounce that does not replay."""
    head, dets, rays = read(path)
    c = config_of(head)
    cm = head.get("cm", 1.0)
    forms = {"sqrt(1-u)": 0, "sqrt(u)": 0}
    n_seg = n_emit = n_flags = 0
    draws_per_bounce = []
    for ray in rays:
        P = np.array(ray["points"]) / cm
        assert len(P) == ray["npoints"] >= 2, ray["index"]
        assert np.allclose(P[0], [c.src[0], c.src[1], c.src[2]], atol=1e-12), ray["index"]
        u = np.array(ray["u"])
        on, cursor, emitted = K_NONE, 0, 0
        for k in range(len(P) - 1):
            seg = P[k + 1] - P[k]
            d = seg / np.linalg.norm(seg)
            kind, q = oracle.next_boundary(c, P[k], d, on)
            want = classify(c, P[k + 1])
            assert want >= 0, (ray["index"], k + 1, "track point on no modelled surface", P[k + 1])
            assert kind == want and np.linalg.norm(q - P[k + 1]) < pos_tol, (ray["index"], k, kind, want, q, P[k + 1])
            n_seg += 1
            if k >= 1:   # the segment leaves a mirror point: cosine law about the geometric normal, one uniform per bounce
                n = oracle.surface_normal(c, on, P[k])
                ct = float(np.dot(d, n))
                assert ct > 0, (ray["index"], k, "emission into the wall", ct)
                rest = u[cursor:]
                hit = None
                for name, val in (("sqrt(1-u)", np.sqrt(1.0 - rest)), ("sqrt(u)", np.sqrt(rest))):
                    j = np.nonzero(np.abs(val - ct) < cos_tol)[0]
                    if len(j):
                        hit = (name, int(j[0]))
                        break
                assert hit is not None, (ray["index"], k, "no logged uniform gives this polar angle", ct)
                forms[hit[0]] += 1
                cursor += hit[1] + 1
                emitted += 1
                n_emit += 1
            on = want
        last = P[-1] - P[-2]
        last /= np.linalg.norm(last)
        if ray["status"] in ("E", "S"):
            assert np.allclose(ray["dir"], last, atol=1e-12), (ray["index"], "final direction is not the last segment's")
            assert classify(c, P[-1]) == K_BOX or ray["status"] == "S", ray["index"]
        if emitted:
            draws_per_bounce.append(len(u) / (emitted + 1.0))
        lp = P[-1] * cm
        for det, flag in zip(dets, ray["flags"]):
            mine = oracle.check_intersection(det["det"], det["width"], lp, ray["dir"]) if lp[2] < head["port_z"] else 0
            assert mine == flag, (ray["index"], det["theta"], det["phi"], mine, flag)
            n_flags += 1
    assert min(forms.values()) == 0, ("the polar angle uses two different transforms of the uniforms", forms)
    return {"rays": len(rays), "segments": n_seg, "emissions": n_emit, "polar_form": max(forms, key=forms.get), "detector_flags": n_flags,
            "uniforms_per_interaction": float(np.mean(draws_per_bounce)) if draws_per_bounce else 0.0,
            "synthetic": bool(head.get("synthetic"))}


def write_synthetic(path, n_rays=24, seed=12345):
    """The oracle's own rays in the dump format (Philox words as the 'uniforms'); every end state is cross-checked against
    isxo_trace_endstates, so the per-bounce hooks used here and in replay() are the oracle's trace loop, not a look-alike."""
    c = oracle.default_config()
    probes = [(0.25, 2.0), (20.25, 46.0), (45.25, 182.0), (70.25, 270.0), (89.75, 358.0)]
    grid = oracle.default_config(); grid.n_theta, grid.n_phi = 360, 180   # theta = (i+.5)/4 deg, phi = (j+.5)*2 deg
    table = oracle.detector_table(grid)
    dets = [table[int(round(t * 4 - 0.5)) * 180 + int(round(p / 2 - 0.5))] for t, p in probes]
    st, npts, lps, dirs = oracle.trace_endstates(c, n_rays, seed)
    rho_thr = int(np.ceil(c.reflectance * 2.0 ** 32 - 0.5))
    with open(path, "w") as f:
        f.write("# isx-robast-bounce-dump 1\n# synthetic : written by tests/robast_dump.py from the CPU oracle, NOT by ROBAST "
                "(U: per emission 1 - cos^2 of its polar angle, then the azimuth uniform)\n")
        f.write("# seed %d rays %d reflectance %.17g roughness %.17g theta_max %.17g r_in %.17g r_out %.17g box_half %.17g limit %d "
                "port_z %.17g src %.17g %.17g %.17g dir %.17g %.17g %.17g cm 1\n" % (
                    seed, n_rays, c.reflectance, c.roughness_rad, c.theta_max_deg, c.r_in, c.r_out, c.box_half, c.max_points,
                    c.exit_port_z, c.src[0], c.src[1], c.src[2], c.dir[0], c.dir[1], c.dir[2]))
        for k, (t, p) in enumerate(probes):
            f.write("# detector %d theta %.17g phi %.17g : %s width %.17g\n" % (k, t, p, " ".join("%.17g" % x for x in dets[k]), c.det_diameter))
        for i in range(n_rays):
            p = np.array([c.src[0], c.src[1], c.src[2]])
            v = np.array([c.dir[0], c.dir[1], c.dir[2]]) / np.linalg.norm([c.dir[0], c.dir[1], c.dir[2]])
            on, j, pts, us, status = K_NONE, 0, [p.copy()], [], "?"
            while True:
                kind, q, v = oracle.next_boundary(c, p, v, on, with_direction=True)   # (v: the unit direction once S1' is left)
                p = q; pts.append(p.copy())
                if kind == K_BOX:
                    status = "E"; break
                on = kind
                w = oracle.philox([i & 0xffffffff, i >> 32, j >> 1, 0], [seed & 0xffffffff, seed >> 32])
                wa, wb = w[2 * (j & 1)], w[2 * (j & 1) + 1]
                j += 1
                if not wb < rho_thr:
                    us.append((wb + 0.5) / 2.0 ** 32); status = "A"; break
                n = oracle.surface_normal(c, kind, p)
                v = oracle.cosine_emission(c, kind, p, wa, wb)                 # ~ n + s (un-normalised on the inner sphere)
                ct = float(np.dot(v, n) / np.linalg.norm(v))
                us += [1.0 - ct * ct, (wb + 0.5) * (1.0 / rho_thr)]
                if len(pts) > c.max_points:
                    status = "U"; break
            want = {1: "E", 2: "A", 3: "U"}[int(st[i])]
            assert status == want and len(pts) == npts[i] and np.array_equal(pts[-1], lps[i]), (i, status, want, len(pts), npts[i])
            if status == "E":
                assert np.array_equal(v, dirs[i]), i
            f.write("R %d %d %s %d 0\n" % (i, len(pts), status, len(us)))
            for k, q in enumerate(pts):
                f.write("P %d %.17g %.17g %.17g\n" % (k, q[0], q[1], q[2]))
            f.write("U " + " ".join("%.17g" % x for x in us) + "\nG\n")
            flags = [oracle.check_intersection(d, c.det_diameter, pts[-1], v) if (status == "E" and pts[-1][2] < c.exit_port_z) else 0 for d in dets]
            f.write("D %.17g %.17g %.17g %s\n" % (v[0], v[1], v[2], " ".join(str(x) for x in flags)))
