"""ctypes driver of tools/research/hyp.c (RESEARCH TOOL: hypothesis scan against the reference's maps)."""
import ctypes as C
import json
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


class Cfg(C.Structure):
    _fields_ = [
        ("r_in", C.c_double), ("r_out", C.c_double), ("theta_max_deg", C.c_double), ("rho", C.c_double),
        ("sigma", C.c_double), ("box_half", C.c_double), ("src", C.c_double * 3), ("dir", C.c_double * 3),
        ("n_theta", C.c_int), ("n_phi", C.c_int), ("det_diameter", C.c_double), ("det_distance", C.c_double),
        ("port_z", C.c_double), ("max_points", C.c_int),
        ("law", C.c_int), ("law_pow", C.c_double), ("rough_lambert", C.c_int), ("rim", C.c_int),
        ("rho_rim", C.c_double), ("outer", C.c_int), ("absorb_after", C.c_int), ("step_back", C.c_double),
        ("count_absorbed", C.c_int), ("hit_line", C.c_int), ("first_specular", C.c_int), ("rho_angle_k", C.c_double),
        ("retry_into_wall", C.c_int), ("det_normal_mode", C.c_int), ("port_test", C.c_int), ("two_sided", C.c_int),
        ("replay_q", C.c_double), ("replay_what", C.c_int),
    ]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("launched", "exited", "counted", "absorbed", "suspended", "wall_hits", "rim_hits", "outer_hits",
                 "increments")]


def default_cfg(**kw):
    c = Cfg()
    c.r_in, c.r_out, c.theta_max_deg, c.rho, c.sigma, c.box_half = 100.1, 101.0, 170.0, 0.99, 0.01, 300.0
    c.src[:] = [-60, 0, -75]
    c.dir[:] = [5, 0, 0]
    c.n_theta, c.n_phi, c.det_diameter, c.det_distance, c.port_z, c.max_points = 180, 90, 40.0, 100.0, -100.0, 50000
    c.rho_rim = -1.0
    for k, v in kw.items():
        if k in ("src", "dir"):
            getattr(c, k)[:] = v
        else:
            setattr(c, k, v)
    return c


_lib = None


def lib():
    global _lib
    if _lib is None:
        so, src = os.path.join(HERE, "libhyp.so"), os.path.join(HERE, "hyp.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["gcc", "-O3", "-march=native", "-fopenmp", "-shared", "-fPIC", "-o", so, src, "-lm"])
        _lib = C.CDLL(so)
        _lib.hyp_run.argtypes = [C.POINTER(Cfg), C.c_uint64, C.c_uint64, C.c_void_p, C.POINTER(Stats), C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p]
    return _lib


def run(cfg, n, seed=1, split=False):
    hits = np.zeros(cfg.n_theta * cfg.n_phi, np.uint64)
    dz = np.zeros(100, np.uint64)
    rad = np.zeros(64, np.uint64)
    st = Stats()
    ba = np.zeros((18, cfg.n_theta), np.uint64)
    br = np.zeros((40, cfg.n_theta), np.uint64)
    lib().hyp_run(C.byref(cfg), n, seed, hits.ctypes.data, C.byref(st), dz.ctypes.data, rad.ctypes.data,
                  ba.ctypes.data if split else None, br.ctypes.data if split else None)
    if split:
        return hits.reshape(cfg.n_theta, cfg.n_phi), st, dz, rad, ba, br
    return hits.reshape(cfg.n_theta, cfg.n_phi), st, dz, rad


def ref_maps():
    z = np.load(os.path.join(ROOT, "tests", "golden", "reference_maps.npz"))
    idx = json.loads(str(z["index_json"]))
    return {i["name"]: (i, z[i["name"] + "_hits"]) for i in idx}


def compare(ref_hits, n_ref, our_hits, n_our, bands=6):
    """Binomial chi2 of a reference per-position map (n_ref rays per bin) against a trace-once map of n_our rays."""
    ok = ref_hits >= 0
    p = our_hits / n_our
    pr = np.where(ok, ref_hits, 0) / n_ref
    var = p * (1 - p) * (1.0 / n_ref + 1.0 / n_our)
    use = ok & (var > 0) & (p * n_ref >= 5)
    z2 = np.where(use, (pr - p) ** 2 / np.where(var > 0, var, 1), 0.0)
    out = {"chi2_dof": z2.sum() / use.sum(), "dof": int(use.sum()),
           "ratio": (p[ok].sum() / pr[ok].sum())}
    nt = ref_hits.shape[0]
    bw = nt // bands
    out["band_ratio"] = [float(p[b * bw:(b + 1) * bw][ok[b * bw:(b + 1) * bw]].sum() /
                               max(1e-30, pr[b * bw:(b + 1) * bw][ok[b * bw:(b + 1) * bw]].sum())) for b in range(bands)]
    out["band_chi2"] = [float(z2[b * bw:(b + 1) * bw].sum() / max(1, use[b * bw:(b + 1) * bw].sum())) for b in range(bands)]
    return out
