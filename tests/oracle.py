"""ctypes binding of the CPU oracle (oracle/libisx_oracle.so) — TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module.  The product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libisx_oracle.so")


class Config(C.Structure):
    """Same layout as isx_config (include/isx.h) and isxo_config (oracle/isx_oracle.h)."""

    _fields_ = [
        ("struct_size", C.c_uint32), ("reserved0", C.c_uint32),
        ("r_in", C.c_double), ("r_out", C.c_double), ("theta_max_deg", C.c_double),
        ("reflectance", C.c_double), ("roughness_rad", C.c_double), ("box_half", C.c_double),
        ("lambertian", C.c_int32), ("max_points", C.c_int32),
        ("src", C.c_double * 3), ("dir", C.c_double * 3),
        ("n_theta", C.c_int32), ("n_phi", C.c_int32),
        ("det_diameter", C.c_double), ("det_distance", C.c_double), ("exit_port_z", C.c_double),
        ("source_model", C.c_int32), ("surface_model", C.c_int32),
        ("brdf", C.c_double * 3),
        ("hit_line_mode", C.c_int32), ("trace_mode", C.c_int32),
    ]

    def copy(self):
        c = Config()
        C.memmove(C.byref(c), C.byref(self), C.sizeof(Config))
        return c


class Stats(C.Structure):
    _fields_ = [
        ("launched", C.c_uint64), ("exited", C.c_uint64), ("counted_below_z", C.c_uint64),
        ("absorbed", C.c_uint64), ("suspended", C.c_uint64), ("bin_increments", C.c_uint64),
        ("wall_hits", C.c_uint64), ("t_kernel_ms", C.c_double),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(
            os.path.join(ORACLE_DIR, "isx_oracle.c")
        ):
            build()
        L = C.CDLL(ORACLE_SO)
        u64, i32, dbl = C.c_uint64, C.c_int32, C.c_double
        P = C.POINTER
        L.isxo_default_config.argtypes = [P(Config)]
        L.isxo_default_config.restype = None
        L.isxo_philox4x32_10.argtypes = [P(C.c_uint32), P(C.c_uint32), P(C.c_uint32)]
        L.isxo_philox4x32_10.restype = None
        L.isxo_u01.argtypes = [C.c_uint32]
        L.isxo_u01.restype = dbl
        L.isxo_log.argtypes = [dbl]
        L.isxo_log.restype = dbl
        L.isxo_sincos2pi.argtypes = [dbl, P(dbl), P(dbl)]
        L.isxo_sincos2pi.restype = None
        L.isxo_circle_point.argtypes = [dbl, P(dbl), P(dbl)]
        L.isxo_circle_point.restype = None
        L.isxo_sincos.argtypes = [dbl, P(dbl), P(dbl)]
        L.isxo_sincos.restype = None
        L.isxo_detector_table.argtypes = [P(Config), P(dbl)]
        L.isxo_check_intersection.argtypes = [P(dbl), dbl, P(dbl), P(dbl)]
        L.isxo_trace_endstates.argtypes = [P(Config), u64, u64, u64, P(i32), P(i32), P(dbl), P(dbl)]
        L.isxo_fluxmap.argtypes = [P(Config), u64, u64, u64, P(u64), P(Stats), C.c_int]
        L.isxo_disc_sweep.argtypes = [P(Config), P(dbl), i32, dbl, dbl, u64, u64, u64, P(u64), P(Stats), C.c_int]
        L.isxo_disc_sweep_per_position.argtypes = [P(Config), P(dbl), i32, dbl, dbl, u64, u64, u64, P(u64), P(Stats), C.c_int]
        L.isxo_exit_dz_hist.argtypes = [P(Config), u64, u64, u64, i32, P(u64), P(Stats), C.c_int]
        L.isxo_fluxmap_per_position.argtypes = [P(Config), u64, i32, u64, u64, u64, u64, P(u64), P(Stats), C.c_int]
        L.isxo_trace_rays_detector.argtypes = [P(Config), P(dbl), dbl, u64, u64, u64, P(u64), P(Stats)]
        L.isxo_exit_directions.argtypes = [P(Config), u64, u64, u64, u64, P(u64), P(dbl), P(u64)]
        L.isxo_max_threads.restype = C.c_int
        L.isxo_next_boundary.argtypes = [P(Config), P(dbl), P(dbl), C.c_int, P(dbl), P(dbl)]
        L.isxo_surface_normal.argtypes = [P(Config), C.c_int, P(dbl), P(dbl)]
        L.isxo_cosine_emission.argtypes = [P(Config), C.c_int, P(dbl), C.c_uint32, C.c_uint32, P(dbl)]
        _lib = L
    return _lib


def default_config():
    c = Config()
    lib().isxo_default_config(C.byref(c))
    return c


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().isxo_philox4x32_10(c, k, o)
    return list(o)


def sincos2pi(u):
    s, c = C.c_double(), C.c_double()
    lib().isxo_sincos2pi(u, C.byref(s), C.byref(c))
    return s.value, c.value


def circle_point(u):
    c, s = C.c_double(), C.c_double()
    lib().isxo_circle_point(u, C.byref(c), C.byref(s))
    return c.value, s.value


def sincos(x):
    s, c = C.c_double(), C.c_double()
    lib().isxo_sincos(x, C.byref(s), C.byref(c))
    return s.value, c.value


def next_boundary(cfg, p, v, on, with_direction=False):
    """(kind, point) of the next boundary from p along v for a ray sitting on boundary `on` (isxo_next_boundary);
    with_direction: also the direction the step was taken along (the unit vector of v once the ray leaves rule S1')."""
    P3, V3, Q3, W3 = (C.c_double * 3)(*p), (C.c_double * 3)(*v), (C.c_double * 3)(), (C.c_double * 3)()
    kind = lib().isxo_next_boundary(C.byref(cfg), P3, V3, int(on), Q3, W3)
    if with_direction:
        return kind, np.array(Q3[:]), np.array(W3[:])
    return kind, np.array(Q3[:])


def surface_normal(cfg, kind, q):
    Q3, N3 = (C.c_double * 3)(*q), (C.c_double * 3)()
    assert lib().isxo_surface_normal(C.byref(cfg), int(kind), Q3, N3) == 0
    return np.array(N3[:])


def cosine_emission(cfg, kind, q, wa, wb):
    """interact()'s cosine emission from surface point q for the two Philox words of an interaction (r_in s - q, un-normalised,
    on the inner sphere, kind 1; the unit vector of n + s elsewhere)"""
    Q3, W3 = (C.c_double * 3)(*q), (C.c_double * 3)()
    assert lib().isxo_cosine_emission(C.byref(cfg), int(kind), Q3, int(wa), int(wb), W3) == 0
    return np.array(W3[:])


def check_intersection(det, width, last_point, direction):
    D6, L3, V3 = (C.c_double * 6)(*det), (C.c_double * 3)(*last_point), (C.c_double * 3)(*direction)
    return int(lib().isxo_check_intersection(D6, float(width), L3, V3))


def detector_table(cfg):
    out = np.zeros((cfg.n_theta * cfg.n_phi, 6), dtype=np.float64)
    rc = lib().isxo_detector_table(C.byref(cfg), _p(out, C.c_double))
    assert rc == 0, rc
    return out


def trace_endstates(cfg, n, seed, first=0):
    status = np.zeros(n, dtype=np.int32)
    npts = np.zeros(n, dtype=np.int32)
    lp = np.zeros((n, 3), dtype=np.float64)
    d = np.zeros((n, 3), dtype=np.float64)
    rc = lib().isxo_trace_endstates(C.byref(cfg), n, seed, first, _p(status, C.c_int32), _p(npts, C.c_int32),
                                    _p(lp, C.c_double), _p(d, C.c_double))
    assert rc == 0, rc
    return status, npts, lp, d


def fluxmap(cfg, n, seed, first=0, nthreads=0):
    hits = np.zeros(cfg.n_theta * cfg.n_phi, dtype=np.uint64)
    st = Stats()
    rc = lib().isxo_fluxmap(C.byref(cfg), n, seed, first, _p(hits, C.c_uint64), C.byref(st), nthreads)
    assert rc == 0, rc
    return hits.reshape(cfg.n_theta, cfg.n_phi), st


def disc_sweep(cfg, centers_axes, radius, half_thick, n, seed, first=0, nthreads=0):
    ca = np.ascontiguousarray(centers_axes, dtype=np.float64)
    nd = ca.shape[0]
    hits = np.zeros(nd, dtype=np.uint64)
    st = Stats()
    rc = lib().isxo_disc_sweep(C.byref(cfg), _p(ca, C.c_double), nd, radius, half_thick, n, seed, first,
                               _p(hits, C.c_uint64), C.byref(st), nthreads)
    assert rc == 0, rc
    return hits, st


def disc_sweep_per_position(cfg, centers_axes, radius, half_thick, rpp, seed, first=0, nthreads=0):
    ca = np.ascontiguousarray(centers_axes, dtype=np.float64)
    nd = ca.shape[0]
    hits = np.zeros(nd, dtype=np.uint64)
    st = Stats()
    rc = lib().isxo_disc_sweep_per_position(C.byref(cfg), _p(ca, C.c_double), nd, radius, half_thick, rpp, seed, first,
                                            _p(hits, C.c_uint64), C.byref(st), nthreads)
    assert rc == 0, rc
    return hits, st


def exit_dz_hist(cfg, n, seed, nbins=100, first=0, nthreads=0):
    hist = np.zeros(nbins, dtype=np.uint64)
    st = Stats()
    rc = lib().isxo_exit_dz_hist(C.byref(cfg), n, seed, first, nbins, _p(hist, C.c_uint64), C.byref(st), nthreads)
    assert rc == 0, rc
    return hist, st


def fluxmap_per_position(cfg, rays_per_position, seed, fold=1, first_group=0, n_groups=None, first=0, nthreads=0):
    nb = cfg.n_theta * cfg.n_phi
    if n_groups is None:
        n_groups = nb // fold - first_group
    hits = np.zeros(nb, dtype=np.uint64)
    st = Stats()
    rc = lib().isxo_fluxmap_per_position(C.byref(cfg), rays_per_position, fold, first_group, n_groups, seed, first,
                                         _p(hits, C.c_uint64), C.byref(st), nthreads)
    assert rc == 0, rc
    return hits.reshape(cfg.n_theta, cfg.n_phi), st


def trace_rays_detector(cfg, detector, width, n, seed, first=0):
    det = np.ascontiguousarray(detector, dtype=np.float64).reshape(6)
    h = C.c_uint64(0)
    st = Stats()
    rc = lib().isxo_trace_rays_detector(C.byref(cfg), _p(det, C.c_double), width, n, seed, first, C.byref(h), C.byref(st))
    assert rc == 0, rc
    return int(h.value), st


def exit_directions(cfg, n, seed, first=0, capacity=None):
    cap = capacity or n
    ids = np.zeros(cap, dtype=np.uint64)
    d = np.zeros((cap, 3), dtype=np.float64)
    cnt = C.c_uint64(0)
    rc = lib().isxo_exit_directions(C.byref(cfg), n, seed, first, cap, _p(ids, C.c_uint64), _p(d, C.c_double), C.byref(cnt))
    assert rc == 0, rc
    k = min(int(cnt.value), cap)
    return ids[:k], d[:k], int(cnt.value)
