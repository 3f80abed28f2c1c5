"""ctypes binding of libisx.so (include/isx.h) — the only way Python reaches the GPU path.

There is no Python/NumPy compute fallback: if libisx.so is missing or no HIP device is
present, every compute call raises IsxError.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("ISX_LIB_PATH") or os.path.join(CSRC, "libisx.so")  # ISX_LIB_PATH: tuning variants

OK = 0
ERR_NO_DEVICE = -1
ERR_BAD_CONFIG = -2
ERR_BAD_ARG = -3
ERR_HIP = -4
ERR_NOT_INIT = -5
ERR_TOO_LARGE = -6

SOURCE_PENCIL = 0
SOURCE_BRDF = 1

RAY_EXITED, RAY_ABSORBED, RAY_SUSPENDED = 1, 2, 3

# every symbol include/isx.h declares (tests check the .so exports exactly these)
EXPORTS = [
    "isx_default_config", "isx_init", "isx_shutdown", "isx_strerror", "isx_last_hip_error", "isx_abi_version", "isx_stream_version",
    "isx_device_info", "isx_fluxmap", "isx_fluxmap_device", "isx_sync", "isx_take_stats", "isx_stream",
    "isx_set_option", "isx_mathprobe", "isx_trace_endstates", "isx_disc_sweep", "isx_detector_table",
    "isx_exit_dz_hist", "isx_fluxmap_per_position", "isx_trace_rays_detector", "isx_exit_directions",
    "isx_fluxmap_series", "isx_disc_sweep_per_position", "isx_last_kernel_ms",
]


class IsxError(RuntimeError):
    def __init__(self, status, where=""):
        self.status = status
        msg = _lib.isx_strerror(status).decode() if _lib is not None else str(status)
        super().__init__(f"libisx {where}: {msg} (status {status})")


class Config(C.Structure):
    """isx_config (include/isx.h)."""

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
    """isx_stats (include/isx.h)."""

    _fields_ = [
        ("launched", C.c_uint64), ("exited", C.c_uint64), ("counted_below_z", C.c_uint64),
        ("absorbed", C.c_uint64), ("suspended", C.c_uint64), ("bin_increments", C.c_uint64),
        ("wall_hits", C.c_uint64), ("t_kernel_ms", C.c_double),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


_lib = None


def load():
    """dlopen libisx.so; raises if it was not built (run __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not built: run `make -C {CSRC}` (or __graft_entry__.build()); "
                          "there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    u64, i32, i64, dbl, P = C.c_uint64, C.c_int32, C.c_int64, C.c_double, C.POINTER
    L.isx_default_config.argtypes = [P(Config)]
    L.isx_default_config.restype = None
    L.isx_init.argtypes = [C.c_int]
    L.isx_shutdown.restype = None
    L.isx_strerror.argtypes = [C.c_int]
    L.isx_strerror.restype = C.c_char_p
    L.isx_device_info.argtypes = [C.c_char_p, C.c_int]
    L.isx_fluxmap.argtypes = [P(Config), u64, u64, u64, P(u64), P(Stats)]
    L.isx_fluxmap_device.argtypes = [P(Config), u64, u64, u64, C.c_void_p]
    L.isx_take_stats.argtypes = [P(Stats)]
    L.isx_stream.restype = C.c_void_p
    L.isx_set_option.argtypes = [C.c_char_p, i64]
    L.isx_mathprobe.argtypes = [C.c_int, P(dbl), P(dbl), P(dbl), P(dbl), i32]
    L.isx_trace_endstates.argtypes = [P(Config), u64, u64, u64, P(i32), P(i32), P(dbl), P(dbl)]
    L.isx_disc_sweep.argtypes = [P(Config), P(dbl), i32, dbl, dbl, u64, u64, u64, P(u64), P(Stats)]
    L.isx_disc_sweep_per_position.argtypes = [P(Config), P(dbl), i32, dbl, dbl, u64, u64, u64, P(u64), P(Stats)]
    L.isx_last_kernel_ms.argtypes = [P(dbl), P(dbl), P(dbl)]
    L.isx_detector_table.argtypes = [P(Config), P(dbl)]
    L.isx_exit_dz_hist.argtypes = [P(Config), u64, u64, u64, i32, P(u64), P(Stats)]
    L.isx_fluxmap_per_position.argtypes = [P(Config), u64, i32, u64, u64, u64, u64, P(u64), P(Stats)]
    L.isx_trace_rays_detector.argtypes = [P(Config), P(dbl), dbl, u64, u64, u64, P(u64), P(Stats)]
    L.isx_exit_directions.argtypes = [P(Config), u64, u64, u64, u64, P(u64), P(dbl), P(u64), P(Stats)]
    L.isx_fluxmap_series.argtypes = [P(Config), i32, u64, u64, u64, P(u64), P(Stats)]
    _lib = L
    return L


def _chk(rc, where):
    if rc != OK:
        raise IsxError(rc, where)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def default_config():
    c = Config()
    load().isx_default_config(C.byref(c))
    return c


def init(device=0):
    _chk(load().isx_init(int(device)), "isx_init")


def shutdown():
    load().isx_shutdown()


def device_info():
    buf = C.create_string_buffer(256)
    cu = load().isx_device_info(buf, 256)
    if cu < 0:
        raise IsxError(cu, "isx_device_info")
    return buf.value.decode(), cu


def set_option(key, value):
    _chk(load().isx_set_option(key.encode(), int(value)), f"isx_set_option({key})")


def fluxmap(cfg, n_rays, seed, first_ray=0):
    """-> (hits[n_theta, n_phi] uint64, Stats).  Host-buffer form of the ABI."""
    hits = np.zeros(cfg.n_theta * cfg.n_phi, dtype=np.uint64)
    st = Stats()
    _chk(load().isx_fluxmap(C.byref(cfg), int(n_rays), int(seed), int(first_ray), _p(hits, C.c_uint64), C.byref(st)),
         "isx_fluxmap")
    return hits.reshape(cfg.n_theta, cfg.n_phi), st


def fluxmap_device(cfg, n_rays, seed, first_ray, d_hits_ptr):
    """Enqueue on the library stream, accumulating into device memory at d_hits_ptr."""
    _chk(load().isx_fluxmap_device(C.byref(cfg), int(n_rays), int(seed), int(first_ray), C.c_void_p(int(d_hits_ptr))),
         "isx_fluxmap_device")


def sync():
    _chk(load().isx_sync(), "isx_sync")


def take_stats():
    st = Stats()
    _chk(load().isx_take_stats(C.byref(st)), "isx_take_stats")
    return st


def trace_endstates(cfg, n, seed, first_ray=0):
    status = np.zeros(n, dtype=np.int32)
    npts = np.zeros(n, dtype=np.int32)
    lp = np.zeros((n, 3), dtype=np.float64)
    d = np.zeros((n, 3), dtype=np.float64)
    _chk(load().isx_trace_endstates(C.byref(cfg), int(n), int(seed), int(first_ray), _p(status, C.c_int32),
                                    _p(npts, C.c_int32), _p(lp, C.c_double), _p(d, C.c_double)), "isx_trace_endstates")
    return status, npts, lp, d


def disc_sweep(cfg, centers_axes, radius, half_thick, n_rays, seed, first_ray=0):
    ca = np.ascontiguousarray(centers_axes, dtype=np.float64)
    nd = ca.shape[0]
    hits = np.zeros(nd, dtype=np.uint64)
    st = Stats()
    _chk(load().isx_disc_sweep(C.byref(cfg), _p(ca, C.c_double), nd, float(radius), float(half_thick), int(n_rays),
                               int(seed), int(first_ray), _p(hits, C.c_uint64), C.byref(st)), "isx_disc_sweep")
    return hits, st


def last_kernel_ms():
    """(single, trace, bin) HIP-event milliseconds of the launches collected by the last blocking call / take_stats()."""
    a, b, c = C.c_double(), C.c_double(), C.c_double()
    _chk(load().isx_last_kernel_ms(C.byref(a), C.byref(b), C.byref(c)), "isx_last_kernel_ms")
    return a.value, b.value, c.value


def disc_sweep_per_position(cfg, centers_axes, radius, half_thick, rays_per_position, seed, first_ray=0):
    """One launch for the reference's per-position disc loop: disc k sees only its own rays."""
    ca = np.ascontiguousarray(centers_axes, dtype=np.float64)
    nd = ca.shape[0]
    hits = np.zeros(nd, dtype=np.uint64)
    st = Stats()
    _chk(load().isx_disc_sweep_per_position(C.byref(cfg), _p(ca, C.c_double), nd, float(radius), float(half_thick),
                                            int(rays_per_position), int(seed), int(first_ray), _p(hits, C.c_uint64),
                                            C.byref(st)), "isx_disc_sweep_per_position")
    return hits, st


def exit_dz_hist(cfg, n_rays, seed, nbins=100, first_ray=0):
    hist = np.zeros(nbins, dtype=np.uint64)
    st = Stats()
    _chk(load().isx_exit_dz_hist(C.byref(cfg), int(n_rays), int(seed), int(first_ray), int(nbins),
                                 _p(hist, C.c_uint64), C.byref(st)), "isx_exit_dz_hist")
    return hist, st


def fluxmap_per_position(cfg, rays_per_position, seed, fold=1, first_group=0, n_groups=None, first_ray=0):
    nb = cfg.n_theta * cfg.n_phi
    if n_groups is None:
        n_groups = nb // fold - first_group
    hits = np.zeros(nb, dtype=np.uint64)
    st = Stats()
    _chk(load().isx_fluxmap_per_position(C.byref(cfg), int(rays_per_position), int(fold), int(first_group), int(n_groups),
                                         int(seed), int(first_ray), _p(hits, C.c_uint64), C.byref(st)),
         "isx_fluxmap_per_position")
    return hits.reshape(cfg.n_theta, cfg.n_phi), st


def trace_rays_detector(cfg, detector, width, n_rays, seed, first_ray=0):
    det = np.ascontiguousarray(detector, dtype=np.float64).reshape(6)
    h = C.c_uint64(0)
    st = Stats()
    _chk(load().isx_trace_rays_detector(C.byref(cfg), _p(det, C.c_double), float(width), int(n_rays), int(seed),
                                        int(first_ray), C.byref(h), C.byref(st)), "isx_trace_rays_detector")
    return int(h.value), st


def exit_directions(cfg, n_rays, seed, first_ray=0, capacity=None):
    """-> (ray_ids[k], directions[k,3], total_count, Stats); k = min(total_count, capacity)."""
    cap = int(capacity or n_rays)
    ids = np.zeros(cap, dtype=np.uint64)
    d = np.zeros((cap, 3), dtype=np.float64)
    cnt = C.c_uint64(0)
    st = Stats()
    _chk(load().isx_exit_directions(C.byref(cfg), int(n_rays), int(seed), int(first_ray), cap, _p(ids, C.c_uint64),
                                    _p(d, C.c_double), C.byref(cnt), C.byref(st)), "isx_exit_directions")
    k = min(int(cnt.value), cap)
    return ids[:k], d[:k], int(cnt.value), st


def fluxmap_series(cfgs, n_rays, seed, first_ray=0):
    """cfgs: list of Config sharing one detector grid -> (hits[n_cfg, n_theta, n_phi], [Stats])."""
    n = len(cfgs)
    arr = (Config * n)(*cfgs)
    nb = cfgs[0].n_theta * cfgs[0].n_phi
    hits = np.zeros(n * nb, dtype=np.uint64)
    st = (Stats * n)()
    _chk(load().isx_fluxmap_series(arr, n, int(n_rays), int(seed), int(first_ray), _p(hits, C.c_uint64), st),
         "isx_fluxmap_series")
    return hits.reshape(n, cfgs[0].n_theta, cfgs[0].n_phi), list(st)


def detector_table(cfg):
    out = np.zeros((cfg.n_theta * cfg.n_phi, 6), dtype=np.float64)
    _chk(load().isx_detector_table(C.byref(cfg), _p(out, C.c_double)), "isx_detector_table")
    return out


def mathprobe(op, a, b=None, c=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = a if b is None else np.ascontiguousarray(b, dtype=np.float64)
    c = a if c is None else np.ascontiguousarray(c, dtype=np.float64)
    out = np.zeros_like(a)
    _chk(load().isx_mathprobe(int(op), _p(a, C.c_double), _p(b, C.c_double), _p(c, C.c_double), _p(out, C.c_double),
                              a.size), "isx_mathprobe")
    return out
