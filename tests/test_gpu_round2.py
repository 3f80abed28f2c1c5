"""GPU tests added in round 2 (all through the C ABI):

* the refill cursor at the very end of the 64-bit index space;
* ABI v3: a config of another size is refused by the compute entry points too;
* the one-launch per-position physical-disc sweep (integratingSphereDetectorSweep.C:54-77) against the oracle;
* full-size property checks for BASELINE configs[2] (5e7 rays, nonLambertianFlux source model) and configs[3]
  (362 disc positions x 1e7 rays with the reference's disc placement) -- the counterparts of
  test_full_size_properties for configs[1];
* every bin of the reference's seven 8.1e8-ray maps (tests/golden/reference_maps.npz) against 8.1e8 rays of this build,
  with pure binomial sigmas.
"""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
SEED = 0x5EED0001


def _census_equal(a, b):
    for k in ("launched", "exited", "counted_below_z", "absorbed", "suspended", "bin_increments", "wall_hits"):
        assert getattr(a, k) == getattr(b, k), k


def test_refill_cursor_at_the_end_of_the_index_space(isx, orc):
    """A wave whose last refill finds more dead lanes than remaining rays must not run past the end of the launch's range --
    also when that end is 2^64-1.  One block, sub-ranges of 65 rays (one more than the lane count) and of other sizes, with
    the pipeline and with the fused kernel; every one bit-equal to the oracle."""
    isx.set_option("grid_blocks", 1)
    try:
        for pipeline in (1, 0):
            isx.set_option("pipeline", pipeline)
            for n, sub in ((16 * 65, 65), (16 * 64 + 1, 64), (17, 0), (16 * 129 + 5, 129), (3000, 1), (3000, 0)):
                isx.set_option("ray_sub", sub)
                first = (1 << 64) - 1 - n
                gh, gst = isx.fluxmap(isx.default_config(), n, SEED, first)
                oh, ost = orc.fluxmap(orc.default_config(), n, SEED, first)
                assert np.array_equal(gh, oh), (n, sub)
                _census_equal(gst, ost)
                assert gst.launched == n
    finally:
        isx.set_option("grid_blocks", 0); isx.set_option("ray_sub", 0); isx.set_option("pipeline", 1)


def test_two_kernel_pipeline_equals_the_fused_kernel(isx, orc):
    """The headline configuration runs as trace kernel -> exit lines in HBM -> binning kernel ("pipeline" = 1, the default);
    "pipeline" = 0 is the fused kernel.  Same histogram and census from both (and from the oracle), also when the launch is
    cut into several trace/bin pairs, when the grid is tiny, and for other workgroup shapes of the trace kernel."""
    c = isx.default_config()
    try:
        isx.set_option("pipeline", 0)
        ref, rst = isx.fluxmap(c, 3_000_000, SEED, 17)
        assert isx.last_kernel_ms()[0] > 0 and isx.last_kernel_ms()[1] == 0
        small_ref, small_st = isx.fluxmap(c, 30000, SEED)
        isx.set_option("pipeline", 1)
        for chunk in (1 << 26, 1_000_000, 4096 * 100 + 1):
            isx.set_option("pipeline_chunk", chunk)
            h, st = isx.fluxmap(c, 3_000_000, SEED, 17)
            assert np.array_equal(h, ref), chunk
            _census_equal(st, rst)
        single, trace, binning = isx.last_kernel_ms()
        assert single == 0 and trace > 0 and binning > 0 and abs(trace + binning - st.t_kernel_ms) < 1e-3 * st.t_kernel_ms
        isx.set_option("pipeline_chunk", 1 << 26)
        gh, gst = isx.fluxmap(c, 30000, SEED)
        oh, ost = orc.fluxmap(orc.default_config(), 30000, SEED)
        assert np.array_equal(gh, oh) and np.array_equal(gh, small_ref)
        _census_equal(gst, ost)
        _census_equal(gst, small_st)
        for block, bpc in ((64, 0), (128, 5), (256, 8), (1024, 1), (512, 2), (512, 4), (512, 13)):
            isx.set_option("trace_block", block); isx.set_option("trace_blocks_per_cu", bpc)
            h, st = isx.fluxmap(c, 3_000_000, SEED, 17)
            assert np.array_equal(h, ref), (block, bpc)
            _census_equal(st, rst)
        isx.set_option("trace_block", 512); isx.set_option("trace_blocks_per_cu", 0)
        # the work queues: rays per grab, shape of the binning kernel (round 3: the waves of a launch share one ray queue and
        # one queue of exit-line regions; nothing of that may show in a result)
        for sub, bblock, bbpc in ((64, 512, 0), (100, 256, 1), (7777, 1024, 1), (1 << 20, 512, 3)):
            isx.set_option("ray_sub", sub); isx.set_option("bin_block", bblock); isx.set_option("bin_blocks_per_cu", bbpc)
            h, st = isx.fluxmap(c, 3_000_000, SEED, 17)
            assert np.array_equal(h, ref), (sub, bblock, bbpc)
            _census_equal(st, rst)
        isx.set_option("ray_sub", 0); isx.set_option("bin_block", 512); isx.set_option("bin_blocks_per_cu", 0)
        isx.set_option("grid_blocks", 1)
        h1, st1 = isx.fluxmap(c, 50000, SEED, 5)
        isx.set_option("grid_blocks", 0)
        h2, st2 = isx.fluxmap(c, 50000, SEED, 5)
        assert np.array_equal(h1, h2)
        _census_equal(st1, st2)
    finally:
        isx.set_option("grid_blocks", 0)
        isx.set_option("trace_block", 512); isx.set_option("trace_blocks_per_cu", 0)
        isx.set_option("ray_sub", 0); isx.set_option("bin_block", 512); isx.set_option("bin_blocks_per_cu", 0)
        isx.set_option("pipeline_chunk", 1 << 26)
        isx.set_option("pipeline", 1)


def _brdf_cfg(mod):
    c = mod.default_config()
    c.source_model = 1; c.brdf[0], c.brdf[1], c.brdf[2] = 0.3, 0.4, 0.6
    c.roughness_rad = 0.5; c.reflectance = 1.0; c.max_points = 10000; c.box_half = 200.0
    return c


@pytest.mark.parametrize("which", ["chord", "brdf"])
def test_pipeline_of_the_chord_and_brdf_variants(isx, orc, which):
    """The chord mode and the BRDF source model (configs[2]) take the same two-kernel pipeline (their own trace kernel, the
    common binning kernel): same histogram and census as their fused kernels and as the oracle, chunked or not."""
    def cfg(mod):
        if which == "brdf":
            return _brdf_cfg(mod)
        c = mod.default_config(); c.trace_mode = 1
        return c
    n = 400_000 if which == "brdf" else 2_000_000
    try:
        isx.set_option("pipeline", 0)
        ref, rst = isx.fluxmap(cfg(isx), n, SEED, 3)
        assert isx.last_kernel_ms()[1] == 0
        isx.set_option("pipeline", 1)
        for chunk in (1 << 26, 100_003):
            isx.set_option("pipeline_chunk", chunk)
            h, st = isx.fluxmap(cfg(isx), n, SEED, 3)
            assert isx.last_kernel_ms()[1] > 0 and isx.last_kernel_ms()[2] > 0
            assert np.array_equal(h, ref), chunk
            _census_equal(st, rst)
        isx.set_option("pipeline_chunk", 1 << 26)
        m = 20000
        gh, gst = isx.fluxmap(cfg(isx), m, SEED)
        oh, ost = orc.fluxmap(cfg(orc), m, SEED)
        assert np.array_equal(gh, oh)
        _census_equal(gst, ost)
    finally:
        isx.set_option("pipeline_chunk", 1 << 26)
        isx.set_option("pipeline", 1)


def test_config_of_another_abi_is_refused_on_the_device_path(isx):
    c = isx.default_config()
    c.struct_size = C.sizeof(isx.Config) - 8
    with pytest.raises(isx.IsxError) as e:
        isx.fluxmap(c, 1000, 1)
    assert e.value.status == isx.abi.ERR_BAD_CONFIG
    h, st = isx.fluxmap(isx.default_config(), 1000, 1)
    assert st.launched == 1000


def test_exit_log_capacity_rules(isx, orc):
    """isx_exit_directions: a capacity above n_rays is treated as n_rays (no 2^28-record buffer is allocated for
    20 000 rays), the log stays sorted and bit-equal to the oracle's; the record limit applies to min(capacity, n_rays)."""
    c = isx.default_config()
    n = 20000
    ids = np.zeros(n, np.uint64); dirs = np.zeros((n, 3)); cnt = C.c_uint64(0); st = isx.Stats()
    for cap in ((1 << 28) + 1, 1 << 27, n):
        cnt.value = 0
        rc = isx.load().isx_exit_directions(C.byref(c), n, 5, 0, cap, ids.ctypes.data_as(C.POINTER(C.c_uint64)),
                                            dirs.ctypes.data_as(C.POINTER(C.c_double)), C.byref(cnt), C.byref(st))
        assert rc == 0 and 0 < cnt.value <= n and cnt.value == st.counted_below_z
        k = cnt.value
        assert np.all(np.diff(ids[:k].astype(np.int64)) > 0)
    oids, odirs, ocnt = orc.exit_directions(orc.default_config(), n, 5)
    assert ocnt == k and np.array_equal(oids[:k], ids[:k]) and np.array_equal(odirs[:k].view(np.uint64), dirs[:k].view(np.uint64))
    # n_rays itself above the limit with a capacity above it: refused before anything is allocated or traced
    rc = isx.load().isx_exit_directions(C.byref(c), (1 << 28) + 1, 5, 0, (1 << 28) + 1, ids.ctypes.data_as(C.POINTER(C.c_uint64)),
                                        dirs.ctypes.data_as(C.POINTER(C.c_double)), C.byref(cnt), C.byref(st))
    assert rc == isx.abi.ERR_TOO_LARGE


def _disc_cfg(mod):
    """integratingSphereDetectorSweep.C:114-123: shell 100.1-105 cm, port 170 deg, Lambertian, rho = 1, limit 10000."""
    c = mod.default_config()
    c.r_out = 105.0; c.reflectance = 1.0; c.roughness_rad = 0.0; c.max_points = 10000
    c.box_half = 200.0; c.src[2] = -80
    return c


def _disc_positions(dtheta=0.5, theta_max=45.0):
    """rootMacros::detectorDiskPlacement (integratingSphereDetectorSweep.C:145-172)."""
    out = []
    for th in np.arange(-theta_max, theta_max + 1e-9, dtheta):
        for ph in (0.0, 180.0):
            t, p = math.radians(th), math.radians(ph)
            x, y, z = 200 * math.sin(t) * math.cos(p), 200 * math.sin(t) * math.sin(p), -200 * math.cos(t)
            dx, dy, dz = 0 - x, 0 - y, -100 - z
            rot = -math.atan2(math.sqrt(dx * dx + dy * dy), dz)
            out.append([x, y, z, math.sin(rot), 0.0, math.cos(rot)])
    return np.array(out)


def test_disc_sweep_per_position_bit_exact(isx, orc):
    ca = _disc_positions(dtheta=5.0)            # 19 x 2 positions
    rpp = 20000
    gh, gst = isx.disc_sweep_per_position(_disc_cfg(isx), ca, 5.0, 0.1, rpp, 99, 1000)
    oh, ost = orc.disc_sweep_per_position(_disc_cfg(orc), ca, 5.0, 0.1, rpp, 99, 1000)
    assert np.array_equal(gh, oh) and gh.sum() > 0
    _census_equal(gst, ost)
    assert gst.launched == rpp * len(ca)
    # position k alone, on its own rays, is the k-th entry of the one-launch sweep
    for k in (0, 7, len(ca) - 1):
        one, _ = isx.disc_sweep(_disc_cfg(isx), ca[k:k + 1], 5.0, 0.1, rpp, 99, 1000 + k * rpp)
        assert one[0] == gh[k]


def test_full_size_disc_sweep_properties(isx, orc):
    """BASELINE configs[3]: the integratingSphereDetectorSweep.C sweep, 181 x 2 positions, 1e7 rays per position
    (3.62e9 rays, one launch)."""
    ca = _disc_positions()
    assert len(ca) == 362
    rpp = 10_000_000
    c = _disc_cfg(isx)
    h, st = isx.disc_sweep_per_position(c, ca, 5.0, 0.1, rpp, 7)
    n = rpp * len(ca)
    assert st.launched == n == st.exited + st.absorbed + st.suspended
    assert st.absorbed == 0                      # rho = 1
    assert st.bin_increments == int(h.sum())
    # a sub-sweep over the first 40 positions is the same numbers (position k owns rays [k*rpp, (k+1)*rpp))
    h40, _ = isx.disc_sweep_per_position(c, ca[:40], 5.0, 0.1, rpp, 7)
    assert np.array_equal(h[:40], h40)
    # the reference's own profile (detector_sweep.txt, 1000 rays per point, phi-mean over 360 azimuths): 0.00275 on axis,
    # 1.2-1.7e-4 at +-45 deg; this sweep has the macro's two azimuths only (the disc faces the port at phi = 0 only)
    frac = h / rpp
    i0 = len(ca) // 2
    assert 0.0024 < frac[i0 - 1:i0 + 1].mean() < 0.0031
    assert 1e-4 < frac[:2].mean() < 4e-4 and 1e-4 < frac[-2:].mean() < 4e-4
    # mirror symmetry of the sweep: theta -> -theta swaps the roles of phi = 0 and phi = 180
    # (the same physical disc, traced with different rays: Poisson-compatible counts)
    hh = h.reshape(181, 2).astype(np.float64)
    z = (hh[:, 0] - hh[::-1, 1]) / np.sqrt(np.maximum(hh[:, 0] + hh[::-1, 1], 1.0))
    assert np.abs(z).max() < 5.0 and abs(z.mean()) < 0.5
    # == oracle at a size it finishes in seconds
    sub = ca[::40]
    gh, gst = isx.disc_sweep_per_position(c, sub, 5.0, 0.1, 20000, 7)
    oh, ost = orc.disc_sweep_per_position(_disc_cfg(orc), sub, 5.0, 0.1, 20000, 7)
    assert np.array_equal(gh, oh)
    _census_equal(gst, ost)


def _brdf_cfg(mod):
    """BASELINE configs[2]: nonLambertianFlux.C source model (BRDF re-scatter, :147-208,253-268) and surface set-up
    (:213-226: rho = 1, limit 10000, box 200 cm, roughness .5) on the 180x90 / 40 cm / src z = -75 grid."""
    c = mod.default_config()
    c.source_model = 1; c.brdf[0], c.brdf[1], c.brdf[2] = 0.3, 0.4, 0.6
    c.roughness_rad = 0.5; c.reflectance = 1.0; c.max_points = 10000; c.box_half = 200.0
    return c


def test_full_size_brdf_source_properties(isx, orc):
    c = _brdf_cfg(isx)
    n = 50_000_000
    h, st = isx.fluxmap(c, n, SEED)
    assert st.launched == n == st.exited + st.absorbed + st.suspended
    assert st.absorbed == 0 and st.bin_increments == int(h.sum())
    assert st.counted_below_z <= st.exited
    # additivity over a split of the same stream
    h1, _ = isx.fluxmap(c, 5_000_000, SEED)
    h2, _ = isx.fluxmap(c, n - 5_000_000, SEED, 5_000_000)
    assert np.array_equal(h, h1 + h2)
    # the culled binning (whole-row walks for the grazing lines of this source) against the brute-force one
    isx.set_option("bin_mode", 0)
    try:
        hb, _ = isx.fluxmap(c, 300_000, SEED, 123)
    finally:
        isx.set_option("bin_mode", 1)
    hc, _ = isx.fluxmap(c, 300_000, SEED, 123)
    assert np.array_equal(hb, hc)
    # == oracle
    gh, gst = isx.fluxmap(c, 200_000, SEED, 77)
    oh, ost = orc.fluxmap(_brdf_cfg(orc), 200_000, SEED, 77)
    assert np.array_equal(gh, oh)
    _census_equal(gst, ost)


# ------------------------------------------------------------------------------------------------ reference maps
def _ref_maps():
    z = np.load(os.path.join(ROOT, "tests", "golden", "reference_maps.npz"))
    idx = json.loads(str(z["index_json"]))
    return [(i, z[i["name"] + "_hits"]) for i in idx]


def _chi2(ref_hits, n_ref, our_hits, n_our):
    """binomial chi2 per bin of two independent estimates of the same hit probabilities (pure sigmas, no floor)"""
    ok = ref_hits >= 0
    p = (np.where(ok, ref_hits, 0) + our_hits) / (n_ref + n_our)       # pooled estimate under the null hypothesis
    var = p * (1 - p) * (1.0 / n_ref + 1.0 / n_our)
    use = ok & (p * min(n_ref, n_our) >= 5)
    z2 = np.where(use, (np.where(ok, ref_hits, 0) / n_ref - our_hits / n_our) ** 2 / np.where(var > 0, var, 1), 0.0)
    return float(z2.sum() / use.sum()), int(use.sum())


# What round 2 established about the residual against the reference's data (profiles/r02_parity_scan.md): with the
# model of this build every one of the seven maps has chi2/dof <= 1.04 over ~16 000 bins, i.e. bin by bin the maps are
# indistinguishable at the resolution of one 50 000-ray bin (3.6 %); summed over a map the total is 0.3-1.0 % low, a
# smooth theta-only pattern (+1 % on axis, -1.5 % at 25-35 deg) that none of the inferred ROBAST behaviours removes and
# that the reference's own single-threaded exit log does not show.  The bounds below are SYMMETRIC about the reference's value
# (round-2 advice): they hold the measured state (totals 0.3-1.0 % low) and would also hold a model that closed the gap; the
# one marker of the gap itself is test_totals_of_the_reference_maps_within_0p15_percent (xfail, strict).
CHI2_MAX = 1.08
TOTAL_WINDOW = (0.9870, 1.0130)


@pytest.mark.parametrize("k", range(7))
def test_every_bin_of_the_reference_maps(isx, k):
    maps = [m for m in _ref_maps() if m[0]["kind"] == "per_position" and m[0]["complete"]]
    info, ref = maps[k]
    c = isx.default_config()
    c.theta_max_deg = info["port_deg"]
    for a in range(3):
        c.src[a] = info["source_position"][a]
        c.dir[a] = info["source_direction"][a]
    assert (info["reflectance"], info["roughness"], info["r_in"], info["r_out"]) == (c.reflectance, c.roughness_rad, c.r_in, c.r_out)
    n = info["rays_per_position"]
    h, st = isx.fluxmap_per_position(c, n, 1234 + k)           # the reference's own procedure: 50 000 fresh rays per position
    chi2, dof = _chi2(ref.astype(np.int64), n, h.astype(np.int64), n)
    ratio = h.sum() / ref.sum()
    print(f"{info['name']} port {info['port_deg']} dir {info['source_direction']}: chi2/dof {chi2:.4f} ({dof} bins), total ratio {ratio:.5f}")
    assert dof > 14000
    assert chi2 < CHI2_MAX, (info["name"], chi2)
    assert TOTAL_WINDOW[0] < ratio < TOTAL_WINDOW[1], (info["name"], ratio)


def test_reference_maps_differ_from_this_build_by_a_smooth_function_of_theta_only(isx):
    """The sharpest statement the reference's data allows (profiles/r02_parity_scan.md section 6): against this build's bin
    probabilities known 'exactly' (2e9-ray trace-once map: every ray tested against every bin), each of the seven 8.1e8-ray maps
    of the reference has chi2/dof 1.03-1.08; after ONE smooth factor of the detector angle theta (6th-order polynomial, 7
    parameters, |f - 1| < 3 %) it is 1.00 +- 0.011 -- the bins scatter binomially, nothing depends on phi, no overdispersion.
    A change of the model that moved the maps in any other way (a phi structure, a rim effect, a different port) fails here."""
    N = 2_000_000_000
    worst = []
    for k, (info, ref) in enumerate(m for m in _ref_maps() if m[0]["kind"] == "per_position" and m[0]["complete"]):
        c = isx.default_config()
        c.theta_max_deg = info["port_deg"]
        for a in range(3):
            c.src[a] = info["source_position"][a]
            c.dir[a] = info["source_direction"][a]
        n = info["rays_per_position"]
        h = np.zeros(16200, np.int64)
        for part in range(2):
            hh, _ = isx.fluxmap(c, N // 2, 4242 + k, part * (N // 2))
            h += hh.astype(np.int64).reshape(-1)
        p = (h / N).reshape(180, 90)
        r = ref.astype(np.int64).reshape(180, 90)
        use = p * n >= 5
        var = n * p * (1 - p) * (1 + n / N)
        row = np.array([r[i][use[i]].sum() / (n * p[i][use[i]]).sum() if use[i].any() else 1.0 for i in range(180)])
        w = np.array([(n * p[i][use[i]]).sum() for i in range(180)])
        th = (np.arange(180) + 0.5) / 180.0
        ok = w > 0
        f = np.polyval(np.polyfit(th[ok], row[ok], 6, w=np.sqrt(w[ok])), th)
        chi2_0 = float((((r - n * p) ** 2 / np.where(var > 0, var, 1))[use]).sum() / use.sum())
        chi2_f = float((((r - n * p * f[:, None]) ** 2 / np.where(var > 0, var * f[:, None], 1))[use]).sum() / (use.sum() - 7))
        print(f"{info['name']}: chi2/dof {chi2_0:.4f} -> {chi2_f:.4f} after f(theta), f in [{f[w > 0.05 * w.max()].min():.4f}, {f[w > 0.05 * w.max()].max():.4f}]")
        assert chi2_0 < 1.12, (info["name"], chi2_0)                   # the residual (1.03-1.08 today) is bounded ...
        assert 0.955 < chi2_f < 1.045, (info["name"], chi2_f)          # ... and it is a smooth factor of theta, nothing else (4 sigma)
        big = w > 0.05 * w.max()                                       # (rows that carry hits: the last degrees before 90 carry almost none)
        assert 0.96 < f[big].min() and f[big].max() < 1.04, (info["name"], f[big].min(), f[big].max())
        worst.append(chi2_f)
    assert abs(np.mean(worst) - 1.0) < 0.02, worst                     # seven maps together: 1.00 +- 0.0043


@pytest.mark.xfail(strict=True, reason="known residual vs the reference's 8.1e8-ray maps: totals 0.3-1.0 % low "
                                       "(profiles/r02_parity_scan.md); the bar is 0.15 %")
def test_totals_of_the_reference_maps_within_0p15_percent(isx):
    worst = 0.0
    for k, (info, ref) in enumerate(m for m in _ref_maps() if m[0]["kind"] == "per_position" and m[0]["complete"]):
        c = isx.default_config()
        c.theta_max_deg = info["port_deg"]
        for a in range(3):
            c.dir[a] = info["source_direction"][a]
        h, _ = isx.fluxmap_per_position(c, info["rays_per_position"], 99 + k)
        worst = max(worst, abs(h.sum() / ref.sum() - 1))
    assert worst < 0.0015, worst


# ------------------------------------------------------------------------------------------------ bench.py, RCCL path
def test_bench_distributed_path_in_a_fresh_process(isx):
    """bench.py under ISX_FORCE_DIST=1 (torch.distributed "nccl" = RCCL with one rank, the same code path the driver's
    N = 2/4/8 launches take), in a child process of its own: the histogram must stay on the device for the all-reduce
    (one HIP runtime in the process), RCCL must report the world it was given, and the all-reduced histogram of the last
    step must be the one isx.fluxmap gives for the same ray indices."""
    import subprocess
    import sys
    rays, steps, warmup = 2_000_000, 2, 1
    env = dict(os.environ, ISX_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", LOCAL_RANK="0",
               WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", str(steps), "--warmup",
                        str(warmup), "--rays", str(rays), "--cpu-rays", "0"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout          # the contract: ONE JSON line on stdout
    out = json.loads(lines[0])
    cfg = out["config"]
    assert cfg["reduce_path"] == "device"
    assert cfg["rccl_world_size"] == 1 and cfg["torch_backend"] == "nccl"
    assert len(cfg["hip_runtime_images"]) == 1
    assert out["n_gpus"] == 1 and out["steps"] == steps and out["roofline"]["bound"] == "valu_issue"
    # last timed step = step index warmup + steps - 1 of the schedule
    first, count = isx.step_slice(warmup + steps - 1, 0, 1, rays)
    h, st = isx.fluxmap(isx.default_config(), count, SEED, first)
    assert out["hist_sum_last_step"] == int(h.sum())
    assert out["census_last_step"]["counted_below_z"] == st.counted_below_z


def test_bench_two_ranks_share_the_gpu_over_gloo(isx):
    """The driver's N = 2 launch line (`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...`) on the one GPU
    of this box, with gloo instead of RCCL (ISX_BENCH_BACKEND: RCCL refuses two ranks on one device): both ranks trace their
    slices of every step, ONE all-reduce sums the histograms, rank 0 prints ONE line -- whose last-step histogram sum and
    census must be those of the two slices traced in this process."""
    import subprocess
    import sys
    rays, steps, warmup = 1_500_000, 2, 1
    env = dict(os.environ, ISX_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", ISX_BENCH_CONFIGS3="1", ISX_BENCH_CONFIGS3_RAYS="300000")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "ISX_FORCE_DIST"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", str(steps),
                        "--warmup", str(warmup), "--rays", str(rays), "--cpu-rays", "0"], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == steps and out["scaling"] == "weak"
    assert out["config"]["rccl_world_size"] == 2 and out["config"]["torch_backend"] == "gloo"
    s_last = warmup + steps - 1
    total, counted0 = 0, 0
    for rank in range(2):
        first, count = isx.step_slice(s_last, rank, 2, rays)
        h, st = isx.fluxmap(isx.default_config(), count, SEED, first)
        total += int(h.sum())
        if rank == 0:
            counted0 = st.counted_below_z
    assert out["hist_sum_last_step"] == total                           # the all-reduced histogram
    assert out["census_last_step"]["counted_below_z"] == counted0       # (rank 0's own census)
    # whole-job throughput: both ranks' rays over the slowest rank's time
    assert abs(out["value"] - 2 * rays / (out["ms_per_step"] * 1e3)) < 1e-6 * out["value"]
    # what a bad scaling point would be diagnosed from: per-rank kernel time, its halves and the one all-reduce, [min, max] over ranks
    pr = out["per_rank_ms_min_max"]
    for key in ("kernel_ms", "trace_ms", "bin_ms", "allreduce_ms"):
        lo, hi = pr[key]
        assert 0 < lo <= hi, (key, pr)
    assert pr["kernel_ms"][1] < out["ms_per_step"] and out["configs4"] is None
    # the configs[3] leg (the driver's N = 8 launch runs it by default; here ISX_BENCH_CONFIGS3=1 rehearses it with two ranks):
    # 362 disc positions share the rays of a step, ray-sharded, ONE all-reduce of the counts -- the last step's summed counts
    # must be those of its two slices swept in this process
    c3 = out["configs3"]
    assert c3 is not None and c3["n_discs"] == 362 and c3["rays_per_gpu_per_step"] == 300000 and c3["value"] > 0
    import math
    discs = []
    for th in np.arange(-45.0, 45.0 + 1e-9, 0.5):
        for ph in (0.0, 180.0):
            t_, p_ = math.radians(th), math.radians(ph)
            x, y, z = 200 * math.sin(t_) * math.cos(p_), 200 * math.sin(t_) * math.sin(p_), -200 * math.cos(t_)
            rot = -math.atan2(math.sqrt(x * x + y * y), -100 - z)
            discs.append([x, y, z, math.sin(rot), 0.0, math.cos(rot)])
    cfg3 = isx.default_config()
    cfg3.r_out = 105.0; cfg3.reflectance = 1.0; cfg3.roughness_rad = 0.0; cfg3.max_points = 10000; cfg3.box_half = 200.0
    cfg3.src[2] = -80.0
    want = 0
    for rank in range(2):
        first, count = isx.step_slice(steps, rank, 2, 300000)   # (the leg's last timed step has index `steps`: one warm-up step before it)
        h3, _ = isx.disc_sweep(cfg3, np.array(discs), 5.0, 0.1, count, 7, first)
        want += int(h3.sum())
    assert c3["disc_hits_last_step"] == want and want > 0


def test_random_configurations_equal_the_oracle(isx, orc):
    """36 random configurations over every field the path reads and every model switch (pencil / BRDF source, explicit / chord
    trace, Lambertian / rough specular / cos^2-lobe surface, last-segment / origin-compat hit line; wall radii, port angle, port
    plane, box, grid, detector size up to 1.2 sphere radii, ray limit, random first ray index): flux map and census bit-equal to
    the oracle's, and per-position maps (both folds) + the exit-dz histogram for the configurations that have them.
    (`tools/soak_oracle.py` is the long version: 180 configurations, clean.)"""
    rng = np.random.default_rng(20261004)
    census = ("launched", "exited", "counted_below_z", "absorbed", "suspended", "bin_increments", "wall_hits")
    for k in range(36):
        v = dict(theta_max_deg=float(rng.uniform(150, 178)), reflectance=float(rng.choice([0.9, 0.97, 0.99, 1.0])),
                 max_points=int(rng.choice([50, 400, 3000])), box_half=float(rng.choice([200.0, 300.0])),
                 n_theta=int(rng.integers(1, 120)), n_phi=2 * int(rng.integers(1, 70)),
                 det_distance=float(rng.choice([30.0, 100.0, 180.0])), exit_port_z=float(rng.choice([-100.0, -120.0, -99.0])),
                 r_in=float(rng.choice([100.1, 60.0])))
        v["r_out"] = v["r_in"] + float(rng.choice([0.9, 5.0]))
        v["det_diameter"] = float(v["det_distance"] * 2 * rng.choice([0.02, 0.2, 0.5, 0.8, 1.2]))
        mode = k % 6
        if mode == 1:
            v["source_model"] = 1
        elif mode == 2:
            v["trace_mode"] = 1
        elif mode == 3:
            v.update(lambertian=0, roughness_rad=float(rng.choice([0.05, 0.3])), reflectance=0.9)
        elif mode == 4:
            v["surface_model"] = 1
        elif mode == 5:
            v["hit_line_mode"] = 1
        src = [float(rng.uniform(-0.6, 0.6) * v["r_in"]), float(rng.uniform(-0.3, 0.3) * v["r_in"]), float(rng.uniform(-0.8, 0.4) * v["r_in"])]
        dr = [float(rng.uniform(1, 6)), float(rng.uniform(-3, 3)), float(rng.uniform(-2, 2))]
        ci, co = isx.default_config(), orc.default_config()
        for c in (ci, co):
            for f, x in v.items():
                setattr(c, f, x)
            for a in range(3):
                c.src[a] = src[a]
                c.dir[a] = dr[a]
        n, first = 20000, int(rng.integers(0, 1 << 48))
        gh, gst = isx.fluxmap(ci, n, 100 + k, first)
        oh, ost = orc.fluxmap(co, n, 100 + k, first)
        assert np.array_equal(gh, oh), (k, v)
        for f in census:
            assert getattr(gst, f) == getattr(ost, f), (k, f, v)
        if mode in (0, 2):
            for fold in (1, 2):
                gp, _ = isx.fluxmap_per_position(ci, 7, 300 + k, fold)
                op, _ = orc.fluxmap_per_position(co, 7, 300 + k, fold)
                assert np.array_equal(gp, op), (k, fold, v)
            gd, _ = isx.exit_dz_hist(ci, n, 400 + k)
            od, _ = orc.exit_dz_hist(co, n, 400 + k)
            assert np.array_equal(gd, od), (k, v)
