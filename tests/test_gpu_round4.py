"""Round 4 (GPU): the default kernels meet the CPU oracle DIRECTLY at sizes where every code path of the new binning kernels
runs (until now the assist-wave + column-slot path met the oracle only up to 4e5 rays, beyond that
through round 2's kernels), the regression tests of the round-3 advisor findings, and the properties the restructured binning
kernels rest on (second caps without a cut-out, one producer for every kind of line)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = 0x5EED0001
CENSUS = ("launched", "exited", "counted_below_z", "absorbed", "suspended", "bin_increments", "wall_hits")


def _same(a, b):
    for k in CENSUS:
        assert getattr(a, k) == getattr(b, k), k


def _reset(isx):
    for k, v in (("assist", 1), ("assist_block", 0), ("bin_slots", 1), ("bin_cols", 1), ("pipeline", 1), ("ray_sub", 0), ("grid_blocks", 0),
                 ("overlap", 0), ("overlap_trace_streams", 1), ("trace_block", 512), ("trace_blocks_per_cu", 0), ("disc_pipeline", 1),
                 ("bin_mode", 1), ("pipeline_chunk", 1 << 26)):
        isx.set_option(k, v)


def _brdf(mod):
    c = mod.default_config()
    c.source_model = 1; c.brdf[0], c.brdf[1], c.brdf[2] = 0.3, 0.4, 0.6
    c.roughness_rad = 0.5; c.reflectance = 1.0; c.max_points = 10000; c.box_half = 200.0
    return c


def test_default_path_equals_the_oracle_at_2e6_rays(isx, orc):
    """BASELINE configs[1] geometry, every option at its default (assist-wave trace kernel + isx_bin_cols_kernel): histogram and
    census == CPU oracle, bit for bit, at 2e6 rays (8.5e5 exit lines: every class queue, the second-cap pass, the deferred
    list and the unit-end drains all run many times).  The oracle needs ~4 s for this on the GPU box's host cores."""
    _reset(isx)
    n = 2_000_000
    gh, gst = isx.fluxmap(isx.default_config(), n, SEED, 5)
    oh, ost = orc.fluxmap(orc.default_config(), n, SEED, 5)
    assert np.array_equal(gh, oh)
    _same(gst, ost)
    assert gst.counted_below_z > 800_000 and int(gh.sum()) == gst.bin_increments


def test_default_brdf_path_equals_the_oracle_at_5e5_rays(isx, orc):
    """BASELINE configs[2] (nonLambertianFlux.C source model), defaults (assist-wave BRDF trace kernel + isx_bin_slots_kernel with
    its one packed producer for caps, second caps and grazing lines' boxes) == oracle at 5e5 rays (4.7e5 exit lines)."""
    _reset(isx)
    n = 500_000
    gh, gst = isx.fluxmap(_brdf(isx), n, SEED, 9)
    oh, ost = orc.fluxmap(_brdf(orc), n, SEED, 9)
    assert np.array_equal(gh, oh)
    _same(gst, ost)
    # and the column-slot kernel on the same lines (bin_cols = 2: caps as column slots, grazing lines one at a time)
    try:
        isx.set_option("bin_cols", 2)
        ch, cst = isx.fluxmap(_brdf(isx), n, SEED, 9)
        assert np.array_equal(ch, oh)
        _same(cst, ost)
    finally:
        _reset(isx)


@pytest.mark.parametrize("case", range(6))
def test_second_caps_never_count_a_bin_twice(isx, case):
    """isx_bin_cols_kernel no longer cuts the first cap's rows out of the second cap's (that kept the line and the first cap alive
    across the consumer: 60 bytes of scratch per lane and batch); instead a line whose two caps COULD share a bin (caps_may_touch)
    takes the box windows.  Geometries in which many lines have both caps among the detector rows, some of them nearly touching:
    small detector spheres (the piercing points come close), coarse grids (wide row slack), detectors of every size.  culled ==
    brute force through every binning kernel."""
    rng = np.random.default_rng(4100 + case)
    try:
        for _ in range(4):
            c = isx.default_config()
            c.det_distance = float(rng.uniform(6.0, 40.0))            # a small sphere of detector centres right below the port
            c.det_diameter = float(c.det_distance * rng.uniform(0.2, 1.6))
            c.n_theta = int(rng.choice([1, 2, 3, 5, 17, 90, 180])); c.n_phi = int(rng.choice([1, 2, 4, 9, 36, 90]))
            c.theta_max_deg = float(rng.uniform(150.0, 172.0))
            c.exit_port_z = float(rng.uniform(-140.0, -100.0))
            n = 60_000
            _reset(isx)
            isx.set_option("bin_mode", 0)
            ref, rst = isx.fluxmap(c, n, SEED + case, 0)
            isx.set_option("bin_mode", 1)
            for slots, cols in ((1, 1), (1, 0), (0, 0)):
                isx.set_option("bin_slots", slots); isx.set_option("bin_cols", cols)
                h, st = isx.fluxmap(c, n, SEED + case, 0)
                assert np.array_equal(h, ref), (c.det_distance, c.det_diameter, c.n_theta, c.n_phi, slots, cols)
                _same(st, rst)
    finally:
        _reset(isx)


@pytest.mark.parametrize("case", range(4))
def test_caps_about_the_bisector_hold_every_hit(isx, case):
    """Round 4: the cap of an exit line is drawn about the bisector of the two ends of the tube's footprint in the plane of the line
    and O (prep_shared), not about the piercing point.  The lines that differ most are those that pass O at 0.4-0.9 R: the BRDF source
    sends rays through the port in every direction, and detector spheres smaller than the port's reach make most of them such lines.
    culled == brute force through every binning kernel, detectors from 2 % to 60 % of R, grids from coarse to fine."""
    rng = np.random.default_rng(4300 + case)
    try:
        for _ in range(3):
            c = _brdf(isx)
            c.det_distance = float(rng.choice([25.0, 40.0, 70.0, 100.0]))
            c.det_diameter = float(c.det_distance * 2 * rng.choice([0.02, 0.1, 0.2, 0.35, 0.6]))
            c.n_theta = int(rng.choice([7, 45, 90, 180])); c.n_phi = int(rng.choice([6, 24, 90, 120]))
            n = 80_000
            _reset(isx)
            isx.set_option("bin_mode", 0)
            ref, rst = isx.fluxmap(c, n, SEED + case, 0)
            isx.set_option("bin_mode", 1)
            for slots, cols in ((1, 1), (1, 0), (0, 0)):
                isx.set_option("bin_slots", slots); isx.set_option("bin_cols", cols)
                h, st = isx.fluxmap(c, n, SEED + case, 0)
                assert np.array_equal(h, ref), (c.det_distance, c.det_diameter, c.n_theta, c.n_phi, slots, cols)
                _same(st, rst)
            assert int(ref.sum()) > 0
    finally:
        _reset(isx)


def test_disc_sweep_with_more_discs_than_the_pipeline_can_hold(isx):
    """isx_bin_discs_kernel keeps histogram + cluster table + per-wave lists in LDS; above ~16 000 discs that does not
    fit and the call must fall through to the fused SINK_DISC kernel (it returned ISX_ERR_BAD_CONFIG).  20 000 discs,
    disc_pipeline = 1 (falls back) == disc_pipeline = 0."""
    rng = np.random.default_rng(7)
    n_disc = 20_000
    th = rng.uniform(0.0, 0.8, n_disc); ph = rng.uniform(0.0, 2 * np.pi, n_disc)
    axes = np.stack([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), -np.cos(th)], axis=1)
    discs = np.concatenate([200.0 * axes, axes], axis=1).astype(np.float64)
    c = isx.default_config()
    c.r_out = 105.0; c.reflectance = 1.0; c.roughness_rad = 0.0; c.max_points = 10000; c.box_half = 300.0
    c.src[2] = -80.0
    try:
        isx.set_option("disc_pipeline", 1)
        h1, s1 = isx.disc_sweep(c, discs, 5.0, 0.1, 40_000, SEED)
        isx.set_option("disc_pipeline", 0)
        h0, s0 = isx.disc_sweep(c, discs, 5.0, 0.1, 40_000, SEED)
        assert np.array_equal(h1, h0) and int(h0.sum()) > 0
        _same(s1, s0)
    finally:
        _reset(isx)


def test_pipeline_chunk_domain(isx):
    """a launch addresses its rays by 30-bit offsets; pipeline_chunk above the documented 2^26 is refused."""
    lib = isx.load()
    assert lib.isx_set_option(b"pipeline_chunk", 1 << 26) == 0
    assert lib.isx_set_option(b"pipeline_chunk", (1 << 26) + 1) != 0
    assert lib.isx_set_option(b"pipeline_chunk", 1 << 32) != 0
    _reset(isx)


def test_long_idle_assist_wave_is_not_a_failure(isx, orc):
    """ADVICE r03 (bounded waits count polls without progress, not idle time): a port opening of 0.6 degrees and reflectance 1 --
    a ray bounces ~36 000 times before it finds the port or is suspended at 50 000 points, so the assist wave of a workgroup sees
    no hand-over for long stretches while its tracers work.  The call must succeed and equal the oracle."""
    def cfg(mod):
        c = mod.default_config()
        c.theta_max_deg = 179.4; c.reflectance = 1.0
        return c
    _reset(isx)
    n = 3_000
    gh, gst = isx.fluxmap(cfg(isx), n, SEED, 1)
    oh, ost = orc.fluxmap(cfg(orc), n, SEED, 1)
    assert np.array_equal(gh, oh)
    _same(gst, ost)
    assert gst.wall_hits > 20_000 * n


def test_the_last_rays_of_a_launch_change_hands_without_changing_results(isx, orc):
    """Round 4: once a launch's ray queue is dry, a tracer wave that is down to a few rays hands them to the waves of its workgroup
    that stay (resume ring, now with several producers) and leaves.  Launches in which that end game is most of the launch --
    few rays per wave, long-lived rays (reflectance 1: ~144 bounces, the longest of a launch thousands), every workgroup shape,
    one workgroup or many -- must give the oracle's map and census, bit for bit; so must the sinks that share the trace kernels."""
    def cfg(mod, rho, chord=0):
        c = mod.default_config()
        c.reflectance = rho; c.max_points = 10000; c.trace_mode = chord
        return c
    try:
        for rho, n in ((1.0, 30_000), (0.99, 90_000)):
            for chord in (0, 1):
                oh, ost = orc.fluxmap(cfg(orc, rho, chord), n, SEED, 17)
                for block, grid in ((768, 0), (768, 1), (128, 2), (256, 5), (448, 0)):
                    isx.set_option("assist_block", block); isx.set_option("grid_blocks", grid)
                    gh, gst = isx.fluxmap(cfg(isx, rho, chord), n, SEED, 17)
                    assert np.array_equal(gh, oh), (rho, chord, block, grid)
                    _same(gst, ost)
        _reset(isx)
        c, o = _brdf(isx), _brdf(orc)
        gh, gst = isx.fluxmap(c, 40_000, SEED, 3)
        oh, ost = orc.fluxmap(o, 40_000, SEED, 3)
        assert np.array_equal(gh, oh)
        _same(gst, ost)
        gp, gpst = isx.fluxmap_per_position(cfg(isx, 1.0), 3, SEED)
        op, opst = orc.fluxmap_per_position(cfg(orc, 1.0), 3, SEED)
        assert np.array_equal(gp, op)
        _same(gpst, opst)
    finally:
        _reset(isx)
