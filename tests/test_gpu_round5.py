"""Round 5 (GPU): the border models that are not the lean Lambertian one -- the cos^2 lobe of "nonLambertianFlux copy.C":31-70
(the de-facto CustomMirror), ROBAST's rough-specular border, the origin-compat hit line of fluxAtObserverFast.C:1181-1201 -- on the
assist-wave pipeline (until round 5 they ran on round 1's fused kernel) == the CPU oracle, bit for bit, at 5e5 rays; small calls;
the bounded-wait failure path."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = 0x5EED0001
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CENSUS = ("launched", "exited", "counted_below_z", "absorbed", "suspended", "bin_increments", "wall_hits")


def _same(a, b):
    for k in CENSUS:
        assert getattr(a, k) == getattr(b, k), k


def _reset(isx):
    for k, v in (("assist", 1), ("assist_block", 0), ("bin_slots", 1), ("bin_cols", 1), ("pipeline", 1), ("ray_sub", 0), ("grid_blocks", 0),
                 ("overlap", 0), ("overlap_trace_streams", 1), ("trace_block", 512), ("trace_blocks_per_cu", 0), ("disc_pipeline", 1),
                 ("bin_mode", 1), ("pipeline_chunk", 1 << 26), ("surface_pipeline", 1), ("rays_per_lane", 0)):
        isx.set_option(k, v)


def _surface(mod, kind):
    c = mod.default_config()
    if kind == "lobe":
        c.surface_model = 1
    elif kind == "lobe_nlcopy":       # the macro's own geometry block ("nonLambertianFlux copy.C":213-226): rho 1, limit 10000, box 200, src z -80
        c.surface_model = 1; c.reflectance = 1.0; c.max_points = 10000; c.box_half = 200.0; c.src[2] = -80.0
        c.lambertian = 0; c.roughness_rad = 0.5
    elif kind == "rough_0p5":
        c.lambertian = 0; c.roughness_rad = 0.5
    elif kind == "rough_0p01":
        c.lambertian = 0; c.roughness_rad = 0.01
    elif kind == "specular":          # sigma = 0: a mirror ball (the pencil circulates in its plane of incidence)
        c.lambertian = 0; c.roughness_rad = 0.0; c.max_points = 400
    elif kind == "compat":
        c.hit_line_mode = 1
    elif kind == "compat_chord":
        c.hit_line_mode = 1; c.trace_mode = 1
    elif kind == "compat_brdf":
        c.hit_line_mode = 1; c.source_model = 1; c.brdf[0], c.brdf[1], c.brdf[2] = 0.3, 0.4, 0.6
        c.roughness_rad = 0.5; c.reflectance = 1.0; c.max_points = 10000; c.box_half = 200.0
    elif kind == "lobe_compat":
        c.surface_model = 1; c.hit_line_mode = 1
    else:
        raise ValueError(kind)
    return c


@pytest.mark.parametrize("kind,n", [("lobe", 500_000), ("lobe_nlcopy", 200_000), ("rough_0p5", 500_000), ("rough_0p01", 500_000),
                                    ("specular", 100_000), ("compat", 500_000), ("compat_chord", 200_000), ("compat_brdf", 200_000),
                                    ("lobe_compat", 200_000)])
def test_surface_pipeline_equals_the_oracle(isx, orc, kind, n):
    """Histogram and census of the assist-wave pipeline == CPU oracle for every border model / hit-line mode it now serves."""
    _reset(isx)
    gh, gst = isx.fluxmap(_surface(isx, kind), n, SEED, 3)
    kinds = isx.last_kernel_ms()
    oh, ost = orc.fluxmap(_surface(orc, kind), n, SEED, 3)
    assert kinds[1] > 0 and kinds[2] > 0 and kinds[0] == 0, "the two-kernel pipeline ran, not the fused kernel"
    assert np.array_equal(gh, oh)
    _same(gst, ost)
    assert int(gh.sum()) == gst.bin_increments and gst.launched == n


@pytest.mark.parametrize("kind", ["lobe", "rough_0p5", "compat"])
def test_surface_pipeline_equals_the_fused_kernel_of_round_1(isx, kind):
    """surface_pipeline = 0 keeps round 1's isx_trace_bin_full_kernel reachable: the same histogram and census."""
    _reset(isx)
    n = 150_000
    try:
        a, sa = isx.fluxmap(_surface(isx, kind), n, SEED, 11)
        isx.set_option("surface_pipeline", 0)
        b, sb = isx.fluxmap(_surface(isx, kind), n, SEED, 11)
        assert isx.last_kernel_ms()[0] > 0 and isx.last_kernel_ms()[1] == 0
        assert np.array_equal(a, b)
        _same(sa, sb)
    finally:
        _reset(isx)


@pytest.mark.parametrize("block", [128, 256, 448, 768])
def test_lobe_pipeline_in_every_workgroup_shape(isx, orc, block):
    """The lobe's tries are spread over the steps of a lane (Ray::k); a wave that gives its last rays away finishes the tries in
    flight first.  Small launches in several workgroup shapes (1 .. 11 tracer waves per assist wave), small ray-queue portions and a
    grid of one workgroup: every end-game path, same histogram."""
    _reset(isx)
    n = 60_000
    oh, ost = orc.fluxmap(_surface(orc, "lobe"), n, SEED, 17)
    try:
        isx.set_option("assist_block", block)
        for grid, sub in ((0, 0), (1, 64), (3, 192)):
            isx.set_option("grid_blocks", grid); isx.set_option("ray_sub", sub)
            gh, gst = isx.fluxmap(_surface(isx, "lobe"), n if grid != 1 else 6_000, SEED, 17)
            if grid != 1:
                assert np.array_equal(gh, oh)
                _same(gst, ost)
            else:
                o1, s1 = orc.fluxmap(_surface(orc, "lobe"), 6_000, SEED, 17)
                assert np.array_equal(gh, o1)
                _same(gst, s1)
    finally:
        _reset(isx)


def test_surface_series_and_chunks(isx, orc):
    """The series driver with one configuration per border model, and a flux map cut into chunks (pipeline_chunk) with the
    origin-compat rewrite between each chunk's trace and binning kernels."""
    _reset(isx)
    n = 40_000
    cfgs = [_surface(isx, k) for k in ("lobe", "compat", "rough_0p5")]
    hits, sts = isx.fluxmap_series(cfgs, n, SEED, 100)
    for k, name in enumerate(("lobe", "compat", "rough_0p5")):
        oh, ost = orc.fluxmap(_surface(orc, name), n, SEED, 100 + k * n)
        assert np.array_equal(hits[k], oh), name
        _same(sts[k], ost)
    try:
        isx.set_option("pipeline_chunk", 8192)
        gh, gst = isx.fluxmap(_surface(isx, "compat"), 50_000, SEED, 1)
        oh, ost = orc.fluxmap(_surface(orc, "compat"), 50_000, SEED, 1)
        assert np.array_equal(gh, oh)
        _same(gst, ost)
    finally:
        _reset(isx)


def test_bounded_wait_failure_is_reported_not_silent():
    """libisx_giveup.so is libisx.so built with -DISX_TEST_GIVEUP -- the assist wave of workgroup 0 finds "no
    room on the resume ring and no progress" at its first batch.  The wave must leave its rays unwritten (the ring stays consistent),
    every wave of the workgroup must leave promptly, and the host must report ISX_ERR_HIP instead of a short histogram."""
    lib = os.path.join(ROOT, "altair-raytracing_amd", "csrc", "libisx_giveup.so")
    assert os.path.exists(lib), "libisx_giveup.so not built (make -C altair-raytracing_amd/csrc libisx_giveup.so)"
    code = r"""
import sys, time
sys.path.insert(0, %r)
import altair_raytracing_amd as isx
isx.load(); isx.init(0)
t0 = time.time()
try:
    isx.fluxmap(isx.default_config(), 200000, 1)
    print("NO-ERROR")
except isx.IsxError as e:
    print("STATUS", e.status, "HIP", isx.load().isx_last_hip_error(), "T", round(time.time() - t0, 3))
# the library is usable afterwards (the failed launch left no wave behind): an option that avoids the assist wave gives a map
isx.set_option("assist", 0)
h, st = isx.fluxmap(isx.default_config(), 20000, 1)
print("AFTER", int(h.sum()) == st.bin_increments, st.launched)
isx.shutdown()
""" % ROOT
    env = dict(os.environ, ISX_LIB_PATH=lib)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.strip().splitlines()
    status = [ln for ln in lines if ln.startswith("STATUS")]
    assert status and status[0].split()[1] == "-4", out.stdout          # ISX_ERR_HIP
    assert float(status[0].split()[-1]) < 30.0, "the failed launch must end promptly, not after every wave's spin limit"
    assert "AFTER True 20000" in out.stdout


@pytest.mark.parametrize("name", ["pp_03_31_3", "pp_04_1_4"])
def test_cut_short_reference_maps_bin_by_bin(isx, name):
    """The two per-position runs of the reference that were interrupted (direction (5,6,0): 12 867 rows; port 175 deg: 2 715 rows,
    tests/golden/reference_maps.npz), traced the reference's way -- 50 000 fresh rays per position, only the positions the file
    holds -- and compared bin by bin with binomial sigmas, like the seven complete maps (tests/test_gpu_round2.py).  The 175-degree
    file ends at theta = 15 deg, where the known smooth residual is +1.3 %: its total ratio is 0.988, inside the common window."""
    import json
    z = np.load(os.path.join(ROOT, "tests", "golden", "reference_maps.npz"))
    info = [i for i in json.loads(str(z["index_json"])) if i["name"] == name][0]
    ref = z[name + "_hits"].astype(np.int64).reshape(-1)
    rows = info["rows_present"]
    c = isx.default_config()
    c.theta_max_deg = info["port_deg"]
    for a in range(3):
        c.src[a] = info["source_position"][a]; c.dir[a] = info["source_direction"][a]
    n = info["rays_per_position"]
    h, st = isx.fluxmap_per_position(c, n, 777, 1, 0, rows)
    h = h.reshape(-1).astype(np.int64)
    assert st.launched == rows * n and h[rows:].sum() == 0
    r, o = ref[:rows], h[:rows]
    p = (r + o) / (2.0 * n)
    use = p * n >= 5
    chi2 = (((r - o) ** 2 / (2.0 * n * p * (1 - p) + 1e-300))[use]).sum() / use.sum()
    ratio = o.sum() / r.sum()
    print(f"{name}: {rows} rows, chi2/dof {chi2:.4f} ({use.sum()} bins), total ratio {ratio:.5f}")
    assert use.sum() > 0.95 * rows and chi2 < 1.08, (name, chi2)
    assert 0.985 < ratio < 1.015, (name, ratio)


def test_lobe_unit_vectors_are_tvector3_unit(isx):
    """lobe_try() normalises with tv_unit_n(): sqrt / reciprocal without the general expansions' range scaling where the squared
    length is in [2^-200, 2^200], TVector3::Unit's plain operations elsewhere.  Bit-equal to a * (1.0 / sqrt(x*x + y*y + z*z)) over
    the operand families of the sampler: near-unit normals of the inner sphere, (w_z, 0, -w_x) projections down to 1e-30 and to
    exact zero, near-unit combinations -- and wild magnitudes on both sides of the guard."""
    rng = np.random.default_rng(5)
    n = 2_000_000
    v = rng.standard_normal((n, 3))
    v /= np.linalg.norm(v, axis=1)[:, None]
    fam = [v * (1.0 + rng.uniform(-1e-12, 1e-12, (n, 1))),                                    # normals: q * (-1/r_in)
           np.stack([v[:, 2], np.zeros(n), -v[:, 0]], 1),                                      # TVector3(0,1,0).Cross(w)
           np.stack([v[:, 2], np.zeros(n), -v[:, 0]], 1) * 10.0 ** rng.uniform(-30, 0, (n, 1)),
           v * 10.0 ** rng.uniform(-140, 140, (n, 1)),                                          # across the guard at 2^-+100 in length
           np.array([[0.0, 0.0, 0.0], [0.0, 1.0, 0.0], [1e-160, 0.0, 0.0], [3.0, 4.0, 12.0], [2.0 ** -100, 0.0, 0.0], [2.0 ** 100, 0.0, 0.0]])]
    for a in fam:
        x, y, z = (np.ascontiguousarray(a[:, k]) for k in range(3))
        tot2 = x * x + y * y + z * z
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            tot = np.where(tot2 > 0, 1.0 / np.sqrt(tot2), 1.0)
        for op, comp in ((12, x), (13, y), (14, z)):
            got = isx.mathprobe(op, x, y, z)
            want = comp * tot
            assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), op


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 50_000, 149_999, 150_000, 299_999, 300_000, 999_999, 1_000_000])
def test_small_launch_shapes_change_nothing(isx, orc, n):
    """small_shape(): below 1e6 rays a launch takes 256-thread workgroups and 1 / 2 / 4 rays per tracer lane, its binning grid is
    sized by work units, results come back through the pinned staging buffer.  Every boundary of that policy, every entry point a
    small call uses: == oracle (flux map + census, one detector, disc sweep)."""
    _reset(isx)
    c, co = isx.default_config(), orc.default_config()
    gh, gst = isx.fluxmap(c, n, SEED, 7)
    oh, ost = orc.fluxmap(co, n, SEED, 7)
    assert np.array_equal(gh, oh)
    _same(gst, ost)
    if n <= 300_000:
        det = isx.detector_table(c)[20 * c.n_phi + 3]
        g1, s1 = isx.trace_rays_detector(c, det, c.det_diameter, n, SEED, 11)
        o1, t1 = orc.trace_rays_detector(co, det, co.det_diameter, n, SEED, 11)
        assert g1 == o1 and s1.counted_below_z == t1.counted_below_z and s1.wall_hits == t1.wall_hits
        discs = np.array([[0.0, 0.0, -200.0, 0.0, 0.0, 1.0], [30.0, 0.0, -197.0, 0.15, 0.0, 0.99]])
        c3, o3 = isx.default_config(), orc.default_config()
        for q in (c3, o3):
            q.r_out = 105.0; q.reflectance = 1.0; q.max_points = 2000; q.box_half = 250.0
        gd, _ = isx.disc_sweep(c3, discs, 5.0, 0.1, n, SEED, 3)
        od, _ = orc.disc_sweep(o3, discs, 5.0, 0.1, n, SEED, 3)
        assert np.array_equal(gd, od)


def test_rays_per_lane_and_block_options(isx, orc):
    """The two knobs of the launch shape (`rays_per_lane`, `assist_block`) over a mid-size launch: same histogram, same census."""
    _reset(isx)
    n = 200_000
    oh, ost = orc.fluxmap(orc.default_config(), n, SEED, 1)
    try:
        for rpl in (0, 1, 3, 16, 4096):
            for blk in (0, 128, 768):
                isx.set_option("rays_per_lane", rpl); isx.set_option("assist_block", blk)
                gh, gst = isx.fluxmap(isx.default_config(), n, SEED, 1)
                assert np.array_equal(gh, oh), (rpl, blk)
                _same(gst, ost)
    finally:
        isx.set_option("rays_per_lane", 0)
        _reset(isx)


def test_surface_pipeline_under_every_host_side_schedule(isx, orc):
    """The host-side switches around the new kernels: the overlapped pipeline (trace of chunk k+1 on one stream, compat rewrite +
    binning of chunk k on another), kernels without an assist wave, the fused kernel, brute-force binning -- same histogram."""
    _reset(isx)
    n = 120_000
    want = {k: orc.fluxmap(_surface(orc, k), n, SEED, 2) for k in ("lobe", "compat", "rough_0p5")}
    try:
        for opts in ({"overlap": 3}, {"overlap": 2, "overlap_trace_streams": 2}, {"assist": 0}, {"pipeline": 0}, {"bin_mode": 0},
                     {"bin_cols": 0}, {"bin_slots": 0}, {"pipeline_chunk": 20_000, "overlap": 2}):
            _reset(isx)
            for key, val in opts.items():
                isx.set_option(key, val)
            for k, (oh, ost) in want.items():
                gh, gst = isx.fluxmap(_surface(isx, k), n, SEED, 2)
                assert np.array_equal(gh, oh), (opts, k)
                _same(gst, ost)
    finally:
        _reset(isx)


@pytest.mark.parametrize("kind", ["lobe", "lobe_nlcopy", "rough_0p5"])
def test_per_position_sinks_with_the_other_borders(isx, orc, kind):
    """The macros that own these borders sweep one detector position at a time with fresh rays ("nonLambertianFlux copy.C":306-345):
    isx_fluxmap_per_position (both folds) and isx_trace_rays_detector on isx_trace_assist_perpos_{lobe,rough}_kernel == oracle,
    and == round 1's kernel (surface_pipeline = 0)."""
    _reset(isx)
    c, co = _surface(isx, kind), _surface(orc, kind)
    for q in (c, co):
        q.n_theta, q.n_phi, q.det_diameter = 45, 20, 10.0
    for fold in (1, 2):
        gh, gst = isx.fluxmap_per_position(c, 600, SEED, fold)
        oh, ost = orc.fluxmap_per_position(co, 600, SEED, fold)
        assert np.array_equal(gh, oh), (kind, fold)
        _same(gst, ost)
    det = isx.detector_table(c)[5 * c.n_phi + 7]
    g1, s1 = isx.trace_rays_detector(c, det, c.det_diameter, 200_000, SEED, 4)
    o1, t1 = orc.trace_rays_detector(co, det, co.det_diameter, 200_000, SEED, 4)
    assert g1 == o1 and g1 > 0 and s1.wall_hits == t1.wall_hits and s1.counted_below_z == t1.counted_below_z
    try:
        isx.set_option("surface_pipeline", 0)
        bh, bst = isx.fluxmap_per_position(c, 600, SEED, 1)
        oh, ost = orc.fluxmap_per_position(co, 600, SEED, 1)
        assert np.array_equal(bh, oh)
        _same(bst, ost)
    finally:
        _reset(isx)


@pytest.mark.parametrize("kind,n", [("lobe", 20_000_000), ("compat", 30_000_000), ("rough_0p5", 20_000_000)])
def test_new_pipelines_culled_equals_brute_force_at_scale(isx, kind, n):
    """Tens of millions of rays through the new trace kernels + the column-slot binning kernel against the brute-force reference-order
    test of all 16 200 positions (bin_mode 0: round 1's fused kernel, every detector tested): the lobe's exit lines are more
    collimated than the Lambertian border's (254 bins per ray against 114), the origin-compat lines all pass through the origin --
    regimes of the cull that the headline's lines do not reach.  ~5e9 hit decisions each."""
    _reset(isx)
    c = _surface(isx, kind)
    try:
        isx.set_option("bin_mode", 0)
        brute, sb = isx.fluxmap(c, n, 13579)
    finally:
        _reset(isx)
    culled, sc = isx.fluxmap(c, n, 13579)
    assert isx.last_kernel_ms()[2] > 0
    assert np.array_equal(brute, culled)
    _same(sb, sc)
    assert int(culled.sum()) == sc.bin_increments > 0


@pytest.mark.parametrize("kind", ["lobe", "rough_0p5", "compat"])
def test_new_pipelines_are_partition_invariant(isx, kind):
    """A ray's history -- every try of the lobe sampler, every roughness draw -- is a function of (seed, ray index) alone: 2e7 rays
    in one call == the sum of five calls over the same index range cut at odd places, histogram and census (what the ray-sharded
    multi-GPU runs rest on)."""
    _reset(isx)
    c = _surface(isx, kind)
    n, first = 20_000_000, 123_456_789_000
    whole, sw = isx.fluxmap(c, n, SEED, first)
    cuts = [0, 1, 3_333_333, 3_333_397, 11_000_000, n]
    acc = np.zeros_like(whole)
    tot = {k: 0 for k in CENSUS}
    for a, b in zip(cuts[:-1], cuts[1:]):
        h, st = isx.fluxmap(c, b - a, SEED, first + a)
        acc += h
        for k in CENSUS:
            tot[k] += getattr(st, k)
    assert np.array_equal(whole, acc)
    for k in CENSUS:
        assert tot[k] == getattr(sw, k), k
