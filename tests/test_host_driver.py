"""Host driver (altair-raytracing_amd/host): the reference's macro entry points and file formats.

CPU part: writers + naming, against the format of the reference's committed CSVs.
GPU part (-m gpu): the CLI runs the entry points end to end; files are parsed with the same contract
as flux_analysis.py:11-57 and compared with the C ABI called directly."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "altair-raytracing_amd", "host")
CLI = os.path.join(HOST, "isx_macro")


def parse_fluxmap(path):
    """flux_analysis.py:11-57 — '#' lines are 'key: value' metadata (split on the first ':'), then a CSV."""
    meta, rows, header = {}, [], None
    with open(path) as f:
        for line in f:
            if line.startswith("#"):
                if ":" in line:
                    k, v = line[1:].strip().split(":", 1)
                    meta[k.strip()] = v.strip()
            elif header is None:
                header = line.strip()
            elif line.strip():
                rows.append([float(x) for x in line.strip().split(",")])
    return meta, header, np.array(rows)


@pytest.fixture(scope="module")
def cli():
    if not os.path.exists(CLI):
        subprocess.check_call(["make", "-s", "-C", HOST])
    return CLI


def test_writer_matches_reference_format(cli, golden, tmp_path):
    out = tmp_path / "w.csv"
    subprocess.check_call([cli, "--selftest-writer", str(out)])
    lines = out.read_text().splitlines()
    g = golden["csv_format_sample"]
    # header: byte-identical to results_overnight_03_31.../fluxmap_50000rays_180x90_src-60_0_-75.csv:1-15
    assert lines[:15] == g["header"]
    assert len(lines) == 15 + 16200
    # rows: %.6f,%.6f,%.6f, theta-major then phi
    assert [l.split(",")[:2] for l in lines[15:18]] == [l.split(",")[:2] for l in g["first_rows"]]
    assert all(re.fullmatch(r"\d+\.\d{6},\d+\.\d{6},\d\.\d{6}", l) for l in lines[15:])
    assert lines[-1].startswith("89.750000,358.000000,")
    meta, header, rows = parse_fluxmap(out)
    assert header == "theta,phi,fraction" and rows.shape == (16200, 3)
    assert list(meta)[:3] == ["Flux Map Data - Generated", "Number of rays per position", "Detector dimensions"]
    assert meta["Exit port angle"] == "170 degrees" and meta["Source position (x,y,z)"] == "-60cm, 0cm, -75cm"
    # synthetic map of the self-test: hits[k] = (k*7919) % 1000, n = 50000
    k = np.arange(16200)
    assert np.allclose(rows[:, 2], np.round(((k * 7919) % 1000) / 50000.0, 6), atol=5e-7)


def test_unique_filename_never_overwrites(cli, tmp_path):
    """getUniqueFilename (fluxAtObserverOptimize.C:336-387): stem_1.ext, stem_2.ext ..."""
    base = tmp_path / "fluxmap_50000rays_180x90_src-60_0_-75.csv"
    def uniq():
        return subprocess.check_output([cli, "--unique", str(base)], text=True).strip()
    assert uniq() == str(base)
    base.write_text("x")
    assert uniq() == str(tmp_path / "fluxmap_50000rays_180x90_src-60_0_-75_1.csv")
    (tmp_path / "fluxmap_50000rays_180x90_src-60_0_-75_1.csv").write_text("x")
    assert uniq() == str(tmp_path / "fluxmap_50000rays_180x90_src-60_0_-75_2.csv")
    noext = tmp_path / "detector_sweep"
    noext.write_text("x")
    assert subprocess.check_output([cli, "--unique", str(noext)], text=True).strip() == str(noext) + "_1"


def test_theta_analysis_matches_scipy(cli, golden, tmp_path):
    """isx_macro --analyze == the numeric part of flux_analysis.py (per-theta mean, a*cos(b*theta)+c fit, R^2)."""
    from scipy.optimize import curve_fit
    m = [m for m in golden["per_position_maps"] if m["port_deg"] == 170.0][0]
    prof = np.array(m["theta_profile"])
    rng = np.random.default_rng(4)
    path = tmp_path / "map.csv"
    with open(path, "w") as f:
        f.write("# Flux Map Data - Generated: 2025-04-01 01:42:14\n# Exit port angle: 170 degrees\ntheta,phi,fraction\n")
        rows = []
        for i in range(180):
            for j in range(90):
                v = max(0.0, prof[i] * (1 + 0.05 * rng.standard_normal()))
                rows.append(((i + .5) * .5, (j + .5) * 4, float(f"{v:.6f}")))
                f.write(f"{rows[-1][0]:.6f},{rows[-1][1]:.6f},{v:.6f}\n")
        f.write("# Sweep completed at: 2025-04-01 05:10:58\n")
    out = subprocess.check_output([cli, "--analyze", str(path)], text=True)
    a, b, c = [float(x) for x in re.search(r"a=([-\d.]+), b=([-\d.]+), c=([-\d.]+)", out).groups()]
    r2 = float(re.search(r"R-squared value: ([-\d.]+)", out).group(1))
    arr = np.array(rows)
    theta = np.unique(arr[:, 0])
    means = np.array([arr[arr[:, 0] == t, 2].mean() for t in theta])

    def cosine_func(x, a, b, c):
        return a * np.cos(np.deg2rad(b * x)) + c
    p0 = [(means.max() - means.min()) / 2, 1.0, means.mean()]
    popt, _ = curve_fit(cosine_func, theta, means, p0=p0)
    assert [a, b, c] == pytest.approx(list(popt), abs=2e-5)
    res = means - cosine_func(theta, *popt)
    assert r2 == pytest.approx(1 - (res ** 2).sum() / ((means - means.mean()) ** 2).sum(), abs=2e-5)
    rep = np.loadtxt(tmp_path / "map_theta_analysis.txt", delimiter=",", comments="#", skiprows=5)
    assert rep.shape == (180, 6) and np.allclose(rep[:, 1], means, atol=1e-9)
    # analytic overlays: closed form of finitePort/test.py:11-14 and the grid sum of projectionFactor.py:19-46
    f = (1 - np.cos(np.deg2rad(10.0))) / 2
    p_exit = f / (1 - 0.99 * (1 - f))
    assert np.allclose(rep[:, 4], p_exit * 0.04 * np.cos(np.deg2rad(theta)), rtol=1e-9)

    def projection(t, R, r_p, n=100):
        r, ph = np.meshgrid(np.linspace(0, r_p, n), np.linspace(0, 2 * np.pi, n))
        den = np.sqrt(np.maximum(R ** 2 + r ** 2 - 2 * R * r * np.sin(ph) * np.tan(t), 1e-10))
        c = np.clip((R - r * np.sin(ph) * np.tan(t)) / den, -1, 1)
        return np.sum(c * r * (r_p / n) * (2 * np.pi / n))
    pf = np.array([projection(np.deg2rad(t), 100.1, 100.1 * np.sin(np.deg2rad(10.0))) for t in theta])
    assert np.allclose(rep[:, 5], p_exit * 0.04 * pf / pf.max(), rtol=1e-8)


def test_folder_average_matches_pandas(cli, tmp_path):
    """isx_macro --analyze <folder> average == flux_analysis.py:128-160,182-192 (groupby theta,phi over files:
    mean, std/sqrt(count); then per-theta means) + the same cosine fit."""
    import pandas as pd
    from scipy.optimize import curve_fit
    rng = np.random.default_rng(11)
    folder = tmp_path / "series"
    folder.mkdir()
    frames = []
    for k in range(3):
        rows = [((i + .5) * 3.0, (j + .5) * 36.0, round(max(0.0, 0.016 * np.cos(np.deg2rad((i + .5) * 3.0)) * (1 + 0.1 * rng.standard_normal())), 6))
                for i in range(30) for j in range(10)]
        with open(folder / f"fluxmap_{k}.csv", "w") as f:
            f.write("# Flux Map Data - Generated: x\n# Exit port angle: 164 degrees\n# Mirror reflectance: 0.99\ntheta,phi,fraction\n")
            f.writelines(f"{t:.6f},{p:.6f},{v:.6f}\n" for t, p, v in rows)
        frames.append(pd.DataFrame(rows, columns=["theta", "phi", "fraction"]))
    (folder / "notes.txt").write_text("ignored")
    out = subprocess.check_output([cli, "--analyze", str(folder), "AVERAGE"], text=True, cwd=tmp_path)
    assert "Averaging data across all files..." in out
    blocks = re.findall(r"File: (\S+)\n  Fit parameters: a=([-\d.]+), b=([-\d.]+), c=([-\d.]+)\n  R-squared value: ([-\d.]+)", out)
    assert [b[0] for b in blocks] == ["fluxmap_0.csv", "fluxmap_1.csv", "fluxmap_2.csv", "AVERAGE"]
    g = pd.concat(frames, ignore_index=True).groupby(["theta", "phi"])["fraction"]
    avg = g.mean().reset_index()
    avg["stderr"] = (g.std() / np.sqrt(g.size())).values
    gt = avg.groupby("theta")
    theta, means, errs = np.array(gt["fraction"].mean().index), gt["fraction"].mean().values, gt["stderr"].mean().values

    def cosine_func(x, a, b, c):
        return a * np.cos(np.deg2rad(b * x)) + c
    popt, _ = curve_fit(cosine_func, theta, means, p0=[(means.max() - means.min()) / 2, 1.0, means.mean()])
    assert [float(x) for x in blocks[3][1:4]] == pytest.approx(list(popt), abs=2e-5)
    txt = (tmp_path / "series_averaged_theta_comparison.txt").read_text().split("# AVERAGE\n")[1].splitlines()[1:]
    rep = np.array([[float(x) for x in ln.split(",")] for ln in txt])
    assert rep.shape == (30, 6)
    assert np.allclose(rep[:, 0], theta) and np.allclose(rep[:, 1], means, atol=1e-12) and np.allclose(rep[:, 2], errs, atol=1e-12)
    # without "average": per-file analyses only
    out2 = subprocess.check_output([cli, "--analyze", str(folder)], text=True, cwd=tmp_path)
    assert "AVERAGE" not in out2 and (tmp_path / "series_theta_comparison.txt").exists()


def test_rank_sharding_rule_matches_python(cli):
    """isxhost::Comm::shard (C++ driver, one process per GPU) == altair_raytracing_amd.shard (bench.py / sharding.py)."""
    from altair_raytracing_amd import shard
    for n in (0, 1, 7, 16200, 50_000_000, 10 ** 9 + 3):
        for world in (1, 2, 3, 8):
            for rank in range(world):
                env = dict(os.environ, ISX_RANK=str(rank), ISX_WORLD=str(world))
                out = subprocess.check_output([cli, "--shard", str(n)], text=True, env=env).split()
                assert [int(x) for x in out] == [rank, world, *shard(n, rank, world)]
    # torchrun-style variables are understood too
    env = {k: v for k, v in os.environ.items() if not k.startswith("ISX_")}
    env.update(RANK="1", WORLD_SIZE="4", LOCAL_RANK="1")
    assert subprocess.check_output([cli, "--shard", "10"], text=True, env=env).split() == ["1", "4", "3", "3"]


def test_cli_fails_loudly_without_gpu(cli, tmp_path):
    import altair_raytracing_amd as isx
    if isx.load().isx_init(0) == 0:
        isx.load().isx_shutdown()
        pytest.skip("GPU present")
    r = subprocess.run([cli, "makeIntegratingSphereNRays"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode != 0 and "no CPU fallback" in r.stderr


def test_collective_status_with_a_failing_rank(cli, tmp_path):
    """ranks must not diverge before the collective.  tests/native/comm_stub_test.cpp joins 2 and 8 "ranks"
    (threads) through an in-process Transport double: SUM/MAX semantics, and with ONE rank failing every rank returns
    that rank's error and nobody is left waiting (the run is under a time-out)."""
    exe = tmp_path / "comm_stub_test"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-o", str(exe),
                           os.path.join(ROOT, "tests", "native", "comm_stub_test.cpp"), "-L" + HOST, "-lisx_macros",
                           "-Wl,-rpath," + HOST, "-lpthread"])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr


def test_multirank_launch_stops_instead_of_hanging(cli, tmp_path):
    """The pre-flight of isx_comm.cpp: a rank that cannot bind its GPU says so in the job's rendezvous directory, so no
    rank enters ncclCommInitRank (which has no time-out).  Here (no GPU) both ranks of a 2-rank launch must stop within
    seconds; a multi-rank launch without a per-job nonce is refused."""
    import time
    import altair_raytracing_amd as isx
    if isx.load().isx_init(0) == 0:
        isx.load().isx_shutdown()
        pytest.skip("GPU present")
    args = ("fluxAtObserverFast::sweepDetectorTraceOnce", "folder=out", "srcZ=-75", "dirY=0", "thetaMax=170")
    base = {k: v for k, v in os.environ.items() if not k.startswith("ISX_") and k not in ("MASTER_PORT", "TORCHELASTIC_RUN_ID", "RANK", "WORLD_SIZE")}
    t0 = time.time()
    procs = []
    for rank in (0, 1):
        env = dict(base, ISX_QUIET="1", ISX_RAYS="1000", ISX_RANK=str(rank), ISX_WORLD="2", ISX_JOB_ID="t-%d" % os.getpid(),
                   ISX_RENDEZVOUS=str(tmp_path))
        procs.append(subprocess.Popen([cli, *args], cwd=tmp_path, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=60) for p in procs]
    assert time.time() - t0 < 30
    for p, (_, err) in zip(procs, outs):
        assert p.returncode != 0 and ("could not bind its GPU" in err or "no CPU fallback" in err), err
    env = dict(base, ISX_QUIET="1", ISX_RAYS="1000", ISX_RANK="0", ISX_WORLD="2", ISX_RENDEZVOUS=str(tmp_path))
    r = subprocess.run([cli, *args], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "per-job nonce" in r.stderr, r.stderr
    # files of an EARLIER launch with the same job tag (a fail.<r> that was never removed, a ready.<r> of a rank
    # that died) carry that launch's nonce and must be ignored, whatever their age.  Relaunch at once with the same tag ...
    for attempt in range(2):
        procs = []
        for rank in (0, 1):
            env = dict(base, ISX_QUIET="1", ISX_RAYS="1000", ISX_RANK=str(rank), ISX_WORLD="2", ISX_JOB_ID="t-%d" % os.getpid(),
                       ISX_RENDEZVOUS=str(tmp_path))
            procs.append(subprocess.Popen([cli, *args], cwd=tmp_path, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        for p in procs:
            _, err = p.communicate(timeout=60)
            assert p.returncode != 0 and ("could not bind its GPU" in err or "no CPU fallback" in err), err
    # ... and a lone rank 0 that finds a well-formed ready.1 of another launch does not take it for a live rank (it would then
    # publish the id and block in ncclCommInitRank for ever): it reports the missing rank after the rendezvous wait.
    import struct
    jobdir = [d for d in os.listdir(tmp_path) if d.startswith("isx_rdzv_")]
    job = "stale-%d" % os.getpid()
    rd = os.path.join(str(tmp_path), "isx_rdzv_%d_%s" % (os.getuid(), job))
    os.makedirs(rd, mode=0o700, exist_ok=True)
    for name in ("ready.1", "fail.1", "launch"):
        with open(os.path.join(rd, name), "wb") as f:
            f.write(b"ISXRDZV2" + struct.pack("<Q", 0x1234567))
    env = dict(base, ISX_QUIET="1", ISX_RAYS="1000", ISX_RANK="0", ISX_WORLD="2", ISX_JOB_ID=job, ISX_RENDEZVOUS=str(tmp_path),
               ISX_RENDEZVOUS_WAIT="2", ISX_COMM_ASSUME_DEVICE="1")
    t0 = time.time()
    r = subprocess.run([cli, *args], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "not every rank showed up" in r.stderr and time.time() - t0 < 30, r.stderr
    # ... nor does a lone FOLLOWER adopt the `launch` / `ready.*` leftovers of a killed launch with its tag (it would
    # read that launch's id and block in ncclCommInitRank): it trusts a nonce only after rank 0 of THIS launch has echoed the
    # follower's own fresh token (hello.<rank> -> ack.<rank>), which no leftover can contain.
    for name in ("ready.0", "ready.1", "launch", "ack.1", "hello.1"):
        with open(os.path.join(rd, name), "wb") as f:
            f.write(b"ISXRDZV2" + struct.pack("<Q", 0x1234567))
    env = dict(env, ISX_RANK="1")
    t0 = time.time()
    r = subprocess.run([cli, *args], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "not every rank showed up" in r.stderr and time.time() - t0 < 30, r.stderr
    del jobdir


# --------------------------------------------------------------------------------------------- GPU
def _run(cli, cwd, entry, *args, rays=None, seed=None, **extra_env):
    env = dict(os.environ, ISX_QUIET="1", **extra_env)
    if rays is not None:
        env["ISX_RAYS"] = str(rays)
    if seed is not None:
        env["ISX_SEED"] = str(seed)
    r = subprocess.run([cli, entry, *args], cwd=cwd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    return r.stdout


@pytest.mark.gpu
def test_traceonce_entry_point(cli, isx, tmp_path):
    """sweepDetectorTraceOnce(false, folder, 1, -60,0,-75, 5,0,0, 170) (fluxAtObserverFast.C:1068)."""
    _run(cli, tmp_path, "fluxAtObserverFast::sweepDetectorTraceOnce", "folder=out", "srcZ=-75", "dirY=0", "thetaMax=170",
         rays=100000, seed=4242)
    path = tmp_path / "out" / "fluxmap_traceonce_100000rays_180x90_src-60_0_-75.csv"
    meta, header, rows = parse_fluxmap(path)
    assert header == "theta,phi,fraction" and rows.shape == (16200, 3)
    assert meta["Number of rays"] == "100000" and meta["Method"].startswith("Trace-Once")
    for k in ("Sweep completed at", "Total execution time", "Ray tracing time", "Detector sweep time", "Total rays exiting port"):
        assert k in meta
    hits, st = isx.fluxmap(isx.default_config(), 100000, 4242, 0)
    assert meta["Total rays exiting port"] == f"{st.counted_below_z} out of 100000"
    want = np.array([float(f"{h / 100000.0:.6f}") for h in hits.reshape(-1)])
    assert np.array_equal(rows[:, 2], want)
    assert np.allclose(rows[:90, 1], (np.arange(90) + .5) * 4) and rows[0, 0] == 0.25 and rows[-1, 0] == 89.75
    # a second run never overwrites
    _run(cli, tmp_path, "fluxAtObserverFast::sweepDetectorTraceOnce", "folder=out", "srcZ=-75", "dirY=0", rays=1000)
    assert (tmp_path / "out" / "fluxmap_traceonce_1000rays_180x90_src-60_0_-75.csv").exists()
    _run(cli, tmp_path, "fluxAtObserverFast::sweepDetectorTraceOnce", "folder=out", "srcZ=-75", "dirY=0", rays=1000)
    assert (tmp_path / "out" / "fluxmap_traceonce_1000rays_180x90_src-60_0_-75_1.csv").exists()


@pytest.mark.gpu
def test_per_position_and_twofold_entry_points(cli, isx, tmp_path):
    """sweepDetector (fluxAtObserverOptimize.C:433) and sweepDetectorTwofold (fluxAtObserverFast.C:518), 200 rays/position."""
    _run(cli, tmp_path, "fluxAtObserverOptimize::sweepDetector", "folder=pp", "srcZ=-75", "dirY=0", "thetaMax=166", rays=200, seed=7)
    meta, header, rows = parse_fluxmap(tmp_path / "pp" / "fluxmap_200rays_180x90_src-60_0_-75.csv")
    c = isx.default_config(); c.theta_max_deg = 166.0
    hits, st = isx.fluxmap_per_position(c, 200, 7, 1)
    assert np.array_equal(rows[:, 2], np.array([float(f"{h / 200.0:.6f}") for h in hits.reshape(-1)]))
    assert meta["Exit port angle"] == "166 degrees"
    assert meta["Total ray hits"] == f"{int(hits.sum())} out of {200 * 16200}"
    assert re.fullmatch(r"\d+\.\d{6} seconds", meta["Total execution time"])   # sticky fixed/6 format of the reference
    _run(cli, tmp_path, "fluxAtObserverFast::sweepDetectorTwofold", "folder=tf", "srcZ=-75", "dirY=0", rays=200, seed=7)
    meta, header, rows = parse_fluxmap(tmp_path / "tf" / "fluxmap_twofold_200rays_180x90_src-60_0_-75.csv")
    hits, st = isx.fluxmap_per_position(isx.default_config(), 200, 7, 2)
    # twofold row order: (theta, phi_j), (theta, phi_j+180), ...
    assert list(rows[:4, 1]) == [2.0, 182.0, 6.0, 186.0]
    order = np.array([[i * 90 + j, i * 90 + j + 45] for i in range(180) for j in range(45)]).reshape(-1)
    assert np.array_equal(rows[:, 2], np.array([float(f"{h / 200.0:.6f}") for h in hits.reshape(-1)[order]]))


@pytest.mark.gpu
def test_per_position_sweep_written_in_flushed_row_batches(cli, isx, tmp_path):
    """fluxAtObserverOptimize.C:575-579 writes and flushes every row as it is produced.  Here: ISX_FLUSH_ROWS theta rows per
    launch, written before the next launch starts -- the same rows and the same footer as the one-launch file, also when the
    batch size does not divide 180."""
    files = {}
    for rows_per_batch in ("0", "7", "180", "1"):
        _run(cli, tmp_path, "fluxAtObserverOptimize::sweepDetector", f"folder=b{rows_per_batch}", "srcZ=-75", "dirY=0", rays=60, seed=11,
             ISX_FLUSH_ROWS=rows_per_batch)
        files[rows_per_batch] = parse_fluxmap(tmp_path / f"b{rows_per_batch}" / "fluxmap_60rays_180x90_src-60_0_-75.csv")
    meta0, header0, rows0 = files["0"]
    hits, st = isx.fluxmap_per_position(isx.default_config(), 60, 11, 1)
    assert np.array_equal(rows0[:, 2], np.array([float(f"{h / 60.0:.6f}") for h in hits.reshape(-1)]))
    for k in ("7", "180", "1"):
        meta, header, rows = files[k]
        assert header == header0 and np.array_equal(rows, rows0), k
        assert meta["Total ray hits"] == meta0["Total ray hits"] == f"{int(hits.sum())} out of {60 * 16200}"


@pytest.mark.gpu
def test_small_macros(cli, isx, tmp_path):
    out = _run(cli, tmp_path, "makeIntegratingSphereNRays", seed=3)
    m = re.search(r"Flux of rays through the exit port: (\d+)", out)
    assert m and 990 <= int(m.group(1)) <= 1000       # 1000 rays, rho = 1 (no SetReflectance): every ray ends up leaving
    _run(cli, tmp_path, "nonLambertianFlux::sweepDetector", rays=300, seed=3)
    meta, header, rows = parse_fluxmap(tmp_path / "fluxmap_data.csv")
    assert header == "theta,phi,fraction" and rows.shape == (900, 3) and meta == {}
    # "nonLambertianFlux copy.C": the same sweep with the cos^2-lobe border; the file is the one isx_fluxmap_per_position gives
    # (never overwritten: the second fluxmap_data.csv of this folder is fluxmap_data_1.csv, as the reference's own second file was)
    _run(cli, tmp_path, "nonLambertianFluxCopy::sweepDetector", rays=400, seed=3)
    meta, header, rows2 = parse_fluxmap(tmp_path / "fluxmap_data_1.csv")
    assert header == "theta,phi,fraction" and rows2.shape == (900, 3) and meta == {}
    c = isx.default_config()
    c.max_points = 10000; c.box_half = 200.0; c.reflectance = 1.0; c.lambertian = 0; c.roughness_rad = 0.5; c.surface_model = 1
    c.src[2] = -80.0; c.n_theta, c.n_phi, c.det_diameter = 45, 20, 10.0
    h, st = isx.fluxmap_per_position(c, 400, 3)
    assert np.allclose(rows2[:, 2], np.round(h.reshape(-1) / 400.0, 6), atol=1e-9) and h.sum() > 0
    assert not np.array_equal(rows2[:, 2], rows[:, 2])
    # ... and its text report of one detector position (:604-667): 20 cm detector, the counts of isx_trace_rays_detector
    out = _run(cli, tmp_path, "nonLambertianFluxCopy::visualizeDetectorText", "theta=10", "phi=30", rays=20000, seed=3)
    m = re.search(r"Rays traced: (\d+)\s+Rays detected: (\d+) \(([0-9.e+-]+)%\)", out)
    assert m and int(m.group(1)) == 20000 and "Angular position: theta = 10°, phi = 30°" in out and "Exit Port at Z=-100cm" in out
    assert "<- Detector" not in out    # (as in the reference: its row index -(z+100)/10 + 15 = 24 lies outside the 16 rows it draws)
    det = np.zeros(6)
    th, ph = np.deg2rad(10.0), np.deg2rad(30.0)
    det[:3] = [100 * np.sin(th) * np.cos(ph), 100 * np.sin(th) * np.sin(ph), -100 - 100 * np.cos(th)]
    dx, dy, dz = det[0], det[1], det[2] + 100
    det[3:] = np.array([-dy, dx, dz]) / np.sqrt(dx * dx + dy * dy + dz * dz)
    c.n_theta, c.n_phi = 180, 90
    hits1, _ = isx.trace_rays_detector(c, det, 20.0, 20000, 3)
    assert int(m.group(2)) == hits1 and hits1 > 0
    _run(cli, tmp_path, "integratingSphereDetectorSweep", rays=2000, seed=3)
    txt = (tmp_path / "detector_sweep3.txt").read_text().splitlines()
    assert txt[0] == "Theta(deg)\tPhi(deg)\tHitFraction" and len(txt) == 1 + 181 * 2
    assert txt[1].startswith("-45\t0\t") and txt[2].startswith("-45\t180\t") and txt[3].startswith("-44.5\t0\t")
    _run(cli, tmp_path, "distributionSphereDetectorSweep", rays=20000, seed=3)
    ad = np.loadtxt(tmp_path / "angular_dist.txt", comments="#")
    assert ad.shape == (100, 2) and ad[0, 0] == pytest.approx(-0.99) and ad[50:, 1].sum() == 0
    log = (tmp_path / "3dRayLog.txt").read_text().splitlines()
    assert log[0] == "# dx dy dz" and len(log) - 1 == int(ad[:, 1].sum())
    dirs = np.loadtxt(tmp_path / "3dRayLog.txt", comments="#")
    assert np.abs((dirs ** 2).sum(1) - 1).max() < 1e-4 and (dirs[:, 2] < 0).all()
    h, _ = np.histogram(dirs[:, 2], bins=100, range=(-1, 1))
    # the log keeps 6 significant digits: a value within 5e-7 of a bin edge may land next door when re-binned
    # (expected ~1 of 20000 values, each flip counts twice)
    assert np.abs(h - ad[:, 1]).sum() <= 12


@pytest.mark.gpu
def test_rccl_reduce_path_of_the_cpp_driver(cli, isx, tmp_path):
    """One process per GPU + one RCCL all-reduce (isx_comm.cpp).  A 1-GPU box can only rehearse it with one rank
    (ISX_FORCE_COMM=1: same file as without RCCL); two ranks on the SAME GPU must be refused loudly, not summed wrongly."""
    def rows(path):
        return [ln for ln in open(path).read().splitlines() if not ln.startswith("#")]
    a, b = tmp_path / "plain", tmp_path / "rccl"
    a.mkdir(); b.mkdir()
    args = ("fluxAtObserverFast::sweepDetectorTraceOnce", "folder=out", "srcZ=-75", "dirY=0", "thetaMax=170")
    _run(cli, a, *args, rays=100000, seed=5)
    env = dict(os.environ, ISX_QUIET="1", ISX_RAYS="100000", ISX_SEED="5", ISX_FORCE_COMM="1", ISX_RENDEZVOUS=str(b))
    r = subprocess.run([cli, *args], cwd=b, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    name = "fluxmap_traceonce_100000rays_180x90_src-60_0_-75.csv"
    assert rows(a / "out" / name) == rows(b / "out" / name) and len(rows(a / "out" / name)) == 16201
    assert not [d for d in os.listdir(b) if d.startswith("isx_rdzv_")]   # the job's rendezvous directory is gone once everyone has joined
    c = tmp_path / "dup"
    c.mkdir()
    procs = []
    for rank in (0, 1):
        env = dict(os.environ, ISX_QUIET="1", ISX_RAYS="20000", ISX_RANK=str(rank), ISX_WORLD="2", ISX_DEVICE="0",
                   ISX_RENDEZVOUS=str(c), ISX_JOB_ID="dup-%d" % os.getpid())
        procs.append(subprocess.Popen([cli, *args], cwd=c, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (_, err) in zip(procs, outs):
        assert p.returncode != 0 and "isx_comm" in err


@pytest.mark.gpu
def test_series_entry_point(cli, isx, tmp_path):
    """sweepSeries() of fluxAtObserverFast.C:1641: five trace-once maps at port 164 in portAngleSweep_04_03_-60_0_-75_164/."""
    _run(cli, tmp_path, "fluxAtObserverFast::sweepSeries", rays=20000, seed=99)
    folder = tmp_path / "portAngleSweep_04_03_-60_0_-75_164"
    names = sorted(os.listdir(folder))
    assert names == ["fluxmap_traceonce_20000rays_180x90_src-60_0_-75.csv"] + [
        f"fluxmap_traceonce_20000rays_180x90_src-60_0_-75_{k}.csv" for k in range(1, 5)]
    c = isx.default_config(); c.theta_max_deg = 164.0
    for k, name in enumerate(["fluxmap_traceonce_20000rays_180x90_src-60_0_-75.csv"] + [
            f"fluxmap_traceonce_20000rays_180x90_src-60_0_-75_{k}.csv" for k in range(1, 5)]):
        meta, header, rows = parse_fluxmap(folder / name)
        hits, st = isx.fluxmap(c, 20000, 99, k * 20000)
        assert meta["Exit port angle"] == "164 degrees"
        assert meta["Total rays exiting port"] == f"{st.counted_below_z} out of 20000"
        assert np.array_equal(rows[:, 2], np.array([float(f"{h / 20000.0:.6f}") for h in hits.reshape(-1)]))


def test_bench_workload_does_not_depend_on_the_number_of_gpus():
    """the points of a scaling curve are the same per-GPU workload (BASELINE configs[1]: 5e7 rays per GPU per
    step), so the N = 1 point equals the single-GPU bench; configs[4] is an extra loop at N = 8, not a different default.  And
    `bench.py --gpus N` without a launcher stops before it touches a GPU instead of measuring one GPU under another label."""
    import importlib.util
    import subprocess
    import sys
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert {bench.default_rays_per_gpu(w) for w in (1, 2, 4, 8)} == {50_000_000}
    assert bench.REFERENCE_PUBLISHED["value"] == pytest.approx(810_000_000 / 12523.937080 / 1e6, rel=1e-3)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "torch.distributed.run" in (r.stderr + r.stdout) and r.stdout.strip() == ""
