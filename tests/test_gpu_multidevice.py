"""The first REAL multi-rank RCCL runs, automatic on any box with two or more GPUs (VERDICT r04 'next' 6).  On the one-GPU test
box they skip -- RCCL has never met a second device in this project, and DESIGN.md section 6 says so -- on a multi-GPU box they are
the first evidence: (1) bench.py through the driver's own N = 2 launch line over RCCL: the all-reduced histogram of the last step
== the two slices traced in this process; (2) the native C++ driver (isx_macro, host/isx_comm.cpp: one ncclAllReduce per call) with
two ranks on two devices: its CSV == the one-rank CSV apart from the timestamp / timing lines."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 0x5EED0001


def _device_count():
    """Visible HIP devices WITHOUT initialising the GPU in this process (torch.cuda.device_count() only counts: see the task's
    notes on exec after GPU initialisation; the tests below start children)."""
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


needs_two = pytest.mark.skipif(_device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device (the one-GPU box "
                                                          "rehearses this path over gloo: tests/test_gpu_round2.py)")


@needs_two
def test_bench_two_ranks_over_rccl(isx):
    rays, steps, warmup = 2_000_000, 2, 1
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "ISX_FORCE_DIST", "ISX_BENCH_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29547", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", str(steps),
                        "--warmup", str(warmup), "--rays", str(rays), "--cpu-rays", "0"], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert out["config"]["rccl_world_size"] == 2 and out["config"]["torch_backend"] == "nccl" and out["config"]["reduce_path"] == "device"
    s_last = warmup + steps - 1
    total = 0
    for rank in range(2):
        first, count = isx.step_slice(s_last, rank, 2, rays)
        h, _ = isx.fluxmap(isx.default_config(), count, SEED, first)
        total += int(h.sum())
    assert out["hist_sum_last_step"] == total
    assert abs(out["value"] - 2 * rays / (out["ms_per_step"] * 1e3)) < 1e-6 * out["value"]
    lo, hi = out["per_rank_ms_min_max"]["allreduce_ms"]
    assert 0 < lo <= hi


@needs_two
def test_native_driver_two_ranks_two_devices(tmp_path):
    cli = os.path.join(ROOT, "altair-raytracing_amd", "host", "isx_macro")
    assert os.path.exists(cli)
    args = ("fluxAtObserverFast::sweepDetectorTraceOnce", "folder=out", "srcZ=-75", "dirY=0", "thetaMax=170")
    name = "fluxmap_traceonce_400000rays_180x90_src-60_0_-75.csv"

    def rows(path):   # everything but the lines that carry a date or a time
        return [ln for ln in open(path).read().splitlines()
                if not any(k in ln for k in ("Generated", "completed at", "time:", "execution time"))]

    one = tmp_path / "one"
    one.mkdir()
    env = dict(os.environ, ISX_QUIET="1", ISX_RAYS="400000", ISX_SEED="5")
    r = subprocess.run([cli, *args], cwd=one, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    two = tmp_path / "two"
    two.mkdir()
    procs = []
    for rank in (0, 1):
        env = dict(os.environ, ISX_QUIET="1", ISX_RAYS="400000", ISX_SEED="5", ISX_RANK=str(rank), ISX_WORLD="2", ISX_DEVICE=str(rank),
                   ISX_RENDEZVOUS=str(two), ISX_JOB_ID="two-%d" % os.getpid(), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([cli, *args], cwd=two, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (_, err) in zip(procs, outs):
        assert p.returncode == 0, err[-2000:]
    a, b = rows(one / "out" / name), rows(two / "out" / name)
    assert len(a) > 16200 and a == b
