"""Lifecycle soak (GPU box): 60 x (isx_init -> a few calls that allocate every pooled buffer, incl. the 2.4 GB pipeline workspace ->
isx_shutdown); free device memory before and after must match (no leak), results identical every cycle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import altair_raytracing_amd as isx
isx.load()
torch.cuda.init(); torch.cuda.synchronize()
free0, total = torch.cuda.mem_get_info()
ref = None
for k in range(60):
    isx.init(0)
    c = isx.default_config()
    h, st = isx.fluxmap(c, 50_000_000 if k % 10 == 0 else 200_000, 3)
    d, _ = isx.exit_dz_hist(c, 100000, 3)
    ids, dirs, cnt, _ = isx.exit_directions(c, 100000, 3)
    pp, _ = isx.fluxmap_per_position(c, 5, 3)
    ca = np.array([[0, 0, -150, 0, 0, 1.0]] * 5)
    ds, _ = isx.disc_sweep(c, ca, 10.0, 0.1, 50000, 3)
    sig = (int(h.sum()) if k % 10 else None, int(d.sum()), cnt, int(pp.sum()), int(ds.sum()))
    if k == 1: ref = sig
    if k > 1 and k % 10: assert sig == ref, (k, sig, ref)
    isx.shutdown()
    if k % 10 == 0:
        torch.cuda.synchronize()
        f = torch.cuda.mem_get_info()[0]
        if k == 0: free_after_first = f      # (the first cycle loads the code objects: ~150 MiB that stay)
        print(k, "free MiB", f >> 20, flush=True)
torch.cuda.synchronize()
free1, _ = torch.cuda.mem_get_info()
print("free before anything", free0 >> 20, "MiB, after the first cycle", free_after_first >> 20, "MiB, after the last", free1 >> 20, "MiB")
sys.exit(0 if abs(free_after_first - free1) < (16 << 20) else 1)
