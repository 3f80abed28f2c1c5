"""Soak of extreme grid shapes (1x1 ... 200x180, single rows / columns; grids whose tables do not fit the LDS are refused with
ISX_ERR_BAD_CONFIG): culled == brute through the pipeline and the fused kernels.  GPU box:  python tools/soak_shapes.py"""
import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import altair_raytracing_amd as isx
isx.load(); isx.init(0)
rng = np.random.default_rng(8)
shapes = [(36000,1),(18000,2),(12000,3),(9000,4),(33000,1),(32768,1),(32767,1),(1,1),(1,2),(2,1),(1,900),(2,900),(3,1200),(40,900),(20,1800),(10,3600),(7,5000),(1,16000),(180,200),(200,180),(179,201),(64,64),(65,63),(129,127)]
bad=0; ran=0
for k,(nt,nph) in enumerate(shapes):
    for rep in range(3):
        c = isx.default_config()
        c.n_theta, c.n_phi = nt, nph
        c.det_distance = float(rng.choice([30.0,100.0,180.0])); c.det_diameter = float(c.det_distance*2*rng.choice([0.02,0.2,0.6,1.2]))
        c.theta_max_deg = float(rng.uniform(155,176))
        if rep==1: c.source_model=1
        if rep==2: c.trace_mode=1
        n=50000
        try:
            isx.set_option("bin_mode",0); brute,sb=isx.fluxmap(c,n,40+k)
        except isx.IsxError as e:
            isx.set_option("bin_mode",1); print((nt,nph),"refused:",e.status); break
        isx.set_option("bin_mode",1)
        culled,sc=isx.fluxmap(c,n,40+k)
        isx.set_option("pipeline",0); fused,sf=isx.fluxmap(c,n,40+k); isx.set_option("pipeline",1)
        ok=np.array_equal(brute,culled) and np.array_equal(brute,fused)
        ran+=1; bad+=(not ok)
        if not ok: print("MISMATCH",(nt,nph),rep,c.det_diameter,c.det_distance,flush=True)
print("done: ran",ran,"mismatches",bad)
