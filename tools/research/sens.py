"""Sensitivity (RESEARCH): band pattern of variant/base - 1 in percent, same seed (detector-side changes see the same rays)."""
import sys, numpy as np, hyp
N=int(float(sys.argv[1])); port=float(sys.argv[2])
np.set_printoptions(linewidth=220, precision=2, suppress=True)
def bands(h): return h.reshape(12,15,-1).sum((1,2)).astype(float)
def parse(spec):
    kw={}
    for kv in spec.split(","):
        k,v=kv.split("="); kw[k]=float(v) if ("." in v or "e" in v) else int(v)
    return kw
h0,st0,_,_=hyp.run(hyp.default_cfg(theta_max_deg=port),N,1); b0=bands(h0)
print("base total", h0.sum()/N, "exit", st0.counted/N)
for s in sys.argv[3:]:
    h,st,_,_=hyp.run(hyp.default_cfg(theta_max_deg=port,**parse(s)),N,1)
    print(f"{s:32s} total {h.sum()/h0.sum()-1:+.4f} exit {st.counted/st0.counted-1:+.4f} bands% {(bands(h)/b0-1)*100}", flush=True)
