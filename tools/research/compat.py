import sys, numpy as np, hyp
np.set_printoptions(linewidth=220, precision=2, suppress=True)
N=int(float(sys.argv[1]))
R=hyp.ref_maps()
def bands(h): return h.reshape(12,15,-1).sum((1,2)).astype(float)
for name in ("to_170","to_164","to_160"):
    info,ref=R[name]
    nref=info["n_files"]*info["rays_per_file"]
    for spec in sys.argv[2:]:
        kw={}
        if spec!="base":
            for kv in spec.split(","):
                k,v=kv.split("="); kw[k]=float(v) if ("." in v or "e" in v) else int(v)
        h,st,_,_=hyp.run(hyp.default_cfg(theta_max_deg=info["port_deg"],hit_line=1,**kw),N,7)
        r=(bands(h)/N)/(bands(ref)/nref)-1
        # per-ray noise estimate for reference band sums: use per-file totals scatter for the total
        pf=np.array(info["per_file_total_hits"]); 
        print(f"{name} {spec:24s} exit ours {st.counted/N:.5f} ref {info['exited_sum']/nref:.5f} (+-{np.sqrt(.25/nref):.5f}) total ratio {h.sum()/N/(ref.sum()/nref):.4f} (ref file scatter {pf.std(ddof=1)/pf.mean()/np.sqrt(len(pf)):.4f}) bands% {r*100}")
