import sys, numpy as np, hyp
np.set_printoptions(linewidth=220, precision=2, suppress=True)
d=np.loadtxt('/root/reference/3dRayLog.txt')
print(d.shape, np.abs(np.linalg.norm(d,axis=1)-1).max(), (d[:,2]<0).mean())
ad=np.loadtxt('/root/reference/angular_dist.txt')
N=int(float(sys.argv[1]))
for spec in sys.argv[2:]:
    kw={}
    if spec!="base":
        for kv in spec.split(","):
            k,v=kv.split("="); kw[k]=float(v) if ("." in v or "e" in v) else int(v)
    c=hyp.default_cfg(rho=1.0,sigma=0.0,box_half=200.0,src=[-60,0,-80],max_points=10000,n_theta=2,n_phi=2,**kw)
    h,st,dz,rad=hyp.run(c,N,3)
    ours=dz[:50].astype(float); ours/=ours.sum()
    hl=np.histogram(d[:,2],bins=100,range=(-1,1))[0][:50].astype(float); nl=hl.sum()
    ha=ad[:50,1]; na=ha.sum()
    # 10 groups of 5 bins
    g=lambda x: x.reshape(10,5).sum(1)
    print(spec, "exit", st.counted/N, "susp", st.suspended/N)
    print(" log/ours-1 %", (g(hl)/nl/g(ours)-1)*100, " sigma%", 100/np.sqrt(g(hl)))
    print(" angdist/ours-1 %", (g(ha)/na/g(ours)-1)*100)
    print(" chi2 log", (((hl-ours*nl)**2)/(ours*nl)).sum()/49, "chi2 angdist", (((ha-ours*na)**2)/(ours*na)).sum()/49)
    # azimuth of log directions
ph=np.degrees(np.arctan2(d[:,1],d[:,0]))%360
print("phi hist of log (12 bins)", np.histogram(ph,bins=12,range=(0,360))[0])
