"""RESEARCH: which re-weighting of exit rays (by direction polar angle alpha, or by crossing radius) would turn the
model's theta-profile into the reference's?  Least squares with smoothness, per port angle."""
import sys, numpy as np, hyp
np.set_printoptions(linewidth=250, precision=2, suppress=True)
N=int(float(sys.argv[1]))
R=hyp.ref_maps()
for name in sys.argv[2:]:
    info,ref=R[name]
    c=hyp.default_cfg(theta_max_deg=info["port_deg"],dir=info["source_direction"],two_sided=1)
    h,st,dz,rad,ba,br=hyp.run(c,N,11,split=True)
    np.savez(f'/tmp/split_{name}.npz',h=h,ba=ba,br=br,N=N)
    rp=ref.sum(1)/info["rays_per_position"]          # reference row sums (fraction units)
    sig=np.sqrt(np.maximum(ref.sum(1),1))/info["rays_per_position"]
    for label,M in (("alpha(5deg)",ba),("psi(5deg;38=rim)",br)):
        M=M.astype(float)/N                            # [class, row]
        use=M.sum(1)>1e-6*M.sum()
        A=(M[use].T)/sig[:,None]; b=rp/sig
        k=use.sum()
        # second-difference smoothness
        D=np.zeros((k-2,k)); 
        for i in range(k-2): D[i,i:i+3]=[1,-2,1]
        for lam in (0.0, 30.0, 300.0):
            Aa=np.vstack([A,lam*D]); bb=np.concatenate([b,np.zeros(k-2)])
            w,res,rk,sv=np.linalg.lstsq(Aa,bb,rcond=None)
            chi=((A@w-b)**2).sum()/len(b)
            print(name,label,"lam",lam,"chi2/row",round(chi,2),"w-1 %",(w-1)*100)
        print("   baseline chi2/row",round(((A@np.ones(k)-b)**2).sum()/len(b),2), "class share %", M[use].sum(1)/M.sum()*100)
