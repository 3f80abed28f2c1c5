import numpy as np, hyp
np.set_printoptions(linewidth=250, precision=2, suppress=True)
d=np.loadtxt('/root/reference/3dRayLog.txt')
al=np.degrees(np.arccos(np.clip(-d[:,2],-1,1)))
hl=np.histogram(al,bins=18,range=(0,90))[0].astype(float)
# our model expectation in the log's configuration: direction classes = by_alpha summed... use dz hist at fine bins instead
c=hyp.default_cfg(rho=1.0,sigma=0.0,box_half=200.0,src=[-60,0,-80],max_points=10000,n_theta=2,n_phi=2)
import ctypes as C
# need alpha histogram: run with split and n_theta=2 -> ba rows are hit-weighted; instead sample exit dirs via dz hist of 100 bins -> convert
h,st,dz,rad=hyp.run(c,4000000,5)
# dz hist bins: dz in [-1,1] 100 bins -> map to alpha classes by fine resampling: approximate using bin centres is too coarse near axis;
# so rerun quickly by brute force in numpy using cosine law through thin port + our rim?  Simpler: use dz cumulative interpolation.
edges=np.linspace(-1,1,101); cum=np.concatenate([[0],np.cumsum(dz)]).astype(float); cum/=cum[-1]
a_edges=np.arange(0,95,5.0); z_edges=-np.cos(np.radians(a_edges))   # dz = -cos(alpha), increasing
Fc=np.interp(z_edges,edges,cum)
exp=np.diff(Fc)*len(d)
print("log counts  ",hl)
print("model expect",exp)
r=(hl/exp-1)*100; s=100/np.sqrt(exp)
print("log/model-1 %",r); print("sigma %     ",s)
pat=np.array([-3.6,-1.56,.34,1.75,2.4,2.34,1.85,1.12,.29,-.51,-1.08,-1.33,-1.35,-1.34,-1.44,-1.61,-1.79,-1.98])
# amplitude a of pattern: minimize sum ((r - a*pat - b)/s)^2 with free normalisation b
W=1/s**2; X=np.vstack([pat,np.ones(18)]).T
cov=np.linalg.inv(X.T@(W[:,None]*X)); ab=cov@(X.T@(W*r))
print("pattern amplitude a = %.2f +- %.2f (1 = same distortion as in the 170deg per-position map), offset %.2f"%(ab[0],np.sqrt(cov[0,0]),ab[1]))
print("chi2 null",(W*(r-np.average(r,weights=W))**2).sum(),"chi2 with pattern",(W*(r-X@ab)**2).sum(),"dof 17/16")
