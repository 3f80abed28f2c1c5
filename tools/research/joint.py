"""RESEARCH: one common emission-angle weight w(psi) (2.5 deg classes, + rim class) fitted to ALL reference maps jointly."""
import sys, os, numpy as np, hyp
np.set_printoptions(linewidth=250, precision=2, suppress=True)
N=int(float(sys.argv[1])); lam=float(sys.argv[2])
names=['pp_03_31_0','pp_03_31_1','pp_03_31_2','pp_04_1_0','pp_04_1_1','pp_04_1_2','pp_04_1_3']
R=hyp.ref_maps()
As=[];bs=[]
for name in names:
    info,ref=R[name]
    f=f'/tmp/psi_{name}_{N}.npz'
    if not os.path.exists(f):
        c=hyp.default_cfg(theta_max_deg=info["port_deg"],dir=info["source_direction"],two_sided=1)
        h,st,dz,rad,ba,br=hyp.run(c,N,21,split=True); np.savez(f,br=br)
    br=np.load(f)['br'].astype(float)/N
    M=np.vstack([br[:36],br[38:39]])      # 36 psi classes + rim
    rp=ref.sum(1)/info["rays_per_position"]; sig=np.sqrt(np.maximum(ref.sum(1),1))/info["rays_per_position"]
    As.append(M.T/sig[:,None]); bs.append(rp/sig)
k=37
D=np.zeros((k-3,k))
for i in range(k-3): D[i,i:i+3]=[1,-2,1]     # smooth over the psi classes only (not the rim)
def fit(idx):
    A=np.vstack([As[i] for i in idx]+[lam*D]); b=np.concatenate([bs[i] for i in idx]+[np.zeros(k-3)])
    w=np.linalg.lstsq(A,b,rcond=None)[0]; return w
w=fit(range(7))
print("joint w-1 % (psi 2.5deg classes, last=rim):"); print((w-1)*100)
for i,name in enumerate(names):
    wi=fit([i])
    print(name,"chi2/row: base %.2f joint %.2f own %.2f | total ratio base %.4f joint %.4f"%(((As[i]@np.ones(k)-bs[i])**2).mean(),((As[i]@w-bs[i])**2).mean(),((As[i]@wi-bs[i])**2).mean(), (As[i]@np.ones(k)*1).sum()/bs[i].sum() if False else 0, 0))
np.save('/tmp/joint_w.npy',w)
