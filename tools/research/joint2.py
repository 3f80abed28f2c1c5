"""RESEARCH: universal emission weight as a low-order polynomial in mu=cos(psi), fitted jointly to the 7 maps."""
import sys, os, numpy as np, hyp
np.set_printoptions(linewidth=250, precision=3, suppress=True)
N=int(float(sys.argv[1]))
names=['pp_03_31_0','pp_03_31_1','pp_03_31_2','pp_04_1_0','pp_04_1_1','pp_04_1_2','pp_04_1_3']
R=hyp.ref_maps()
As=[];bs=[]
for name in names:
    info,ref=R[name]
    br=np.load(f'/tmp/psi_{name}_{N}.npz')['br'].astype(float)/N
    M=np.vstack([br[:36],br[38:39]])
    rp=ref.sum(1)/info["rays_per_position"]; sig=np.sqrt(np.maximum(ref.sum(1),1))/info["rays_per_position"]
    As.append(M.T/sig[:,None]); bs.append(rp/sig)
psi=np.radians((np.arange(36)+0.5)*2.5); mu=np.cos(psi)
for order in (1,2,3,4,6):
    # w(psi) = sum_k c_k mu^k ; rim weight separate
    B=np.zeros((37,order+2))
    for k in range(order+1): B[:36,k]=mu**k
    B[36,order+1]=1
    A=np.vstack([a@B for a in As]); b=np.concatenate(bs)
    c=np.linalg.lstsq(A,b,rcond=None)[0]
    w=B@c
    chis=[((a@w-bb)**2).mean() for a,bb in zip(As,bs)]
    print("order",order,"chi2/row per map",np.round(chis,2),"rim w",round(w[36],3))
    print("   w(psi)-1 % at psi=1.25,6.25,..:",(w[:36:2]-1)*100)
