"""Does the reference map carry the AZIMUTHAL signature of a specular first interaction?  108 sector sums (9 x 10 deg in theta, 12 x 30 deg
in phi) of the 170-degree map against the base run of hyp.c, theta-only part divided out, fitted with the phi-structure of the
`first_specular` variant (up to -36 % / +18 % per sector).  CPU, ~2 min.   python tools/research/phifit.py"""
import sys, os, json, numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import hyp
N=20_000_000
maps=hyp.ref_maps()
info,ref=maps['pp_03_31_0']; ref=ref.reshape(180,90).astype(float); n=info['rays_per_position']
hb,_,_,_=hyp.run(hyp.default_cfg(),N,11); hf,_,_,_=hyp.run(hyp.default_cfg(first_specular=1),N,11)
hb=hb.astype(float); hf=hf.astype(float)
def sect(m): return np.array([[m[a*20:(a+1)*20, b*15//2:(b+1)*15//2].sum() for b in range(12)] for a in range(9)])
Sr=sect(ref)/n; Sb=sect(hb)/N; Sf=sect(hf)/N
Rr=Sr/Sb; Rf=Sf/Sb
# remove the theta-only part (row means weighted)
Rr_phi=Rr/ (Sr.sum(1)/Sb.sum(1))[:,None]-1
Rf_phi=Rf/ (Sf.sum(1)/Sb.sum(1))[:,None]-1
sig=1/np.sqrt(sect(ref))   # relative sigma of reference sector sums
w=1/sig**2
a=(w*Rr_phi*Rf_phi).sum()/(w*Rf_phi**2).sum()
err=1/np.sqrt((w*Rf_phi**2).sum())
print("phi-structure amplitude of 'first interaction specular' in the reference map (170 deg): a = %.3f +- %.3f"%(a,err))
print("largest phi-structure of the variant (per cent):", np.round(100*Rf_phi[np.unravel_index(np.argmax(np.abs(Rf_phi)),Rf_phi.shape)],1), "at sector", np.unravel_index(np.argmax(np.abs(Rf_phi)),Rf_phi.shape))
print("chi2 of reference phi-structure alone:", float((w*Rr_phi**2).sum()), "over 108 sectors")
np.set_printoptions(linewidth=200, precision=1, suppress=True)
print("variant phi-structure (%) rows=theta 10deg bands, cols=phi 30deg sectors:\n", 100*Rf_phi)
print("reference phi-structure (%):\n", 100*Rr_phi)
