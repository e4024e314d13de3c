import sys; sys.path.insert(0,"tests"); sys.path.insert(0,".")
from helpers import uvs, abi, synth, lm_reduced_system, unpad
from oracle_binding import Oracle
import numpy as np
np.set_printoptions(linewidth=250, precision=2)
o=Oracle(); s=uvs.api.Solver(max_batch=4); w=synth.make_window(4)
eo=o.evaluate(w, robust=True); ref=lm_reduced_system(w, eo); d=s.debug_first_iteration(w)
S=unpad(d["S"]); S=S+np.tril(S,-1).T; R=ref["S"]
E=np.abs(S-R)
blk=np.zeros((11,11))
for a in range(11):
    for b in range(11):
        sub=E[15*a:15*a+15,15*b:15*b+15]; den=np.abs(R[15*a:15*a+15,15*b:15*b+15]).max()
        blk[a,b]=sub.max()/max(den,1e-300)
print("per-block relative error (block max norm):"); print(blk)
a=np.unravel_index(np.argmax(blk), blk.shape); print("worst block", a)
sub=E[15*a[0]:15*a[0]+15,15*a[1]:15*a[1]+15]/np.abs(R[15*a[0]:15*a[0]+15,15*a[1]:15*a[1]+15]).max()
print((sub>1e-9).astype(int))
for name in ("hd","dd","g"):
    x=unpad(d[name]); r=ref[name][:165]
    e=np.abs(x-r)/np.maximum(np.abs(r),1e-300)
    print(name, "max elementwise rel err", e.max(), "at", int(e.argmax()), "frame", int(e.argmax())//15, "dof", int(e.argmax())%15)
