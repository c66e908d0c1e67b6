import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
import raptor_amd.l2f as l2f
from oracle import oracle as O
from test_gpu_parity import World, _well_conditioned
w8=np.fromfile('raptor_amd/data/raptor_policy.bin','<f4')
dev=l2f.Device(0)
w=World(dev,O,512,seed=11,domain_randomization=1)
w.sync_oracle_to_gpu_state()
w.vector.rollout(dev,w.env,w.params,w.state,w.policy,w.rng,500,'fused',False)
Sp,stp=_well_conditioned(w,w8,500,0)
O.rollout(w.cfg,w8,11,0,0,w.P,w.S,w.H,500,0,w.st,8)
S=w.state.numpy()
g_cnt,g_len,g_term=w.env.finished_counts(),w.env.finished_lengths(),w.env.finished_terminated()
same=(g_cnt==w.st.fin_counts)&(g_len==w.st.fin_lengths)&(g_term==w.st.fin_terminated)
ins=(np.abs(Sp[:,:13]-w.S[:,:13]).max(1)<1e-5)&(stp.fin_counts==w.st.fin_counts)&(stp.fin_lengths==w.st.fin_lengths)
print('same',same.mean(),'ins',ins.mean(),'same|ins',same[ins].mean())
sel=ins&same
d=np.abs(S[sel,:13]-w.S[sel,:13]).max(1)
print('d quant',np.quantile(d,[0.5,0.9,0.99,1.0]))
all_d=np.abs(S[:,:13]-w.S[:,:13]).max(1); ref_d=np.abs(Sp[:,:13]-w.S[:,:13]).max(1)
print('all med',np.nanmedian(all_d),(all_d>1e-2).mean(),(ref_d>1e-2).mean())
print('returns close', np.abs(w.env.finished_returns()[sel]-w.st.fin_returns[sel]).max())
