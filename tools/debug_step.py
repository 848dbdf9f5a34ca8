import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import drift_step as OS, params as OP
from wheeledlab_amd.core import DriftBatch
n=1024
env=DriftBatch(n,device="cuda:0",seed=5); env.reset(); torch.cuda.synchronize()
p=OP.drift_params()
rng=np.random.RandomState(0)
names=["px","py","pz","qw","qx","qy","qz","vx","vy","vz","wx","wy","wz","wbl","wbr","wfl","wfr","th","om","a0","a1","thf","tlf"]
for k in range(3):
    st=env.state.cpu().numpy().copy(); ep=env.episode_len.cpu().numpy().copy(); st0=st.copy()
    a=rng.uniform(-1.2,1.2,(n,2)).astype(np.float32); a[:,0]=np.abs(a[:,0])
    env.step(torch.from_numpy(a).cuda()); torch.cuda.synchronize()
    OS.step(p,st,ep,env.ref_table.cpu().numpy(),a,5,k)
    got=env.state.cpu().numpy()
    d=np.abs(got[:23,:n]-st[:23,:n])
    print("step",k,"row max diffs:",{names[i]:float(d[i].max()) for i in range(23) if d[i].max()>1e-5})
    worst=np.unravel_index(d.argmax(),d.shape)
    e=worst[1]
    print(" worst env",e,"row",names[worst[0]],"gpu",got[:23,e][worst[0]],"oracle",st[:23,e][worst[0]])
    print(" init state",dict(zip(names,st0[:23,e])),"mu",st0[23:27,e],"act",a[e])
    print(" gpu   ",dict(zip(names,got[:23,e])))
    print(" oracle",dict(zip(names,st[:23,e])))
