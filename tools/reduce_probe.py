import numpy as np, torch, sys
sys.path.insert(0,'.')
from text2loc_amd import synth
from text2loc_amd.engine import Engine
e=Engine(0); e.set_option("profile_events",1)
for name,n_pts in (("uniform2000", np.full(14000,2000,dtype=np.int64)), ("uniform200", np.full(140000,200,dtype=np.int64)), ("lognormal", np.clip(np.round(np.random.default_rng(11).lognormal(6.98,1.0,size=16000)),25,60000).astype(np.int64))):
    poff=np.zeros(len(n_pts)+1,dtype=np.int64); np.cumsum(n_pts,out=poff[1:]); total=int(poff[-1])
    x=torch.rand((total,3),device="cuda"); r=torch.rand((total,3),device="cuda"); o=poff
    for _ in range(2): e.reduce_objects(x,r,o,synth.COLORS,np.arange(8,dtype=np.int32))
    e.kernel_stats("reduce_objects")
    for _ in range(5): e.reduce_objects(x,r,o,synth.COLORS,np.arange(8,dtype=np.int32))
    torch.cuda.synchronize(); ms,n=e.kernel_stats("reduce_objects")
    print(name, total, round(ms*1e3,1),"us", round(total*24/ms/1e6,1),"GB/s")
