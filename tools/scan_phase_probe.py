import ctypes as C, numpy as np, torch, sys
sys.path.insert(0,'.')
from text2loc_amd import synth
from text2loc_amd.engine import Engine
db,q,_=synth.make_retrieval_problem(11259,4096,seed=1)
for mode,var,ns in ((3,80,8),):
    e=Engine(0); e.set_option("search_mode",3); e.set_option("scan_variant",var); e.set_option("search_nsplit",ns)
    e.db_set(torch.from_numpy(db).cuda()); dq=torch.from_numpy(q).cuda()
    for _ in range(3): e.search(dq,10)
    torch.cuda.synchronize()
    e.lib.t2l_debug_counters.argtypes=[C.c_void_p,C.c_void_p,C.c_int32]
    out=np.zeros(36,dtype=np.int64)
    e.lib.t2l_debug_counters(e._h,out.ctypes.data,36)
    st=out[20:36].reshape(4,4); base=st[:,0].min()
    print(" block stamps (cycles rel. to earliest start) [start, prologue done, loop done, end] for blocks 0,17,mid,last:")
    print(st-base)
    out=out[:20]
    tiles = 352//ns
    print("var",var,"nsplit",ns,"tiles/WG",tiles)
    print(" per-tile cycles [vmcnt wait, barrier, glds issue, first LDS, MFMA+select] per wave:")
    print((out.reshape(4,5)/tiles).round(0))
