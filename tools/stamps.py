# dev tool: build libt2l_stamps.so with -DT2L_STAMPS (make -C text2loc_amd/csrc stamps) and print the in-kernel phase times of
# the scan (s_memrealtime, 100 MHz) for the bench workload
import ctypes as C, sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2loc_amd import engine as E, synth
E._LIB_PATH = os.path.join(os.path.dirname(E._LIB_PATH), "libt2l_stamps.so")
eng = E.Engine(0)
db, qs, _ = synth.make_retrieval_problem(11259, 4096, seed=1, noise=0.5)
eng.db_set(torch.from_numpy(db).cuda())
dq = torch.from_numpy(qs).cuda()
for a in sys.argv[1:]:
    k, v = a.split("=")
    eng.set_option(k, float(v))
out = (C.c_longlong * 8)()
eng.lib.t2l_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
for _ in range(20):
    eng.search(dq, 10)
for rep in range(3):
    eng.lib.t2l_debug_stamps(eng._h, out)  # arms the min/max slots
    eng.search(dq, 10)
    eng.lib.t2l_debug_stamps(eng._h, out)
    t = [out[i] for i in range(4)]
print("prologue %.2f us, loop %.2f us, epilogue %.2f us (one workgroup, 100 MHz clock)" % ((t[1]-t[0])/100, (t[2]-t[1])/100, (t[3]-t[2])/100))
print("workgroup starts spread over %.2f us; first start -> first end %.2f us, -> last end %.2f us" % ((out[5]-out[4])/100, (out[6]-out[4])/100, (out[7]-out[4])/100))
