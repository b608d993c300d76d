# dev tool: build libt2l_stamps.so with -DT2L_STAMPS (make -C text2loc_amd/csrc stamps) and print the in-kernel phase times of
# the scan (s_memrealtime, 100 MHz) for the bench workload
import ctypes as C, sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2loc_amd import engine as E, synth
E._LIB_PATH = os.path.join(os.path.dirname(E._LIB_PATH), "libt2l_stamps.so")
eng = E.Engine(0)
QN = int(os.environ.get("T2L_Q", "4096"))
db, qs, _ = synth.make_retrieval_problem(11259, QN, seed=1, noise=0.5)
eng.db_set(torch.from_numpy(db).cuda())
dq = torch.from_numpy(qs).cuda()
for a in sys.argv[1:]:
    k, v = a.split("=")
    eng.set_option(k, float(v))
out = (C.c_longlong * 16)()
eng.lib.t2l_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
for _ in range(1500):
    eng.search(dq, 10)
for rep in range(3):
    for _ in range(10):
        eng.search(dq, 10)  # steady state: back-to-back calls
    eng.lib.t2l_debug_stamps(eng._h, out)  # arms the min/max slots (synchronises)
    for _ in range(4):
        eng.search(dq, 10)
    eng.lib.t2l_debug_stamps(eng._h, out)  # block 37's stamps are those of the LAST call; min/max span the 4 calls
    t = [out[i] for i in range(4)]
    u = [out[8 + i] for i in range(6)]
    print("  inside the prologue (us from kernel start): loads0 landed %.2f, exchange0 written+barrier %.2f, frags0 read %.2f, loads1 landed %.2f, barrier1 %.2f, frags1 read %.2f" % tuple((x - t[0]) / 100 for x in u))
    print("prologue %.2f us, loop %.2f us, epilogue %.2f us (one workgroup of the last of 4 back-to-back calls, 100 MHz clock)" % ((t[1]-t[0])/100, (t[2]-t[1])/100, (t[3]-t[2])/100))
