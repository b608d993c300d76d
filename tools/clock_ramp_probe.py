"""Dev probe (GPU box): per-step time of the stream-ordered search loop over the first 300 ms after an idle period — how long the
MI355X takes to reach its sustained clocks (56.6 us per step in the first 5 ms, 46.8 from 40 ms on)."""
import sys, time, os, numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import text2loc_amd.engine as E
from text2loc_amd import synth
N, Q, K = 11259, 4096, 10
db, qs0, target = synth.make_retrieval_problem(N, Q, 256, seed=1, noise=0.5)
eng = E.Engine(0)
eng.db_set(torch.from_numpy(db).cuda())
q = torch.from_numpy(qs0).cuda()
torch.cuda.synchronize()
time.sleep(0.5)
out = []
T0 = time.perf_counter()
for blk in range(60):
    t0 = time.perf_counter()
    for i in range(100): eng.search(q, K)
    torch.cuda.synchronize()
    out.append((time.perf_counter() - T0, (time.perf_counter()-t0)/100*1e6))
print(" ".join(f"{t*1e3:.0f}ms:{u:.1f}" for t, u in out))
