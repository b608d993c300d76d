import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from text2loc_amd import synth
from text2loc_amd.engine import Engine
db, qs, _ = synth.make_retrieval_problem(11259, 4096, seed=1, noise=0.5)
e = Engine(0); e.db_set(torch.from_numpy(db).cuda())
for qn in (1, 2, 8, 64, 256):
    dq = torch.from_numpy(np.ascontiguousarray(qs[:qn])).cuda()
    o = (torch.empty((qn, 10), dtype=torch.int32, device="cuda"), torch.empty((qn, 10), dtype=torch.float64, device="cuda"))
    for _ in range(2000): e.search(dq, 10, out=o)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(1000): e.search(dq, 10, out=o)
    torch.cuda.synchronize(); print(qn, "us/call", (time.perf_counter() - t0) / 1000 * 1e6, e.search_counters())
