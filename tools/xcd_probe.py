"""dev probe: step time of the headline search under the XCD rectangle mappings (option search_xcd_qgroups)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2loc_amd import synth
from text2loc_amd.engine import Engine

N, Q, K = 11259, 4096, 10
db, qs, _ = synth.make_retrieval_problem(N, Q, 256, seed=1, noise=0.5)
batches = [qs] + [synth.make_queries_for(db, Q, seed=100 + i, noise=0.5)[0] for i in range(1, 4)]
eng = Engine(0)
eng.db_set(torch.from_numpy(db).cuda())
dq = [torch.from_numpy(np.ascontiguousarray(b)).cuda() for b in batches]
outs = [(torch.empty((Q, K), dtype=torch.int32, device="cuda"), torch.empty((Q, K), dtype=torch.float64, device="cuda")) for _ in range(12)]
ref = None
for gq, lanes in ((1, 1), (4, 1), (2, 1), (8, 1), (4, 3), (1, 3), (4, 1)):
    eng.set_option("search_xcd_qgroups", gq)
    eng.set_option("search_lanes", lanes)
    for i in range(1500):
        eng.search(dq[i % 4], K, out=outs[i % 12], join=lanes == 1)
    eng.search_join()
    eng.set_option("stats_reset", 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 804
    for i in range(n):
        eng.search(dq[i % 4], K, out=outs[i % 12], join=lanes == 1)
    eng.search_join()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    span = eng.kernel_stats("search_scan_span")[0]
    busy = eng.kernel_stats("search_scan_busy")[0]
    eng.set_option("search_lanes", 1)
    idx = eng.search(dq[0], K)[0].clone()
    if ref is None:
        ref = idx
    print(f"xcd_qgroups={gq} lanes={lanes}: {dt * 1e6:.2f} us/step, scan span {span * 1e3:.2f} us, busy {busy * 1e3:.2f} us, ids equal {bool(torch.equal(idx, ref))}")
