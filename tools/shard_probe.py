"""dev: kernel times of the per-rank pieces of an 8-GPU row-sharded step on one GPU (scan, re-rank, merge)"""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from text2loc_amd import synth
from text2loc_amd.engine import Engine
P, Q, K, N = 8, 4096, 10, 11259
db, qs, _ = synth.make_retrieval_problem(N, Q, seed=1, noise=0.5)
e = Engine(0)
n_shard = (N + P - 1) // P
dq = torch.from_numpy(qs).cuda()
buf, idx, sc, bb, so = e.result_block(Q, K, "cuda", parts=P)
for r in range(P):
    e.db_set(torch.from_numpy(db[r * n_shard:(r + 1) * n_shard]).cuda(), r * n_shard)
    i, s = e.search(dq, K)
    buf[r, :Q * K * 4].view(torch.int32).view(Q, K).copy_(i)
    buf[r, so:so + Q * K * 8].view(torch.float64).view(Q, K).copy_(s)
e.set_option("profile_events", 1)
for _ in range(300):
    mi, ms = e.merge_gathered(buf.view(-1), bb, so, P, Q, K)
torch.cuda.synchronize()
print("merge kernel us", e.kernel_stats("merge")[0] * 1e3)
ri, _ = __import__("oracle.c_oracle", fromlist=["x"]).retrieve_topk(db, qs[:256], K)
print("merged == unsharded oracle:", bool(np.array_equal(mi[:256].cpu().numpy().astype(np.int64), ri)))
