import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from text2loc_amd import synth
from text2loc_amd.engine import Engine
N, Q = 11259, 4096
db, qs, _ = synth.make_retrieval_problem(N, Q, 256, seed=1, noise=0.5)
sd = synth.make_object_branch_weights(0)
cells = synth.make_cells(N, seed=0)
packed = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in cells.items() if k != "counts"}
dq = torch.from_numpy(qs).cuda()
e = Engine(0)
e.load_weights(sd, class_embed=True, color_embed=True)
def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
a = torch.randn(4096, 4096, device="cuda")
for _ in range(30): a @ a
emb = e.encode_cells(packed)
print("encode_cells ms", t(lambda: e.encode_cells(packed)))
print("db_set ms", t(lambda: e.db_set(emb)))
print("search ms", t(lambda: e.search(dq, 10)))
print("db_set+search ms", t(lambda: (e.db_set(emb), e.search(dq, 10))))
print("all ms", t(lambda: (e.db_set(e.encode_cells(packed)), e.search(dq, 10))))
e.set_option("profile_events", 1)
for _ in range(3): e.db_set(emb); e.search(dq, 10)
torch.cuda.synchronize()
for k in ("search_scan", "search_rerank", "search_exact", "search_fallback"):
    print(k, e.kernel_stats(k))
print(e.search_counters())
