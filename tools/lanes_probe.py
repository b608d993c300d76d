"""Dev probe (GPU box): stream-ordered vs pipelined searches (option search_lanes = 1..4, t2l_search_join), results compared
with the stream-ordered ones."""
import sys, time, numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from text2loc_amd.engine import Engine
N, Q, K = 11259, 4096, 10
g = torch.Generator().manual_seed(0)
db = torch.nn.functional.normalize(torch.randn(N, 256, generator=g)).cuda()
qs = [torch.nn.functional.normalize(torch.randn(Q, 256, generator=g)).cuda() for _ in range(4)]
eng = Engine(0)
eng.db_set(db)
ref = [tuple(t.clone() for t in eng.search(q, K)) for q in qs]
torch.cuda.synchronize()
for NL in (1, 2, 3, 4):
    eng.set_option("search_lanes", NL)
    outs = [(torch.empty(Q, K, dtype=torch.int32, device="cuda"), torch.empty(Q, K, dtype=torch.float64, device="cuda")) for _ in range(4)]
    def run(steps):
        for i in range(steps):
            eng.search(qs[i % 4], K, out=outs[i % 4], join=False)
        eng.search_join()
    run(40); torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter(); run(200); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 200)
    ok = all(torch.equal(outs[i][0], ref[i][0]) and torch.equal(outs[i][1], ref[i][1]) for i in range(4))
    print(f"lanes {NL}: {best*1e6:.1f} us/step  {Q/best/1e6:.1f} M q/s  identical={ok}  counters={eng.search_counters()}")
