# dev probe (GPU box): the one-launch small-batch search against the two-launch path, us per call issued from C, over workgroup counts and query counts
set -e
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_search.py -x -q -k "one_launch_exact or ragged_shapes or unrepresentable or streaming_small or exact_ties or near_ties" 2>&1 | tail -15
timeout 600 python - <<'PY'
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import bench
from text2loc_amd import synth
from text2loc_amd.engine import Engine
db, qs, _ = synth.make_retrieval_problem(11259, 4096, 256, seed=1, noise=0.5)
eng = Engine(0)
eng.db_set(torch.from_numpy(db).cuda())
big = torch.from_numpy(qs).cuda()
for _ in range(1200): eng.search(big, 10)
for wgs in (256, 192, 128, 96):
    eng.set_option("search_small_wgs", wgs)
    for qn in (1, 2, 4, 8, 16):
        nb = 1000
        dqm = torch.from_numpy(np.ascontiguousarray(np.tile(qs[:qn][None], (nb, 1, 1)))).cuda()
        om = (torch.empty((nb, qn, 10), dtype=torch.int32, device="cuda"), torch.empty((nb, qn, 10), dtype=torch.float64, device="cuda"))
        for small in (1, 0):
            eng.set_option("search_small", small)
            for _ in range(2): eng.search_many(dqm, 10, out=om)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): eng.search_many(dqm, 10, out=om)
            torch.cuda.synchronize()
            print(f"wgs={wgs} q={qn} small={small}: {(time.perf_counter()-t0)/5000*1e6:.2f} us per call from C")
        eng.set_option("search_small", 1)
eng.set_option("search_small_wgs", 0)
eng.set_option("profile_events", 1)
dq1 = torch.from_numpy(qs[:1]).cuda()
eng.kernel_stats("search_small")
for _ in range(50): eng.search(dq1, 10)
torch.cuda.synchronize()
print("kernel ms, n:", eng.kernel_stats("search_small"))
PY
