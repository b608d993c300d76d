"""dev: the paired scan with the per-score insertion (search_tile_sel = 0) against the tile-local top-3 selection (= 1) on the headline
workload (Q = 4096, N = 11,259, K = 10): ids / scores must be identical (both are exact by contract), step time, scan / re-rank kernel
time (events around sampled launches, span stamps), certificate counters. A/B alternated on one box."""
import sys
import time

import numpy as np
import torch

from text2loc_amd import synth
from text2loc_amd.engine import Engine

N, Q = 11259, 4096
db, qs, _ = synth.make_retrieval_problem(N, Q, seed=0)
eng = Engine(0)
eng.db_set(torch.from_numpy(db).cuda())
qd = torch.from_numpy(qs).cuda()
a = torch.randn(4096, 4096, device="cuda")
res = {}
for rnd in range(3):
    for sel in (0, 1):
        eng.set_option("search_tile_sel", sel)
        for _ in range(60):
            a @ a  # clock ramp
        torch.cuda.synchronize()
        for _ in range(50):
            idx, sc = eng.search(qd, 10)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(400):
            eng.search(qd, 10)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 400
        eng.set_option("profile_events", 1)
        eng.set_option("profile_rerank", 1)
        for nme in ("search_scan", "search_rerank", "search_scan_span", "search_scan_busy"):
            eng.kernel_stats(nme)
        for _ in range(200):
            idx, sc = eng.search(qd, 10)
        torch.cuda.synchronize()
        ks = {nme: round(eng.kernel_stats(nme)[0] * 1e3, 2) for nme in ("search_scan", "search_rerank", "search_scan_span", "search_scan_busy")}
        eng.set_option("profile_events", 0)
        eng.set_option("profile_rerank", 0)
        res[sel] = (idx.cpu().numpy().copy(), sc.cpu().numpy().copy())
        print("tile_sel", sel, "us/step %.2f" % (dt * 1e6), ks, "counters", eng.search_counters(), flush=True)
    print("ids equal:", bool(np.array_equal(res[0][0], res[1][0])), "scores equal:", bool(np.array_equal(res[0][1], res[1][1])), flush=True)
# a clustered database (neighbouring rows similar: what overlapping KITTI360Pose cells look like): certificate failures per mode
rng = np.random.default_rng(5)
base = rng.standard_normal((N // 4 + 1, 256)).astype(np.float32)
dbc = np.repeat(base, 4, axis=0)[:N] + 0.15 * rng.standard_normal((N, 256)).astype(np.float32)
dbc /= np.linalg.norm(dbc, axis=1, keepdims=True)
qc = dbc[rng.integers(0, N, Q)] + 0.05 * rng.standard_normal((Q, 256)).astype(np.float32)
qc /= np.linalg.norm(qc, axis=1, keepdims=True)
eng.db_set(torch.from_numpy(dbc).cuda())
qd = torch.from_numpy(qc).cuda()
for sel in (0, 1):
    eng.set_option("search_tile_sel", sel)
    for _ in range(20):
        idx, sc = eng.search(qd, 10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        eng.search(qd, 10)
    torch.cuda.synchronize()
    res[sel] = idx.cpu().numpy().copy()
    print("clustered(runs of 4): tile_sel", sel, "us/step %.2f" % ((time.perf_counter() - t0) / 100 * 1e6), eng.search_counters(), flush=True)
print("clustered ids equal:", bool(np.array_equal(res[0], res[1])))
