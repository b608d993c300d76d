"""dev: option search_merge_lists — identical results to the plain lists? step time A/B on the headline shape."""
import time

import numpy as np
import torch

from text2loc_amd.engine import Engine

rng = np.random.default_rng(0)


def unit(x):
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)


def kernels(eng, qd, merge):
    eng.set_option("search_merge_lists", merge)
    eng.set_option("profile_events", 1)
    eng.set_option("profile_rerank", 1)
    for nme in ("search_scan", "search_rerank"):
        eng.kernel_stats(nme)
    for _ in range(300):
        eng.search(qd, 10)
    torch.cuda.synchronize()
    out = {nme: eng.kernel_stats(nme) for nme in ("search_scan", "search_rerank")}
    eng.set_option("profile_events", 0)
    eng.set_option("profile_rerank", 0)
    return {k: (round(v[0] * 1e3, 2), v[1]) for k, v in out.items()}


def run(eng, qd, merge, steps=400):
    eng.set_option("search_merge_lists", merge)
    for _ in range(30):
        eng.search(qd, 10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.search(qd, 10)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


eng = Engine(0)
a = torch.randn(4096, 4096, device="cuda")
for name, db in [("gauss", unit(rng.standard_normal((11259, 256)))),
                 ("clustered3", unit(3.0 * np.repeat(rng.standard_normal((704, 256)), 16, axis=0)[:11259] + rng.standard_normal((11259, 256)))),
                 ("small", unit(rng.standard_normal((1000, 256)))), ("odd", unit(rng.standard_normal((11259 - 37, 256))))]:
    q = unit(rng.standard_normal((4096, 256)) + 0.0)
    q[:512] = unit(db[rng.integers(0, len(db), 512)] + 0.3 * rng.standard_normal((512, 256)))
    eng.db_set(torch.from_numpy(db).cuda())
    qd = torch.from_numpy(q).cuda()
    res = []
    for merge in (0, 1):
        eng.set_option("search_merge_lists", merge)
        idx, sc = eng.search(qd, 10)
        torch.cuda.synchronize()
        res.append((idx.cpu().numpy(), sc.cpu().numpy(), eng.search_counters()))
    ref = np.argsort(-(q.astype(np.float64) @ db.astype(np.float64).T), axis=1, kind="stable")[:, :10]
    for m in (0, 1):
        bad = np.nonzero((res[m][0] != ref).any(axis=1))[0]
        print("   merge", m, "queries off the oracle:", len(bad), bad[:8])
        if len(bad):
            b = bad[0]
            print("      got", res[m][0][b], "\n      ref", ref[b], "\n      sc ", res[m][1][b][:4], (q[b].astype(np.float64) @ db[ref[b][:4]].astype(np.float64).T))
    print(name, "ids equal", np.array_equal(res[0][0], res[1][0]), "scores equal", np.array_equal(res[0][1], res[1][1]),
          "oracle", np.array_equal(res[1][0], ref), res[0][2], res[1][2])
    if name == "gauss":
        for _ in range(2):
            for _ in range(60):
                a @ a
            print("  kernels plain", kernels(eng, qd, 0), "merged", kernels(eng, qd, 1))
            print("  plain %.2f us   merged %.2f us   plain %.2f us   merged %.2f us" % (run(eng, qd, 0), run(eng, qd, 1), run(eng, qd, 0), run(eng, qd, 1)))
