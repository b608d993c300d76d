"""dev: step time of the headline search (Q = 4096, N = 11,259) with the library named by T2L_LIB — used to A/B the shipped build against
`make exp_thirdlists` (the scan stores a third of its candidate lists; results are wrong, only the time means anything)."""
import sys
import time

import numpy as np
import torch

from text2loc_amd.engine import Engine, _LIB_PATH

rng = np.random.default_rng(0)
db = rng.standard_normal((11259, 256)).astype(np.float32)
db /= np.linalg.norm(db, axis=1, keepdims=True)
q = rng.standard_normal((4096, 256)).astype(np.float32)
q /= np.linalg.norm(q, axis=1, keepdims=True)
eng = Engine(0)
eng.db_set(torch.from_numpy(db).cuda())
qd = torch.from_numpy(q).cuda()
a = torch.randn(4096, 4096, device="cuda")
for _ in range(3):
    for _ in range(60):
        a @ a  # clock ramp
    torch.cuda.synchronize()
    for _ in range(50):
        eng.search(qd, 10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(400):
        eng.search(qd, 10)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 400
    eng.set_option("profile_events", 1)
    eng.set_option("profile_rerank", 1)
    for nme in ("search_scan", "search_rerank"):
        eng.kernel_stats(nme)
    for _ in range(200):
        eng.search(qd, 10)
    torch.cuda.synchronize()
    ks = {nme: round(eng.kernel_stats(nme)[0] * 1e3, 2) for nme in ("search_scan", "search_rerank")}
    eng.set_option("profile_events", 0)
    print(_LIB_PATH.split("/")[-1], "us/step %.2f" % (dt * 1e6), ks)
