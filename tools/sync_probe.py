"""dev: what the K = 20 contract region pays beside its 20 steps — the host's wait at the closing synchronize — with the default and the
spinning device schedule (hipSetDeviceFlags)."""
import ctypes
import sys
import time

import numpy as np
import torch

spin = int(sys.argv[1]) if len(sys.argv) > 1 else 0
hip = ctypes.CDLL("libamdhip64.so")
if spin:
    print("hipSetDeviceFlags(spin) ->", hip.hipSetDeviceFlags(ctypes.c_uint(1)))  # hipDeviceScheduleSpin = 0x1
from text2loc_amd.engine import Engine

rng = np.random.default_rng(0)
db = rng.standard_normal((11259, 256)).astype(np.float32)
db /= np.linalg.norm(db, axis=1, keepdims=True)
q = rng.standard_normal((4096, 256)).astype(np.float32)
q /= np.linalg.norm(q, axis=1, keepdims=True)
eng = Engine(0)
eng.db_set(torch.from_numpy(db).cuda())
qd = torch.from_numpy(q).cuda()
a = torch.randn(4096, 4096, device="cuda")
for _ in range(60):
    a @ a
res = []
for rep in range(12):
    for _ in range(5):
        eng.search(qd, 10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        eng.search(qd, 10)
    t_issue = time.perf_counter()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    res.append(((t1 - t0) / 20 * 1e6, (t_issue - t0) / 20 * 1e6))
print("spin", spin, "us/step (region of 20):", " ".join("%.1f" % r[0] for r in res), "| host issue us/step:", "%.1f" % np.median([r[1] for r in res]))
