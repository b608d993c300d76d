"""dev: headline search step under option sweeps (merged records on)."""
import time
import numpy as np, torch
from text2loc_amd.engine import Engine
rng = np.random.default_rng(0)
db = rng.standard_normal((11259, 256)).astype(np.float32); db /= np.linalg.norm(db, axis=1, keepdims=True)
q = rng.standard_normal((4096, 256)).astype(np.float32); q /= np.linalg.norm(q, axis=1, keepdims=True)
eng = Engine(0)
eng.db_set(torch.from_numpy(db).cuda())
qd = torch.from_numpy(q).cuda()
a = torch.randn(4096, 4096, device="cuda")
def run():
    for _ in range(40): eng.search(qd, 10)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(400): eng.search(qd, 10)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 400 * 1e6
for _ in range(60): a @ a
print("base %.2f" % run())
for name, vals, dflt in (("search_xcd_qgroups", (1, 2, 4, 8), 4), ("search_wide_repair", (0, 512), 512)):
    for v in vals:
        eng.set_option(name, v)
        print(name, v, "%.2f" % run())
    eng.set_option(name, dflt)
print("base %.2f" % run())
