"""dev: where the cold end-to-end path (encode cells -> db_set -> search) spends its wall time."""
import time
import numpy as np, torch
from text2loc_amd import synth
from text2loc_amd.engine import Engine
eng = Engine(0)
eng.load_weights({k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_object_branch_weights(0).items()}, class_embed=True, color_embed=True)
cells = synth.make_cells(11259, seed=1)
pc = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in cells.items() if k != "counts"}
q = torch.nn.functional.normalize(torch.randn(4096, 256, device="cuda"))
def T(f, n=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, r
for _ in range(2):
    emb = eng.encode_cells(pc); eng.db_set(emb); eng.search(q, 10)
t_enc, emb = T(lambda: eng.encode_cells(pc))
t_db, _ = T(lambda: eng.db_set(emb))
t_s, _ = T(lambda: eng.search(q, 10))
def all3():
    eng.db_set(eng.encode_cells(pc)); return eng.search(q, 10)
t_all, _ = T(all3)
print("encode %.3f ms  db_set %.3f ms  search %.3f ms  all %.3f ms" % (t_enc, t_db, t_s, t_all))
