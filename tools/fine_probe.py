"""dev: fine_match / fine_objects / pointnet eval kernel times with the library named by T2L_LIB."""
import time
import numpy as np, torch
from text2loc_amd import synth
from text2loc_amd.engine import Engine, _LIB_PATH
N, Q, K = 11259, 4096, 10
eng = Engine(0)
eng.fine_load_weights(synth.make_fine_weights(0), class_embed=True, color_embed=True)
cells16 = synth.make_cells(N, seed=17, min_obj=16, max_obj=16)
p16 = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in cells16.items() if k != "counts"}
desc = eng.fine_encode_objects(p16)
rs = np.random.default_rng(3)
hints = torch.from_numpy(rs.standard_normal((Q, 6, 128)).astype(np.float32)).cuda()
ci = torch.from_numpy(rs.integers(0, N, size=Q * K).astype(np.int32)).cuda()
hi = torch.arange(Q, dtype=torch.int32, device="cuda").repeat_interleave(K).contiguous()
a = torch.randn(4096, 4096, device="cuda")
for _ in range(40):
    a @ a
def T(f, n):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print(_LIB_PATH.split("/")[-1], "fine_match ms %.3f" % T(lambda: eng.fine_match(desc, hints, ci, hi), 10), " fine_objects ms %.3f" % T(lambda: eng.fine_encode_objects(p16), 10))
