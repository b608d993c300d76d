"""dev: encode_cells and t2l_text_inter timed with the library named by T2L_LIB (A/B against `make exp_enc EXPFLAG=-DT2L_EXP_HOTW`)."""
import time
import numpy as np, torch
from text2loc_amd import synth
from text2loc_amd.engine import Engine, _LIB_PATH
eng = Engine(0)
sd = synth.make_language_head_weights(0)
eng.text_head_load_weights({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
x = torch.randn(4096 * 6, 256, device="cuda")
eng.load_weights({k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_object_branch_weights(0).items()}, class_embed=True, color_embed=True)
cells = synth.make_cells(11259, seed=1)
pc = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in cells.items() if k != "counts"}
a = torch.randn(4096, 4096, device="cuda")
for _ in range(40):
    a @ a
for name, fn, reps in (("text_inter", lambda: eng.text_inter(x, 4096, check=False), 100), ("encode_cells", lambda: eng.encode_cells(pc), 20)):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    print(_LIB_PATH.split("/")[-1], name, "ms %.4f" % ((time.perf_counter() - t0) / reps * 1e3))
