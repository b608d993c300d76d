"""dev: t2l_text_inter, one launch vs the GEMM chain, 4,096 descriptions x 6 sentences."""
import time
import numpy as np, torch
from text2loc_amd import synth
from text2loc_amd.engine import Engine
eng = Engine(0)
sd = synth.make_language_head_weights(0)
eng.text_head_load_weights({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
x = torch.randn(4096 * 6, 256, device="cuda")
a = torch.randn(4096, 4096, device="cuda")
for _ in range(40):
    a @ a
for fused in (1, 2, 0, 1, 2, 0):
    eng.set_option("text_inter_fused", fused)
    for _ in range(10):
        eng.text_inter(x, 4096, check=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        eng.text_inter(x, 4096, check=False)
    torch.cuda.synchronize()
    print("fused", fused, "ms %.4f" % ((time.perf_counter() - t0) / 100 * 1e3))
