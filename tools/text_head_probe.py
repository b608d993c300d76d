"""dev probe: t2l_text_head at one search step's worth of queries (4,096 descriptions x 6 hints x 16 tokens)."""
import sys
import time

import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from text2loc_amd import synth
from text2loc_amd.engine import Engine

n_desc = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
L = int(sys.argv[2]) if len(sys.argv) > 2 else 16
eng = Engine(0)
eng.text_head_load_weights(synth.make_language_head_weights(0))
S = n_desc * 6
g = torch.Generator(device="cuda").manual_seed(0)
hidden = 0.2 * torch.randn(S, L, 1024, device="cuda", generator=g)
if len(sys.argv) > 3 and sys.argv[3] == "zeros":   # power probe: the same launches on all-zero activations
    hidden.zero_()
for f16 in (0, 1):
    eng.set_option("encoder_f16", f16)
    for _ in range(2):
        out, bad = eng.text_head(hidden)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        out, flag = eng.text_head(hidden, check=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    flops = 2.0 * S * L * (1024 * 3072 + 1024 * 1024 + 2 * 1024 * 4096) + 2.0 * S * 1024 * 256
    print(f"encoder_f16={f16}: {dt * 1e3:.2f} ms per {n_desc} descriptions ({S * L} tokens), {flops / dt / 1e12:.0f} TFLOP/s algorithmic, "
          f"{flops * (1 if f16 else 3) / dt / 1e12:.0f} executed, overflow={bad}")
