"""dev: the text head's training step alone (B=64 x 6 x 16), for rocprofv3 --kernel-trace --stats; argv: bf16 mode (1 | 2)"""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from text2loc_amd.engine import Engine
from text2loc_amd import synth
mode = int(sys.argv[1])
eng = Engine(0)
sd = synth.make_language_head_weights(0)
P = "language_encoder."
tensors = {}
for k, v in sd.items():
    if not k.startswith((P + "intra_module.0.", P + "inter_mlp.0.", P + "inter_module.0.")) or k.endswith("num_batches_tracked"):
        continue
    t = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).cuda()
    tensors[k] = (t, None if "running_" in k else torch.zeros_like(t))
eng.text_train_bind(tensors)
eng.set_option("text_train_bf16", mode)
hidden = 0.2 * torch.randn(384, 16, 1024, device="cuda")
g = torch.randn(64, 256, device="cuda")
import time
for i in range(3):
    eng.text_head_train(hidden, 64, 0.1, i); eng.text_head_backward(g)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(10):
    eng.text_head_train(hidden, 64, 0.1, i); eng.text_head_backward(g)
torch.cuda.synchronize(); print("ms per fwd+bwd", (time.perf_counter() - t0) / 10 * 1e3)
