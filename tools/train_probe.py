"""Dev probe: N training steps of the object branch at B=64 (for rocprofv3 --kernel-trace --stats)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from text2loc_amd import synth
from text2loc_amd.engine import Engine

def main(steps=30, B=64, streams=1, bf16=0, block=0, xcd=0):
    eng = Engine(0)
    eng.set_option("train_bf16", bf16)
    if block: eng.set_option("train_gemm_block", block)
    sd = synth.make_object_branch_weights(0)
    cells = synth.make_cells(B, seed=9)
    tens = {}
    for k, v in sd.items():
        if k.endswith("num_batches_tracked") or ".color_encoder." in k or ".mlp_pointnet." in k or ".pointnet." in k:
            continue
        t = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).cuda()
        tens[k] = (t, None if "running_" in k else torch.zeros_like(t))
    eng.train_bind(tens, class_embed=True, color_embed=True)
    p = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in cells.items() if k != "counts"}
    anchor = torch.nn.functional.normalize(torch.randn(B, 256, device="cuda"))
    def step(i):
        eng.zero_grad()
        pos = eng.encode_cells_train(p, dropout_p=0.1, seed=i)
        loss, _, gp = eng.contrastive_loss(anchor, pos, 0.1)
        eng.encode_cells_backward(gp)
        eng.adam_step(1e-3)
    for i in range(3): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps): step(10 + i)
    torch.cuda.synchronize()
    print("steps", steps, "B", B, "bf16", bf16, "gemm block", block or "default", "xcd map", xcd, "ms/step", (time.perf_counter() - t0) / steps * 1e3)

if __name__ == "__main__":
    main(*(int(a) for a in sys.argv[1:]))
