"""PointNet++ training-mode forward/backward timing at the published batch (64 cells): python tools/pn_train_probe.py [B] [bf16] [v1] [iterations]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from text2loc_amd import synth
from text2loc_amd.engine import Engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
bf16 = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # 0 f32, 1 bf16, 2 split-bf16
eng = Engine(0)
eng.set_option("train_bf16", bf16)
cells = synth.make_cells(B, seed=1)
pos, rgb = synth.make_sampled_points(cells, 1)
sd = dict(synth.make_object_branch_weights(2)); sd.update(synth.make_pointnet_weights(1))
tensors = {}
for k, v in sd.items():
    if k.endswith("num_batches_tracked") or k.endswith("_embedding.weight") or "classifier" in k:
        continue
    t = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).cuda()
    tensors[k] = (t, None if "running_" in k else torch.zeros_like(t))
eng.train_bind(tensors, class_embed=False, color_embed=False)
offs = np.asarray(cells["offsets"], dtype=np.int32)
dpos, drgb = torch.from_numpy(pos).cuda(), torch.from_numpy(rgb).cuda()
g = torch.randn(pos.shape[0], 256, device="cuda")
print("objects", pos.shape[0])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 3
for it in range(iters):  # (the GPU reaches its sustained clocks after ~40 ms of load: the last lines are the steady state)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    f2 = eng.pointnet_features_train(dpos, drgb, offs)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    eng.pointnet_backward(g)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"forward {1e3*(t1-t0):.2f} ms  backward {1e3*(t2-t1):.2f} ms")
free, total = torch.cuda.mem_get_info()
print(f"HBM in use {(total-free)/2**30:.1f} GiB")
