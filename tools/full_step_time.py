import sys, json
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import bench
from text2loc_amd.engine import Engine
from text2loc_amd import synth
eng = Engine(0)
sd = synth.make_object_branch_weights(0)
cells64 = synth.make_cells(64, seed=9)
tens = {}
for k, v in sd.items():
    if k.endswith("num_batches_tracked") or ".color_encoder." in k or ".mlp_pointnet." in k or ".pointnet." in k:
        continue
    t = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).cuda()
    tens[k] = (t, None if "running_" in k else torch.zeros_like(t))
eng.train_bind(tens, class_embed=True, color_embed=True)
p64 = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in cells64.items() if k != "counts"}
print(json.dumps(bench.full_train_step_measure(eng, p64, 30, 30), indent=1))
