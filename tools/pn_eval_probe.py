"""Dev tool (GPU box): time of t2l_pointnet_features on bench.py's 512-cell batch (10,698 objects), default split-f16 and option
encoder_f16, and the largest deviation from the restatement on the first two cells. Run under rocprofv3 --kernel-trace --stats
(tools/pn_eval_prof.sh) for the per-kernel split."""
import os.path as osp
import sys

import numpy as np
import torch

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
from text2loc_amd import synth  # noqa: E402
from text2loc_amd.engine import Engine  # noqa: E402

n_pc = int(sys.argv[1]) if len(sys.argv) > 1 else 512
cells = synth.make_cells(n_pc, seed=13)
n_obj = int(cells["offsets"][-1])
pos_np, rgb_np = synth.make_sampled_points(cells, 13)
sd = dict(synth.make_object_branch_weights(0))
sd.update(synth.make_pointnet_weights(0))
eng = Engine(0)
eng.set_option("profile_events", 1)
eng.load_weights(sd, class_embed=False, color_embed=False)
d_pos, d_rgb = torch.from_numpy(pos_np).cuda(), torch.from_numpy(rgb_np).cuda()
res = {}
for name, opt in (("split_f16", 0), ("plain_f16", 1)):
    eng.set_option("encoder_f16", opt)
    for _ in range(3):
        f = eng.pointnet_features(d_pos, d_rgb, cells["offsets"])
    eng.kernel_stats("pointnet")
    for _ in range(5):
        f = eng.pointnet_features(d_pos, d_rgb, cells["offsets"])
    torch.cuda.synchronize()
    ms, n = eng.kernel_stats("pointnet")
    res[name] = f
    print(f"{name}: {ms:.3f} ms per call, {n_obj} objects, {n_obj / ms * 1e3:.0f} objects/s")
from oracle import t2l_oracle_pointnet as OP  # noqa: E402

n2 = int(cells["offsets"][2])
ref = OP.pointnet_features(pos_np[:n2], rgb_np[:n2], cells["offsets"][:3], sd)
print("max |split - restatement| on", n2, "objects:", float(np.abs(res["split_f16"][:n2].cpu().numpy() - ref).max()),
      " max |plain - split|:", float((res["plain_f16"] - res["split_f16"]).abs().max()), " max |feature|:", float(res["split_f16"].abs().max()))
