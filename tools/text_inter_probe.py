"""Is t2l_text_inter's result for a description independent of the batch it sits in? (tests/test_gpu_e2e.py)"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from text2loc_amd import synth
from text2loc_amd.text_cache import TextCache
from text2loc_amd.kitti360pose import Kitti360PoseDataset

cells, poses = synth.make_k360_records(150, 96, seed=3, pts_per_obj=(25, 40))
ds = Kitti360PoseDataset.from_records(cells, poses)
args = synth.coarse_args(class_embed=True, color_embed=True)
m = synth.make_coarse_model(args, sentences=TextCache.sentences_of(ds), seed=1)
le = m.language_encoder
eng = le._head_engine(torch.device("cuda", 0))
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(96 * 6, 256, device="cuda", generator=g)
full, _ = eng.text_inter(x, 96, check=False)
for n in (1, 7, 10, 11, 20, 21, 64):
    part, _ = eng.text_inter(x[: n * 6].contiguous(), n, check=False)
    print(n, "equal" if torch.equal(part, full[:n]) else float((part - full[:n]).abs().max()))
# offset start
part, _ = eng.text_inter(x[7 * 6: 14 * 6].contiguous(), 7, check=False)
print("offset 7..14", "equal" if torch.equal(part, full[7:14]) else float((part - full[7:14]).abs().max()))
texts = ds.eval_texts()
with torch.no_grad():
    for bs in (1, 7, 96):
        loop = torch.cat([m.encode_text(texts[i:i + bs]) for i in range(0, len(texts), bs)])
        one = m.encode_text_batches(texts, bs)
        d = (one - loop).abs().max(dim=1)[0]
        print("bs", bs, "equal" if torch.equal(one, loop) else (float(d.max()), d.nonzero().flatten().tolist()[:20]))
