"""dev: cProfile of a WARM run_coarse (embedding feature mode, config-2 size) — where the host time of the drop-in surface goes."""
import cProfile, contextlib, io, pstats, sys, time
import torch
sys.path.insert(0, "/root/repo")
from text2loc_amd import synth
from text2loc_amd.coarse import run_coarse
from text2loc_amd.kitti360pose import Kitti360PoseDataset
from text2loc_amd.text_cache import TextCache

cells, poses = synth.make_k360_records(11259, 4096, seed=0)
ds = Kitti360PoseDataset.from_records(cells, poses, object_points=None, seed=0)
args = synth.coarse_args(class_embed=True, color_embed=True)
model = synth.make_coarse_model(args, sentences=TextCache.sentences_of(ds), seed=0)
args.batch_size = 1
dl = torch.utils.data.DataLoader(ds, batch_size=1, collate_fn=ds.collate_fn, shuffle=False)
for _ in range(3):
    with contextlib.redirect_stdout(io.StringIO()):
        run_coarse(model, dl, args)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    with contextlib.redirect_stdout(io.StringIO()):
        run_coarse(model, dl, args)
torch.cuda.synchronize()
print("warm wall ms", (time.perf_counter() - t0) / 5 * 1e3)
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    with contextlib.redirect_stdout(io.StringIO()):
        run_coarse(model, dl, args)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(35)
print(s.getvalue()[:6000])
