import sys, numpy as np, torch
sys.path.insert(0, ".")
from text2loc_amd import synth
from text2loc_amd.engine import Engine
eng = Engine(0)
eng.fine_load_weights(synth.make_fine_weights(5), class_embed=True, color_embed=True)
N, Q = 37, 53
cells16 = synth.make_cells(N, seed=77, min_obj=16, max_obj=16)
pk = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in cells16.items() if k != "counts"}
desc = eng.fine_encode_objects(pk)
rng = np.random.default_rng(5)
hints = torch.from_numpy(rng.standard_normal((Q, 6, 128)).astype(np.float32)).cuda()
ci = torch.from_numpy(rng.integers(0, N, size=Q * 10).astype(np.int32)).cuda()
hi = torch.arange(Q, dtype=torch.int32, device="cuda").repeat_interleave(10)
for f32 in (0, 1):
    eng.set_option("encoder_f32", f32)
    a = eng.fine_match(desc, hints, ci, hi).clone()
    b = eng.fine_match(desc, hints, ci, hi).clone()
    print("f32", f32, "run-to-run equal:", bool(torch.equal(a, b)))
    for shift in (1, 2, 3, 5):
        c = eng.fine_match(desc, hints, ci[shift:].contiguous(), hi[shift:].contiguous())
        d = (c - a[shift:]).abs()
        print("  shift", shift, "equal:", bool(torch.equal(c, a[shift:])), "max diff", float(d.max()), "n diff rows", int((d.max(dim=1).values > 0).sum()), "first bad", (d.max(dim=1).values > 0).nonzero().flatten()[:8].tolist())
