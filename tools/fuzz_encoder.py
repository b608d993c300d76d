"""dev: the two-cell encoder kernel against the one-cell kernel and the f32 kernel on random batches (cell counts odd / even / 1, ragged
object counts up to 60, both feature modes), and t2l_text_inter in split-f16 against its plain-f16 option on random (n_desc, S)."""
import sys
import numpy as np, torch
from text2loc_amd import synth
from text2loc_amd.engine import Engine

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for it in range(40):
    embed = bool(rng.integers(0, 2))
    feats = [("class", "color", "position", "num"), ("class", "position"), ("color", "num", "position")][int(rng.integers(0, 3))]
    n = int(rng.choice([1, 2, 3, 7, 64, 255, 1001]))
    sd = synth.make_object_branch_weights(int(rng.integers(0, 50)), use_features=feats)
    cells = synth.make_cells(n, seed=int(rng.integers(0, 1 << 20)), min_obj=1, max_obj=int(rng.choice([5, 28, 60])), with_pn_feat=True)
    pc = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in cells.items() if k != "counts"}
    eng = Engine(0)
    eng.load_weights(sd, class_embed=embed, color_embed=embed, use_features=feats)
    outs = {}
    for name, f32, two in (("two", 0, 1), ("one", 0, 0), ("f32", 1, 0)):
        eng.set_option("encoder_f32", f32)
        eng.set_option("encoder_two_cells", two)
        outs[name] = eng.encode_cells(pc).cpu().numpy()
    e1, e2 = np.abs(outs["two"] - outs["one"]).max(), np.abs(outs["two"] - outs["f32"]).max()
    if not (e1 < 2e-6 and e2 < 2e-6 and np.isfinite(outs["two"]).all()):
        bad += 1
        print("ENCODER MISMATCH", dict(n=n, embed=embed, feats=feats), e1, e2)
    eng.close()
eng = Engine(0)
eng.text_head_load_weights({k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_language_head_weights(1).items()})
for it in range(40):
    S = int(rng.integers(1, 33))
    nd = int(rng.choice([1, 2, 5, 9, 10, 11, 63, 500, 2049]))
    x = torch.from_numpy(rng.standard_normal((nd * S, 256)).astype(np.float32)).cuda()
    outs = []
    for f16 in (0, 1):
        eng.set_option("encoder_f16", f16)
        o, flag = eng.text_inter(x, nd)
        outs.append(o.cpu().numpy())
        assert not flag
    eng.set_option("encoder_f16", 0)
    e = np.abs(outs[0] - outs[1]).max()
    if not e < 2e-3 * max(1.0, np.abs(outs[0]).max()):
        bad += 1
        print("INTER MISMATCH", dict(nd=nd, S=S), e)
print("fuzz done, mismatches:", bad)
sys.exit(1 if bad else 0)
