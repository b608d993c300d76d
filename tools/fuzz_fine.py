"""Dev tool (GPU box): randomised sweep of t2l_fine_match against the numpy restatement — weight seeds, decoder depth 0 / 1 / 2, pair
counts that do not fill the last workgroup, 1..8 hints per pose, hint magnitudes around the split-f16 guard, plain-f16 option.
python tools/fuzz_fine.py [n_draws] [seed]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import t2l_oracle_fine as OF
from text2loc_amd import synth
from text2loc_amd.engine import Engine


def main(n_draws=30, seed=1):
    rng = np.random.default_rng(seed)
    eng, bad = Engine(0), 0
    for it in range(n_draws):
        layers = int(rng.choice([0, 1, 2]))
        sd = synth.make_fine_weights(int(rng.integers(0, 100)), num_layers=layers)
        embed = bool(rng.integers(0, 2))
        eng.fine_load_weights(sd, class_embed=embed, color_embed=embed, num_layers=layers)
        n_cells = int(rng.choice([1, 3, 20]))
        cells = synth.make_cells(n_cells, seed=int(rng.integers(1 << 20)), min_obj=16, max_obj=16, with_pn_feat=not embed)
        pk = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in cells.items() if k != "counts"}
        desc = eng.fine_encode_objects(pk)
        ref_desc = OF.fine_object_encodings(cells, sd, embed, embed)
        n_pairs, n_h = int(rng.choice([1, 2, 3, 4, 5, 7, 9, 33])), int(rng.integers(1, 9))
        hints = (rng.standard_normal((n_pairs, n_h, 128)) * float(rng.choice([0.1, 1.0, 5.0, 30.0]))).astype(np.float32)
        ci = rng.integers(0, n_cells, size=n_pairs).astype(np.int32)
        hi = np.arange(n_pairs, dtype=np.int32)
        ref = OF.cross_match(ref_desc[ci], hints[hi], sd, n_layers=layers)
        scale = max(1.0, float(np.abs(ref).max()))
        for f16, tol in ((0, 5e-5), (1, 5e-3)):
            eng.set_option("encoder_f16", f16)
            got = eng.fine_match(desc, torch.from_numpy(hints).cuda(), torch.from_numpy(ci).cuda(), torch.from_numpy(hi).cuda()).cpu().numpy()
            err = float(np.abs(got - ref).max())
            if not (np.isfinite(got).all() and err < tol * scale):
                bad += 1
                print("MISMATCH", dict(layers=layers, embed=embed, n_cells=n_cells, n_pairs=n_pairs, n_h=n_h, f16=f16), err, scale)
        eng.set_option("encoder_f16", 0)
    print(f"{n_draws} draws, {bad} mismatches")
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(*(int(a) for a in sys.argv[1:])) else 0)
