"""GPU tier: the fine stage through the C ABI — t2l_fine_load_weights / t2l_fine_encode_objects /
t2l_fine_match — against the goldens of the reference's own CrossMatch.forward and against the numpy oracle."""
import numpy as np
import pytest
import torch

from oracle import t2l_oracle_fine as OF
from text2loc_amd import synth

pytestmark = pytest.mark.gpu


def to_dev(cells, embed):
    keys = ["offsets", "class_idx", "color_idx", "rgb", "center", "n_pts"] + ([] if embed else ["pn_feat"])
    return {k: torch.from_numpy(np.ascontiguousarray(cells[k])).cuda() for k in keys}


@pytest.fixture(scope="module")
def eng():
    from text2loc_amd.engine import Engine

    e = Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize("mode", ["embed", "pn"])
def test_crossmatch_matches_the_reference_run(eng, golden, mode):
    g = golden(f"fine_{mode}")
    embed = mode == "embed"
    sd = synth.make_fine_weights(int(g["weight_seed"]))
    eng.fine_load_weights(sd, class_embed=embed, color_embed=embed)
    cells = {k[3:]: g[k] for k in g.files if k.startswith("in_")}
    desc = eng.fine_encode_objects(to_dev(cells, embed))
    assert desc.shape == (int(g["n_cells"]), 16, 128)
    assert np.abs(desc.cpu().numpy() - g["object_encodings"]).max() < 5e-6
    off = eng.fine_match(desc, torch.from_numpy(g["hint_encodings"]).cuda())
    torch.cuda.synchronize()
    assert np.abs(off.cpu().numpy() - g["offsets_out"]).max() < 5e-5


def test_pairs_by_index_and_ragged_hints(eng):
    """Q queries x K retrieved cells as (cell_index, hint_index) pairs over shared tables; n_hints in {1, 6, 8}."""
    sd = synth.make_fine_weights(3)
    eng.fine_load_weights(sd, class_embed=True, color_embed=True)
    n_cells, Q, K = 7, 5, 3
    cells = synth.make_cells(n_cells, seed=2, min_obj=16, max_obj=16)
    desc = eng.fine_encode_objects(to_dev(cells, True))
    ref_desc = OF.fine_object_encodings(cells, sd, True, True)
    assert np.abs(desc.cpu().numpy() - ref_desc).max() < 5e-6
    rng = np.random.default_rng(0)
    for n_hints in (1, 6, 8):
        hints = rng.standard_normal((Q, n_hints, 128)).astype(np.float32)
        ci = rng.integers(0, n_cells, size=Q * K).astype(np.int32)
        hi = np.repeat(np.arange(Q, dtype=np.int32), K)
        off = eng.fine_match(desc, torch.from_numpy(hints).cuda(), torch.from_numpy(ci).cuda(), torch.from_numpy(hi).cuda())
        ref = OF.cross_match(ref_desc[ci], hints[hi], sd)
        assert off.shape == (Q * K, 2)
        assert np.abs(off.cpu().numpy() - ref).max() < 5e-5


def test_feature_subsets_and_errors(eng):
    from text2loc_amd.engine import T2LError

    sd = synth.make_fine_weights(1)
    cells = synth.make_cells(3, seed=4, min_obj=16, max_obj=16, with_pn_feat=True)
    for feats in (("class", "position"), ("color", "position", "num")):
        sdf = {k: v for k, v in sd.items()}
        rng = np.random.default_rng(len(feats))
        sdf["object_encoder.mlp_merge.0.0.weight"] = (rng.standard_normal((128, 128 * len(feats))) * 0.05).astype(np.float32)
        eng.fine_load_weights(sdf, class_embed=False, color_embed=False, use_features=feats)
        desc = eng.fine_encode_objects(to_dev(cells, False))
        ref = OF.fine_object_encodings(cells, sdf, False, False, use_features=feats)
        assert np.abs(desc.cpu().numpy() - ref).max() < 5e-6
    eng.fine_load_weights(sd, class_embed=True, color_embed=True)
    ragged = synth.make_cells(3, seed=4, min_obj=5, max_obj=9)
    with pytest.raises(T2LError, match="exactly pad_size = 16"):
        eng.fine_encode_objects(to_dev(ragged, True))
    d = torch.zeros(2, 16, 128, device="cuda")
    with pytest.raises(T2LError, match="n_hints <= 8"):
        eng.fine_match(d, torch.zeros(2, 9, 128, device="cuda"))
    broken = {k: v for k, v in sd.items() if k != "cross_hints.1.norm3.bias"}
    with pytest.raises(T2LError, match="cross_hints.1.norm3.bias"):
        eng.fine_load_weights(broken, class_embed=True, color_embed=True)
