"""GPU tier: fused per-cell encoder kernel (through the C ABI) vs reference goldens and the oracle.
Tolerance: 1e-3 absolute on unit-norm embeddings is the north-star bar; we assert 2e-5 (fp32 round-off)."""
import numpy as np
import pytest

from oracle import t2l_oracle as O
from text2loc_amd import synth

pytestmark = pytest.mark.gpu
TOL = 2e-5
OBJ_KEYS = ("class_idx", "color_idx", "rgb", "center", "n_pts", "pn_feat")


def _to_gpu(cells):
    import torch

    return {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in cells.items() if k != "counts"}


def _take(cells, n_cells=None, n_objects=None):
    """First n_cells cells, or one cell made of the first n_objects objects."""
    out = {}
    if n_objects is not None:
        out["counts"] = np.array([n_objects], dtype=np.int32)
        out["offsets"] = np.array([0, n_objects], dtype=np.int32)
        hi = n_objects
    else:
        out["counts"] = cells["counts"][:n_cells]
        out["offsets"] = cells["offsets"][: n_cells + 1]
        hi = int(cells["offsets"][n_cells])
    for k in OBJ_KEYS:
        if k in cells:
            out[k] = cells[k][:hi]
    return out


@pytest.fixture(scope="module", params=[(0, 1), (0, 0), (1, 0)], ids=["split-f16-two-cells", "split-f16-one-cell", "f32"])
def eng(request):
    """All encoder kernels against the same bars: split-f16 MFMAs for the big contractions in both forms (two cells per eight-wave
    workgroup on LDS planes — the default — and one cell per four-wave workgroup on f32 tiles), and everything on the f32 MFMA."""
    from text2loc_amd.engine import Engine

    e = Engine(0)
    e.set_option("encoder_f32", request.param[0])
    e.set_option("encoder_two_cells", request.param[1])
    e.encoder_f32 = request.param[0]
    yield e
    e.close()


def _cells(g):
    return {k[3:]: g[k] for k in g.files if k.startswith("in_")}


def test_embed_mode_vs_reference_golden(eng, golden):
    g = golden("encoder_embed")
    sd = synth.make_object_branch_weights(int(g["weight_seed"]))
    eng.load_weights(sd, class_embed=True, color_embed=True)
    out = eng.encode_cells(_to_gpu(_cells(g))).cpu().numpy()
    err = np.abs(out - g["cell_embeddings"]).max()
    print("embed-mode max err", err, "f32" if eng.encoder_f32 else "split-f16")
    assert err < TOL, err


def test_published_mode_downstream_of_pointnet(eng, golden):
    g = golden("encoder_pn")
    sd = synth.make_object_branch_weights(int(g["weight_seed"]))
    cells = _cells(g)
    cells["pn_feat"] = synth.make_cells(int(g["n_cells"]), seed=int(g["cell_seed"]), with_pn_feat=True)["pn_feat"]
    eng.load_weights(sd, class_embed=False, color_embed=False)
    out = eng.encode_cells(_to_gpu(cells)).cpu().numpy()
    err = np.abs(out - g["cell_embeddings"]).max()
    assert err < TOL, err


def test_e2e_golden_cells_then_ids(eng, golden):
    """Encode the 64 golden cells on the GPU, search the golden text embeddings: ids equal the reference's
    wherever the reference's own top-k score gaps exceed the embedding tolerance."""
    import torch

    g = golden("retrieval_e2e")
    sd = synth.make_object_branch_weights(int(g["weight_seed"]))
    eng.load_weights(sd, class_embed=True, color_embed=True)
    enc = eng.encode_cells(_to_gpu(_cells(g)))
    assert np.abs(enc.cpu().numpy() - g["cell_encodings"]).max() < TOL
    eng.db_set(enc)
    k = int(g["top_k"].max())
    idx, _ = eng.search(torch.from_numpy(g["text_encodings"]).cuda(), k)
    got = idx.cpu().numpy()
    full = np.sort(g["cell_encodings"].astype(np.float64) @ g["text_encodings"].astype(np.float64).T, axis=0)[::-1]
    gaps = np.abs(np.diff(full[: k + 1], axis=0)).min(axis=0)
    safe = gaps > 1e-5  # encoder error is ~1e-7 per component; score error well below this
    assert safe.sum() >= 8
    assert np.array_equal(got[safe], g["top_rows"][safe])


@pytest.mark.parametrize("mode", ["embed", "pn", "mixed"])
@pytest.mark.parametrize("feats", [("class", "color", "position", "num"), ("class", "position"), ("num",)])
def test_feature_subsets_vs_oracle(eng, mode, feats):
    ce, co = {"embed": (True, True), "pn": (False, False), "mixed": (True, False)}[mode]
    sd = synth.make_object_branch_weights(3, use_features=feats)
    cells = synth.make_cells(25, seed=12, min_obj=1, max_obj=40, with_pn_feat=True)  # (odd: the two-cell form's tail workgroup)
    ref = O.encode_cells(cells, sd, ce, co, use_features=feats)
    eng.load_weights(sd, class_embed=ce, color_embed=co, use_features=feats)
    out = eng.encode_cells(_to_gpu(cells)).cpu().numpy()
    assert np.abs(out - ref).max() < TOL


def test_counts_edge_cases(eng):
    """1 object, exactly 28, 29 (first truncation), 60 objects; unit-norm outputs; truncation beyond 28."""
    sd = synth.make_object_branch_weights(0)
    eng.load_weights(sd, class_embed=True, color_embed=True)
    big = synth.make_cells(1, seed=2, min_obj=60, max_obj=60)
    for n in (1, 28, 29, 60):
        cells = _take(big, n_objects=n)
        ref = O.encode_cells(cells, sd, True, True)
        out = eng.encode_cells(_to_gpu(cells)).cpu().numpy()
        assert np.abs(out - ref).max() < TOL
        assert abs(np.linalg.norm(out[0]) - 1.0) < 1e-5
    a = eng.encode_cells(_to_gpu(_take(big, n_objects=28))).cpu().numpy()
    b = eng.encode_cells(_to_gpu(_take(big, n_objects=45))).cpu().numpy()
    assert np.array_equal(a, b)


def test_missing_weight_key_is_loud(eng):
    from text2loc_amd.engine import T2LError

    sd = synth.make_object_branch_weights(0)
    del sd["obj_inter_module.1.linear2.weight"]
    with pytest.raises(T2LError, match="missing required key"):
        eng.load_weights(sd, class_embed=True, color_embed=True)


def test_full_size_batch(eng):
    """11,259 cells in one launch: unit norms; a sample equals the oracle (batch independence)."""
    sd = synth.make_object_branch_weights(0)
    eng.load_weights(sd, class_embed=True, color_embed=True)
    cells = synth.make_cells(11259, seed=4)
    out = eng.encode_cells(_to_gpu(cells)).cpu().numpy()
    assert np.abs(np.linalg.norm(out, axis=1) - 1).max() < 1e-5
    ref = O.encode_cells(_take(cells, n_cells=40), sd, True, True)
    assert np.abs(out[:40] - ref).max() < TOL


@pytest.mark.parametrize("name,embed", [("encoder_embed", True), ("encoder_pn", False)])
def test_plain_f16_option_meets_the_north_star_bar(golden, name, embed):
    """Option encoder_f16: one f16 product per operand pair instead of the three of the split form (28 % less time). The bar it
    has to meet is the north star's 1e-3 on unit-norm embeddings; measured ~7e-5 — asserted at 3e-4 — and it really is the
    other kernel (the default's 2e-7 is out of its reach)."""
    from text2loc_amd.engine import Engine

    g = golden(name)
    sd = synth.make_object_branch_weights(int(g["weight_seed"]))
    e = Engine(0)
    try:
        e.load_weights(sd, class_embed=embed, color_embed=embed)
        cells = _cells(g)
        if not embed:
            cells["pn_feat"] = synth.make_cells(int(g["n_cells"]), seed=int(g["cell_seed"]), with_pn_feat=True)["pn_feat"]
        ref = e.encode_cells(_to_gpu(cells)).cpu().numpy()
        e.set_option("encoder_f16", 1)
        out = e.encode_cells(_to_gpu(cells)).cpu().numpy()
        err = np.abs(out - g["cell_embeddings"]).max()
        assert 1e-6 < err < 3e-4, err
        assert np.abs(ref - g["cell_embeddings"]).max() < TOL
        assert np.abs(np.linalg.norm(out, axis=1) - 1).max() < 1e-5
    finally:
        e.close()
