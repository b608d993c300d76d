"""CPU tier: host-side logic of the drop-in surface — packer, accuracy bookkeeping of eval_epoch / run_coarse,
the PyTorch text head, checkpoint key names — against the reference goldens. No GPU, no HIP compute calls."""
import argparse
import types

import numpy as np
import pytest
import torch

from oracle import t2l_oracle as O
from text2loc_amd import packing, synth


class Obj:  # plain container duck-typing the reference's Object3d (xyz, rgb, label)
    def __init__(self, label, xyz, rgb):
        self.label, self.xyz, self.rgb = label, xyz, rgb


def make_objects(cells, seed):
    out = [[] for _ in range(len(cells["counts"]))]
    for b, o, label, xyz, rgb in synth.make_object_points(cells, seed):
        out[b].append(Obj(label, xyz, rgb))
    return out


def test_packer_matches_reference_inputs(golden):
    g = golden("encoder_embed")
    n_cells = 6
    cells = synth.make_cells(int(g["n_cells"]), seed=int(g["cell_seed"]))
    objs = make_objects(cells, int(g["cell_seed"]))[:n_cells]
    p = packing.pack_cells(objs, packing.class_table(synth.KNOWN_CLASS))
    hi = int(g["in_offsets"][n_cells])
    assert np.array_equal(p["offsets"], g["in_offsets"][: n_cells + 1])
    for k in ("class_idx", "color_idx", "rgb", "center", "n_pts"):
        assert np.array_equal(p[k], g["in_" + k][:hi]), k  # bit-exact: same numpy reductions, same f32 cast


def test_packer_unknown_label_and_gray_duplicate():
    xyz = np.zeros((30, 3))
    gray1 = np.tile(synth.COLORS[1], (30, 1)).astype(np.float32)  # nearest centre is index 1 ('gray')
    p = packing.pack_cells([[Obj("spaceship", xyz, gray1)]], packing.class_table(synth.KNOWN_CLASS))
    assert p["class_idx"][0] == 0  # known_classes.get(label, 0)
    assert p["color_idx"][0] == 4  # {'gray': 4} wins in the reference's dict (object_encoder.py:35)
    dark = np.tile(synth.COLORS[0], (30, 1)).astype(np.float32)
    p = packing.pack_cells([[Obj("pole", xyz, dark)]], packing.class_table(synth.KNOWN_CLASS))
    assert p["color_idx"][0] == 0 and p["class_idx"][0] == synth.KNOWN_CLASS.index("pole") + 1


# ---------------------------------------------------------------------------------------------------------
class StubCell:
    def __init__(self, cid, bbox, size):
        self.id, self.bbox_w, self.cell_size = str(cid), bbox, float(size)

    def get_center(self):
        return 0.5 * (self.bbox_w[0:3] + self.bbox_w[3:6])


class StubPose:
    def __init__(self, cell_id, pose_w):
        self.cell_id, self.pose_w = str(cell_id), pose_w


def stub_world(g):
    cells = [StubCell(c, b, g["cell_size"]) for c, b in zip(g["db_cell_ids"], g["cell_bbox_w"])]
    poses = [StubPose(c, p) for c, p in zip(g["query_cell_ids"], g["query_pose_w"])]

    class CellDs:
        def __init__(self):
            self.cells = cells

        def __len__(self):
            return len(cells)

        def __getitem__(self, i):
            return {"cells": cells[i], "cell_ids": cells[i].id, "objects": i, "object_points": None}

    class Ds:
        all_cells, all_poses = cells, poses

        def __len__(self):
            return len(poses)

        def __getitem__(self, i):
            return {"texts": i, "cell_ids": poses[i].cell_id}

        def get_cell_dataset(self):
            return CellDs()

    class Model:
        embed_dim = 256

        def eval(self):
            pass

        def encode_text(self, idx):
            return torch.from_numpy(g["text_encodings"][np.array(idx)])

        def encode_objects(self, idx, _):
            return torch.from_numpy(g["cell_encodings"][np.array(idx)])

    ds = Ds()
    from text2loc_amd.coarse import collate_fn
    dl = torch.utils.data.DataLoader(ds, batch_size=16, collate_fn=collate_fn, shuffle=False)
    return Model(), dl


def oracle_retrieve(cell_enc, text_enc, k):
    return O.retrieve_topk(cell_enc.numpy(), text_enc.numpy(), k)


def test_eval_epoch_and_run_coarse_bookkeeping_vs_reference(golden):
    from text2loc_amd.coarse import eval_epoch, run_coarse

    g = golden("retrieval_e2e")
    model, dl = stub_world(g)
    args = argparse.Namespace(ranking_loss="contrastive", top_k=[int(k) for k in g["top_k"]],
                              threshs=[int(t) for t in g["threshs"]], batch_size=16)
    acc, close, retr, ce, te, dists, scores = eval_epoch(model, dl, args, return_distance=True,
                                                         retrieve=oracle_retrieve)
    assert np.array_equal(np.array([acc[k] for k in args.top_k]), g["acc"])
    assert np.array_equal(np.array([close[k] for k in args.top_k]), g["acc_close"])
    ids = g["db_cell_ids"]
    for q in range(len(retr)):
        assert retr[q].dtype.kind == "U" and np.array_equal(retr[q], ids[g["top_rows"][q]])
    assert np.abs(dists - g["top_dists"]).max() < 1e-9
    assert np.abs(scores - g["top_scores"]).max() < 1e-12
    assert ce.dtype == np.float64 and np.array_equal(ce, g["cell_encodings"].astype(np.float64))
    retrievals, at = run_coarse(model, dl, args, retrieve=oracle_retrieve)
    got = np.array([[at[k][t] for t in args.threshs] for k in args.top_k])
    assert np.array_equal(got, g["acc_thresh"])
    assert len(retrievals) == len(dl.dataset.all_poses)


def test_text_head_matches_reference(golden):
    from text2loc_amd.cell_retrieval import LanguageEncoder

    g = golden("text_head")
    B, L = int(g["batch"]), int(g["n_tokens"])
    hidden = torch.from_numpy(synth.make_t5_hidden(6 * B, L, seed=int(g["hidden_seed"])))
    enc = LanguageEncoder(256, fixed_embedding=True, intra_module_num_layers=1, inter_module_num_layers=1,
                          llm_model=object(), tokenizer=None, input_dim=1024)
    sd = {k[len("language_encoder."):]: torch.from_numpy(v) for k, v in
          synth.make_language_head_weights(int(g["weight_seed"])).items()}
    missing, unexpected = enc.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    enc.eval()
    with torch.no_grad():
        out = torch.nn.functional.normalize(enc.head(hidden, B)).numpy()
    assert np.abs(out - g["text_embeddings"]).max() < 2e-5


def test_checkpoint_key_names_match_the_reference_layout():
    from text2loc_amd.cell_retrieval import CellRetrievalNetwork

    args = argparse.Namespace(coarse_embed_dim=256, object_size=28, object_inter_module_num_heads=4,
                              object_inter_module_num_layers=2, hungging_model=None, fixed_embedding=True,
                              intra_module_num_layers=1, intra_module_num_heads=4, inter_module_num_layers=1,
                              inter_module_num_heads=4, class_embed=True, color_embed=True,
                              use_features=["class", "color", "position", "num"])
    from text2loc_amd.cell_retrieval import LanguageEncoder
    le = LanguageEncoder(256, fixed_embedding=True, intra_module_num_layers=1, inter_module_num_layers=1,
                         llm_model=object(), input_dim=1024)
    model = CellRetrievalNetwork(synth.KNOWN_CLASS, synth.COLOR_NAMES, args, language_encoder=le)
    want = {k: v for k, v in synth.make_object_branch_weights(0).items()}
    want.update(synth.make_language_head_weights(0))
    want.update(synth.make_pointnet_weights(0, n_classes=len(synth.KNOWN_CLASS), n_colors=len(synth.COLOR_NAMES)))
    have = model.state_dict()
    for k, v in want.items():
        assert k in have, k
        assert tuple(have[k].shape) == tuple(np.asarray(v).shape), k
    missing, unexpected = model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in want.items()},
                                                strict=True)
    assert not missing and not unexpected
    with pytest.raises(Exception, match="Not implemented"):
        model.forward()
    model.eval()
    with pytest.raises(Exception, match="no CPU fallback|MI355X"):
        model.encode_objects([[Obj("pole", np.zeros((30, 3)), np.zeros((30, 3), np.float32))]], [None])


def test_pointnet_parameter_names_match_the_reference_module(golden):
    """object_encoder.pointnet.* of the mirror == state_dict of the reference's own PointNet2(22, 9, args) (names, shapes)."""
    from text2loc_amd.cell_retrieval import PointNet2Params

    g = golden("pointnet_keys")
    want = {str(n): tuple(int(d) for d in str(s).split(",") if d) for n, s in zip(g["names"], g["shapes"])}
    have = {k: tuple(v.shape) for k, v in PointNet2Params(22, 9).state_dict().items()}
    assert have == want
    sd = synth.make_pointnet_weights(0, n_classes=22, n_colors=9)
    assert {k[len("object_encoder.pointnet."):]: tuple(np.asarray(v).shape) for k, v in sd.items()} == want


def test_crossmatch_mirror_has_the_reference_key_layout():
    """The fine model's parameter names/shapes == synth.make_fine_weights, whose names the reference's own CrossMatch accepted
    with no unexpected key (oracle/gen_golden_fine.py asserts that when the goldens are generated)."""
    from text2loc_amd.cross_matcher import CrossMatch, pad_objects

    args = argparse.Namespace(fine_embed_dim=128, fine_num_decoder_heads=4, fine_num_decoder_layers=2, pad_size=16,
                              fine_intra_module_num_layers=1, fine_intra_module_num_heads=4, hungging_model=None,
                              fixed_embedding=True, class_embed=True, color_embed=True,
                              use_features=["class", "color", "position", "num"])
    model = CrossMatch(synth.KNOWN_CLASS, synth.COLOR_NAMES, args, language_encoder=torch.nn.Identity())
    have = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    for k, v in synth.make_fine_weights(0).items():
        assert have.get(k) == tuple(np.asarray(v).shape), k
    objs = pad_objects([Obj("pole", np.zeros((30, 3)), np.zeros((30, 3), np.float32))] * 3)
    assert len(objs) == 16 and objs[3].label == "pad" and objs[3].xyz.shape == (8, 3) and float(np.abs(objs[3].xyz).max()) < 1e-3
    assert len(pad_objects(objs * 2)) == 16
    with pytest.raises(Exception, match="MI355X|no CPU fallback"):
        model.eval()
        model.encode_cells([objs])


class _WsTokenizer:
    """whitespace tokenizer with padding='longest' (id 0 = pad), the interface LanguageEncoder.forward uses"""

    def __call__(self, sentences, return_tensors="pt", padding="longest"):
        toks = [[(hash(w) % 97) + 1 for w in s.split()] for s in sentences]
        L = max(len(t) for t in toks)
        ids = torch.tensor([t + [0] * (L - len(t)) for t in toks])
        return {"input_ids": ids, "attention_mask": (ids != 0).long()}


class _EmbedLLM(torch.nn.Module):
    """stand-in for the frozen T5 encoder: an embedding table; pad positions get a (non-zero) pad embedding, as T5's do"""

    def __init__(self, dim):
        super().__init__()
        self.table = torch.nn.Embedding(128, dim)

    def forward(self, input_ids, attention_mask, output_attentions=False):
        return argparse.Namespace(last_hidden_state=self.table(input_ids))


def test_run_fine_hint_encoding_pads_per_pose_like_the_reference():
    """evaluation/pipeline.py:113-116 calls the model once per pose, so padding='longest' pads to that pose's longest hint;
    the intra module has no padding mask and max-pools over all positions. encode_pose_hints must reproduce the per-pose
    result when it batches poses — a naive joint call does not."""
    from text2loc_amd.cell_retrieval import LanguageEncoder
    from text2loc_amd.cross_matcher import encode_pose_hints
    from text2loc_amd.engine import T2LError

    torch.manual_seed(0)
    enc = LanguageEncoder(128, fixed_embedding=True, intra_module_num_layers=1, is_fine=True, llm_model=_EmbedLLM(64),
                          tokenizer=_WsTokenizer(), input_dim=64).eval()
    texts = ["The pose is north of a red car. The pose is on-top of a dark-green traffic light pole thing.",
             "The pose is east of a box. The pose is west of a wall.",
             "The pose is south of a blue building. The pose is north of a gray very long vegetation strip here.",
             "The pose is west of a pole. The pose is east of a road."]
    with torch.no_grad():
        per_pose = torch.cat([enc([t]) for t in texts])
        grouped = encode_pose_hints(enc, texts)
        joint = enc(texts)
    assert grouped.shape == (4, 2, 128)
    assert torch.allclose(grouped, per_pose, atol=1e-6)
    assert (joint - per_pose).abs().max() > 1e-4  # the pad length is part of the reference's result
    with pytest.raises(T2LError, match="same number of sentences"):
        enc(["One sentence.", "Two sentences. Here."])


@pytest.mark.parametrize("name", ["pairwise", "hardest"])
def test_other_ranking_losses_match_the_reference(golden, name):
    """training/losses.py:178-253 (value + autograd gradients from the imported reference, oracle/gen_golden_losses.py)"""
    from text2loc_amd.losses import HardestRankingLoss, PairwiseRankingLoss

    g = golden("loss_ranking")
    crit = {"pairwise": PairwiseRankingLoss, "hardest": HardestRankingLoss}[name](margin=float(g["margin"]))
    a = torch.tensor(g["im"], requires_grad=True)
    b = torch.tensor(g["s"], requires_grad=True)
    loss = crit(a, b)
    loss.backward()
    assert abs(float(loss) - float(g[name + "_loss"])) < 1e-5 * max(1.0, abs(float(g[name + "_loss"])))
    assert np.abs(a.grad.numpy() - g[name + "_grad_im"]).max() < 1e-6
    assert np.abs(b.grad.numpy() - g[name + "_grad_s"]).max() < 1e-6


# ---- TextCache host behaviour (advisor, round 4): no pickle in the file, geometric growth, a cap, memos extended not cleared --------
class _FakeTok:
    def __call__(self, sentences, return_tensors=None, padding=None, max_length=None):
        ids = [[1 + (sum(map(ord, w)) % 50) for w in s.split()] + [1] for s in sentences]
        if return_tensors is None:
            return {"input_ids": ids}
        import torch

        L = max_length
        inp = torch.zeros((len(ids), L), dtype=torch.long)
        att = torch.zeros((len(ids), L), dtype=torch.long)
        for i, r in enumerate(ids):
            inp[i, :len(r)] = torch.tensor(r)
            att[i, :len(r)] = 1
        return {"input_ids": inp, "attention_mask": att}


class _FakeT5:
    calls = 0

    def __call__(self, input_ids, attention_mask, output_attentions=False):
        import types

        import torch

        _FakeT5.calls += 1
        h = torch.sin(input_ids.float()[..., None] * torch.arange(1, 9).float() + torch.arange(input_ids.shape[1]).float()[None, :, None])
        return types.SimpleNamespace(last_hidden_state=h)


class _FakeHead:
    runs = []

    def _head_first_half(self, hidden):
        _FakeHead.runs.append(int(hidden.shape[0]))
        return hidden.amax(dim=1)


def test_text_cache_growth_cap_memo_and_pickle_free_file(tmp_path):
    import torch

    from text2loc_amd.text_cache import TextCache

    c = TextCache(_FakeTok(), _FakeT5(), "cpu", max_tokens=6, dim=8, max_sentences=40)
    first = [f"a b c{i}" for i in range(10)]
    assert c.add(first) and c.hidden.shape == (10, 6, 8)
    buf_ptr = c._buf.data_ptr()
    keep = c.hidden.clone()
    _FakeHead.runs.clear()
    v1 = c.sentence_vectors(_FakeHead(), 4, version=1)
    assert v1.shape == (10, 8) and _FakeHead.runs == [10]
    # a miss appends into the same buffer (capacity 256 > 15): no copy of the cache, earlier rows untouched, memo EXTENDED by 5 rows
    assert c.add([f"d e f{i}" for i in range(5)]) and c._buf.data_ptr() == buf_ptr and torch.equal(c.hidden[:10], keep)
    v2 = c.sentence_vectors(_FakeHead(), 4, version=1)
    assert v2.shape == (15, 8) and _FakeHead.runs == [10, 5] and torch.equal(v2[:10], v1)
    assert c.sentence_vectors(_FakeHead(), 4, version=1) is v2 and _FakeHead.runs == [10, 5]
    # the cap: refused (caller falls back to T5), nothing changed
    assert not c.add([f"g h i{i}" for i in range(30)]) and len(c.index) == 15 and c.hidden.shape[0] == 15
    assert not c.add(["one two three four five six seven"])  # longer than max_tokens
    # lookup of a novel sentence adds exactly it
    rows, L = c.lookup(["a b c3", "x y"])
    assert rows.tolist() == [3, 15] and L == 4 and len(c.index) == 16
    # persistence: a file np.load opens with allow_pickle=False, same content back
    path = str(tmp_path / "cache.npz")
    c.save(path)
    z = np.load(path, allow_pickle=False)
    assert z["sentences"].dtype.kind == "U" and list(z["sentences"])[:2] == ["a b c0", "a b c1"]
    d = TextCache.load(path, device="cpu")
    assert d.index == c.index and torch.equal(d.hidden, c.hidden) and np.array_equal(d.n_tok, c.n_tok)
    # a round-4 style file (object array = pickle) is refused with an explanation, never unpickled
    bad = str(tmp_path / "old.npz")
    np.savez(bad, sentences=np.array(["a b"], dtype=object), n_tok=np.array([3], dtype=np.int32), hidden=np.zeros((1, 6, 8), np.float32),
             max_tokens=np.int32(6), dim=np.int32(8))
    with pytest.raises(ValueError, match="never unpickles"):
        TextCache.load(bad, device="cpu")
