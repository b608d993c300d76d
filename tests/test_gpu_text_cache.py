"""GPU tier, a5 / VERDICT r3 item 4b: the per-sentence cache of the frozen T5 encoder (text2loc_amd.text_cache.TextCache) in front
of the engine's head. ``eval_epoch`` on the k360_tiny dataset (written by the reference's own classes, tests/golden) with a tiny
random T5 (d_model 1024 so the head's shapes are t5-large's; transformers is in the image, no weights are downloaded): the cached
run — gather + t2l_text_head + t2l_text_inter, no tokenizer-to-T5 pass — equals the uncached run (tokenizer(padding="longest") ->
T5 -> head, models/language_encoder.py:106-148) to 2e-5 in every text embedding and retrieves the same cells."""
import os.path as osp

import numpy as np
import pytest
import torch

from text2loc_amd import synth

pytestmark = pytest.mark.gpu
GOLDEN = osp.join(osp.dirname(osp.abspath(__file__)), "golden")


def _tiny_t5(sentences, seed=0):
    from tokenizers import Tokenizer, models, pre_tokenizers, processors
    from transformers import PreTrainedTokenizerFast, T5Config, T5EncoderModel

    vocab = {"<pad>": 0, "</s>": 1, "<unk>": 2}
    pre = pre_tokenizers.Whitespace()
    for s in sentences:
        for w, _ in pre.pre_tokenize_str(s):
            vocab.setdefault(w, len(vocab))
    tok = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre
    tok.post_processor = processors.TemplateProcessing(single="$A </s>", special_tokens=[("</s>", 1)])
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, pad_token="<pad>", eos_token="</s>", unk_token="<unk>")
    torch.manual_seed(seed)
    cfg = T5Config(vocab_size=len(vocab), d_model=1024, d_kv=64, d_ff=256, num_layers=2, num_heads=4, is_encoder_decoder=False,
                   use_cache=False)
    return fast, T5EncoderModel(cfg).eval()


def _model_and_data():
    from tests.test_gpu_train_loop import _args
    from text2loc_amd import kitti360pose as K
    from text2loc_amd.cell_retrieval import CellRetrievalNetwork, LanguageEncoder
    from text2loc_amd.text_cache import TextCache

    g = np.load(osp.join(GOLDEN, "k360_tiny.npz"), allow_pickle=False)
    ds = K.Kitti360PoseDataset(osp.join(GOLDEN, "k360_tiny"), [str(s) for s in g["scenes"]])
    sentences = TextCache.sentences_of(ds)
    tok, t5 = _tiny_t5(sentences)
    le = LanguageEncoder(256, fixed_embedding=True, intra_module_num_layers=1, intra_module_num_heads=4, inter_module_num_layers=1,
                         inter_module_num_heads=4, llm_model=t5, tokenizer=tok)
    args = _args(top_k=[1, 3, 5], batch_size=5)
    model = CellRetrievalNetwork(ds.get_known_classes(), synth.COLOR_NAMES, args, language_encoder=le)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_object_branch_weights(4).items()}
    sd.update({k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_language_head_weights(2).items()})
    model.load_state_dict(sd, strict=False)
    model = model.to("cuda").eval()
    dl = torch.utils.data.DataLoader(ds, batch_size=5, collate_fn=K.Kitti360PoseDataset.collate_fn, shuffle=False)
    return model, ds, dl, args, sentences


def test_eval_epoch_with_the_sentence_cache_equals_the_uncached_run():
    from text2loc_amd.cell_retrieval import LanguageEncoder
    from text2loc_amd.coarse import eval_epoch
    from text2loc_amd.text_cache import TextCache

    model, ds, dl, args, sentences = _model_and_data()
    le = model.language_encoder
    lens = [len(i) for i in le.tokenizer(sentences)["input_ids"]]
    assert len(set(lens)) > 1  # the batches' longest sentence varies: pad positions are exercised
    t5_0 = LanguageEncoder.t5_calls
    acc_a, close_a, ret_a, ce_a, te_a = eval_epoch(model, dl, args, return_encodings=True)
    n_batches = LanguageEncoder.t5_calls - t5_0
    assert n_batches == (len(ds) + 4) // 5

    cache = TextCache.build(le, ds)
    assert len(cache.index) == len(sentences) and cache.max_tokens >= max(lens)
    le.text_cache = cache
    runs = {}
    for memo in (True, False):
        le.memoise_sentence_vectors = memo
        t5_1, c_1, e_1 = LanguageEncoder.t5_calls, LanguageEncoder.cache_calls, LanguageEncoder.head_engine_calls
        acc_b, close_b, ret_b, ce_b, te_b = eval_epoch(model, dl, args, return_encodings=True)
        assert LanguageEncoder.t5_calls == t5_1 and LanguageEncoder.cache_calls == c_1 + n_batches  # no tokenizer -> T5 pass at all
        assert LanguageEncoder.head_engine_calls > e_1  # the head ran in the engine (memo: once per distinct L, else per batch)
        assert np.abs(te_b - te_a).max() < 2e-5, np.abs(te_b - te_a).max()
        assert np.array_equal(ce_b, ce_a) and acc_b == acc_a and close_b == close_a
        assert all(np.array_equal(ret_b[k], ret_a[k]) for k in ret_a)
        runs[memo] = te_b
    assert np.abs(runs[True] - runs[False]).max() < 1e-6  # the memo changes where a sentence is computed, not what
    st = cache.stats()
    assert st["t5_sentences"] == len(sentences) and st["batches_with_misses"] == 0

    # the cache as a file ("T5 embeddings precomputed"): a fresh load serves the same embeddings without tokenizer or T5
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        fn = osp.join(td, "t5_states.npz")
        cache.save(fn)
        loaded = TextCache.load(fn, device="cuda")
        assert loaded.tokenizer is None and len(loaded.index) == len(cache.index)
        le.text_cache = loaded
        le.memoise_sentence_vectors = True
        _, _, _, _, te_c = eval_epoch(model, dl, args, return_encodings=True)
        assert np.abs(te_c - te_a).max() < 2e-5
        le.text_cache = cache
    # an unseen sentence is encoded once and added; one the cache cannot hold sends the batch through T5
    new = "The pose is north of a gray pole."
    texts = [" ".join([new] + list(ds.hint_descriptions[0][1:]))]
    with torch.no_grad():
        a = model.encode_text(texts)
        le.text_cache = None
        b = model.encode_text(texts)
        le.text_cache = cache
    assert float((a - b).abs().max()) < 2e-5 and cache.stats()["batches_with_misses"] >= (0 if new in sentences else 1)
    long = " ".join(["The pose is north of a gray pole pole pole pole pole pole pole pole pole pole pole pole pole pole pole pole pole pole pole pole pole pole pole pole pole pole pole."] * 1
                    + list(ds.hint_descriptions[0][1:]))
    t5_2 = LanguageEncoder.t5_calls
    with torch.no_grad():
        model.encode_text([long])
    assert LanguageEncoder.t5_calls == t5_2 + 1


def test_cache_serves_training_mode_with_the_frozen_t5():
    """--fixed_embedding (README.md:87-99): opt-in (cache_in_training) — the reference leaves the frozen T5's dropout active under
    model.train(); with that dropout off T5 is a constant of the sentence in training too, and the head behind the cache keeps its
    PyTorch modules and its gradients."""
    from text2loc_amd.cell_retrieval import LanguageEncoder
    from text2loc_amd.text_cache import TextCache

    model, ds, dl, args, _ = _model_and_data()
    le = model.language_encoder
    batch = next(iter(dl))
    model.train()
    for m in le.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    for layer in list(le.intra_module) + list(le.inter_module):
        layer.self_attn.dropout = 0.0
    a = model.encode_text(batch["texts"])
    le.llm_model.eval()  # (T5's own dropout off: what the cache holds)
    a0 = model.encode_text(batch["texts"])
    assert float((a - a0).abs().max()) > 1e-3  # the reference's training step really carries T5 dropout noise
    le.text_cache = TextCache.build(le, ds)
    t5 = LanguageEncoder.t5_calls
    b = model.encode_text(batch["texts"])
    assert LanguageEncoder.t5_calls == t5 + 1  # not served from the cache unless asked to
    le.cache_in_training = True
    t5 = LanguageEncoder.t5_calls
    b = model.encode_text(batch["texts"])
    a = a0
    assert LanguageEncoder.t5_calls == t5 and b.requires_grad
    assert float((a - b).abs().max()) < 2e-5
    b.sum().backward()
    assert le.inter_mlp[0][0].weight.grad is not None
