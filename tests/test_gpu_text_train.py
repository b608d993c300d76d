"""GPU tier, SURVEY.md 8 f-4 / the text half of a9: the head after the frozen T5 in TRAINING mode on the engine
(t2l_text_train_bind / t2l_text_head_train / t2l_text_head_backward, behind ``LanguageEncoder.head`` under ``model.train()``):
against the float64 oracle (oracle/t2l_oracle_text_train.py, pinned to the reference by tests/test_oracle_train.py) with the
counter-based dropout masks ON, against the reference's own step (tests/golden/train_step_text.npz, dropout sites at p = 0), and
against torch autograd over the same nn.Modules for a few Adam steps."""
import copy

import numpy as np
import pytest
import torch

from text2loc_amd import synth

pytestmark = pytest.mark.gpu
P = "language_encoder."


def _bind(eng, sd):
    tensors = {}
    for k, v in sd.items():
        if not k.startswith((P + "intra_module.0.", P + "inter_mlp.0.", P + "inter_module.0.")) or k.endswith("num_batches_tracked"):
            continue
        t = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).cuda()
        tensors[k] = (t, None if "running_" in k else torch.zeros_like(t))
    eng.text_train_bind(tensors)
    return tensors


def _check_grads(tensors, ref_grads, tol_rms, frac=0.97):
    worst = {}
    for n, rg in ref_grads.items():
        g = tensors[n][1].cpu().numpy().astype(np.float64)
        rg = np.asarray(rg, dtype=np.float64).reshape(g.shape)
        rms = float(np.sqrt((rg ** 2).mean()))
        err = np.abs(g - rg)
        if n.endswith(("inter_mlp.0.0.bias", "intra_module.0.norm2.bias")):
            # true gradient 0: a constant per column in front of a BatchNorm (the Linear's bias; LayerNorm2's bias shifts every token
            # of a column alike, so it passes the max and the Linear as a constant) — float32 leaves noise
            assert np.abs(rg).max() < 1e-9 and err.max() < 1e-4, n
            continue
        if n.endswith("in_proj_bias"):  # the key third has true gradient 0 (softmax is shift-invariant)
            D = g.size // 3
            sel = np.r_[0:D, 2 * D:3 * D]
            err, rms = err[sel], float(np.sqrt((rg[sel] ** 2).mean()))
        worst[n] = float(err.max() / (rms + 1e-12))
        # (a ReLU input within rounding of 0 lands on the other side in another arithmetic and moves that unit's whole row / column of
        # the upstream gradients: the tail bound is on the 99.5th percentile, not on the maximum — cf. DESIGN.md §1, train_step goldens)
        assert float((err < tol_rms * rms + 1e-7).mean()) >= frac and float(np.quantile(err, 0.995)) < 20 * tol_rms * rms + 1e-6, (n, worst[n])
    return worst


@pytest.mark.parametrize("arith", [2, 1], ids=["split_bf16", "bf16"])
@pytest.mark.parametrize("n_desc,S,L,p", [(4, 6, 7, 0.1), (3, 6, 16, 0.0), (2, 5, 32, 0.1), (9, 1, 1, 0.1), (16, 6, 12, 0.1)])
def test_engine_text_train_matches_the_float64_oracle(n_desc, S, L, p, arith):
    from oracle import t2l_oracle_text_train as OTT
    from text2loc_amd.engine import Engine

    sd = synth.make_language_head_weights(6)
    hidden = synth.make_t5_hidden(n_desc * S, L, seed=n_desc * 10 + L)
    rng = np.random.default_rng(L)
    G = rng.standard_normal((n_desc, 256)).astype(np.float32)
    seed = 1234 + L
    eng = Engine(0)
    try:
        tensors = _bind(eng, sd)
        eng.set_option("text_train_bf16", arith)  # default 2 (split-bf16: f32-class); 1 = plain bf16 operands (config 4)
        out = eng.text_head_train(torch.from_numpy(hidden).cuda(), n_desc, dropout_p=p, seed=seed)
        eng.text_head_backward(torch.from_numpy(G).cuda())
        torch.cuda.synchronize()
        ref, info = OTT.text_head_train(hidden, sd, n_desc, grad_out=G, p_drop=float(np.float32(p)), seed=seed)
        if arith == 1:  # bf16 operands: 2^-9 per product — the forward within 2 % of the output scale, gradients not compared element-wise
            assert np.abs(out.cpu().numpy() - ref).max() < 2e-2 * max(1.0, np.abs(ref).max())
            g = tensors[P + "inter_module.0.linear2.weight"][1].cpu().numpy().astype(np.float64)
            rg = np.asarray(info["grads"][P + "inter_module.0.linear2.weight"]).reshape(g.shape)
            cos = float((g * rg).sum() / np.sqrt((g * g).sum() * (rg * rg).sum()))
            assert cos > 0.9, cos  # (a BatchNorm over a few dozen rows amplifies 2^-9 operand rounding: direction, not elements)
            return
        assert np.abs(out.cpu().numpy() - ref).max() < 1e-4 * max(1.0, np.abs(ref).max())
        # float32 summation order under a BatchNorm over a few dozen rows and through two LayerNorm'd layers: median-tight, tails bounded
        _check_grads(tensors, info["grads"], tol_rms=1e-2, frac=0.9)
        new = __import__("oracle.t2l_oracle_train", fromlist=["x"]).bn_running_update(sd, info["bn_stats"])
        for k in (P + "inter_mlp.0.1.running_mean", P + "inter_mlp.0.1.running_var"):
            assert np.allclose(tensors[k][0].cpu().numpy(), new[k], rtol=2e-4, atol=2e-5), k
        # a second backward of the same forward accumulates (+=)
        g1 = tensors[P + "inter_module.0.linear2.weight"][1].clone()
        eng.text_head_backward(torch.from_numpy(G).cuda())
        torch.cuda.synchronize()
        assert torch.allclose(tensors[P + "inter_module.0.linear2.weight"][1], 2 * g1, rtol=1e-4, atol=1e-6)
    finally:
        eng.close()


def _encoder(seed):
    from text2loc_amd.cell_retrieval import LanguageEncoder

    enc = LanguageEncoder(256, fixed_embedding=True, intra_module_num_layers=1, inter_module_num_layers=1, llm_model=object(),
                          tokenizer=None, input_dim=1024)
    sd = {k[len(P):]: torch.from_numpy(v) for k, v in synth.make_language_head_weights(seed).items()}
    missing, unexpected = enc.load_state_dict(sd, strict=False)
    assert not missing and not unexpected
    return enc.cuda()


def _no_dropout(enc):
    for layer in list(enc.intra_module) + list(enc.inter_module):
        layer.dropout.p = layer.dropout1.p = layer.dropout2.p = 0.0
        layer.self_attn.dropout = 0.0


def test_language_encoder_train_step_matches_the_reference_step(golden):
    """LanguageEncoder.head under train() -> F.normalize -> ContrastiveLoss -> backward, served by the engine, against the imported
    reference's own run of exactly that (train_step_text.npz: head output, loss, every head gradient, the BatchNorm buffers)."""
    from tests.test_oracle_train import golden_view
    from text2loc_amd.cell_retrieval import LanguageEncoder
    from text2loc_amd.losses import ContrastiveLoss

    g = golden("train_step_text")
    B, S, L = int(g["batch"]), int(g["n_hints"]), int(g["n_tokens"])
    enc = _encoder(int(g["weight_seed"]))
    _no_dropout(enc)
    enc.train()
    hidden = torch.from_numpy(synth.make_t5_hidden(B * S, L, seed=int(g["hidden_seed"]))).cuda()
    n0 = LanguageEncoder.train_engine_calls
    out = enc.head(hidden, B)
    assert LanguageEncoder.train_engine_calls == n0 + 1 and out.requires_grad
    assert np.abs(out.detach().cpu().numpy() - g["head_out"]).max() < 5e-5 * max(1.0, np.abs(g["head_out"]).max())
    anchor = torch.nn.functional.normalize(out)
    loss = ContrastiveLoss(float(g["temperature"]))(anchor, torch.from_numpy(g["cells"]).cuda())
    assert abs(float(loss.detach()) - float(g["loss"])) < 2e-5 * abs(float(g["loss"]))
    loss.backward()
    torch.cuda.synchronize()
    params = dict(enc.named_parameters())
    for n in [str(x) for x in g["used_params"]]:
        grad = params[n[len(P):]].grad
        assert grad is not None, n
        exp, got = golden_view(g, "grad", n, grad.cpu().numpy())
        rms = float(g[f"grad_norm/{n}"]) / np.sqrt(max(grad.numel(), 1))
        if n.endswith(("inter_mlp.0.0.bias", "intra_module.0.norm2.bias")):
            assert np.abs(got).max() < 1e-4
            continue
        if n.endswith("in_proj_bias") and len(got) <= 1024:
            D = len(got) // 3
            sel = np.r_[0:D, 2 * D:3 * D]
            exp, got = exp[sel], got[sel]
        err = np.abs(got - exp)
        assert (err < 1e-2 * rms + 1e-6).mean() >= 0.95 and err.max() < 0.2 * rms + 1e-5, (n, float(err.max()), rms)
    for k in g.files:
        if k.startswith("buf/"):
            got = dict(enc.named_buffers())[k[4 + len(P):]].cpu().numpy()
            assert np.allclose(got, g[k], rtol=2e-4, atol=2e-5), k


def test_engine_train_head_tracks_torch_autograd_over_adam_steps():
    """Four optimisation steps (dropout off): the engine-served head + torch.optim.Adam on its parameters follows a deep copy
    trained by torch autograd on the PyTorch modules; then eval-mode agreement of the two."""
    from text2loc_amd.cell_retrieval import LanguageEncoder
    from text2loc_amd.losses import ContrastiveLoss

    B, S, L = 16, 6, 10
    enc = _encoder(3)
    _no_dropout(enc)
    ref = copy.deepcopy(enc)
    ref.use_engine_head = False
    opt, opt_ref = torch.optim.Adam(enc.parameters(), lr=1e-4), torch.optim.Adam(ref.parameters(), lr=1e-4)
    crit = ContrastiveLoss(0.1)
    enc.train()
    ref.train()
    cells = torch.nn.functional.normalize(torch.randn(B, 256, generator=torch.Generator().manual_seed(0))).cuda()
    losses, losses_ref = [], []
    n0, t0 = LanguageEncoder.train_engine_calls, LanguageEncoder.head_torch_calls
    for step in range(4):
        hidden = torch.from_numpy(synth.make_t5_hidden(B * S, L, seed=50 + step)).cuda()
        for o, m, acc in ((opt, enc, losses), (opt_ref, ref, losses_ref)):
            o.zero_grad()
            loss = crit(torch.nn.functional.normalize(m.head(hidden, B)), cells)
            loss.backward()
            o.step()
            acc.append(float(loss.detach()))
    assert LanguageEncoder.train_engine_calls == n0 + 4 and LanguageEncoder.head_torch_calls == t0 + 4
    assert np.allclose(losses, losses_ref, rtol=2e-3), (losses, losses_ref)
    enc.eval()
    ref.eval()
    hidden = torch.from_numpy(synth.make_t5_hidden(B * S, L, seed=99)).cuda()
    with torch.no_grad():
        a, b = enc.head(hidden, B), ref.head(hidden, B)
    assert float((a - b).abs().max()) < 5e-3 * float(b.abs().max())
    assert int(enc.inter_mlp[0][1].num_batches_tracked) == int(ref.inter_mlp[0][1].num_batches_tracked)


def test_training_mode_keeps_the_pytorch_modules_where_the_engine_does_not_apply():
    from text2loc_amd.cell_retrieval import LanguageEncoder

    enc = _encoder(1)
    enc.train()
    hidden = torch.from_numpy(synth.make_t5_hidden(12, 8, seed=1)).cuda()
    n0 = LanguageEncoder.train_engine_calls
    enc.intra_module[0].dropout1.p = 0.3  # site-specific probabilities: the PyTorch path
    y = enc.head(hidden, 2)
    assert LanguageEncoder.train_engine_calls == n0 and y.requires_grad
    enc.intra_module[0].dropout1.p = 0.1
    h2 = hidden.clone().requires_grad_(True)  # a trainable T5 upstream needs d/d hidden
    enc.head(h2, 2).sum().backward()
    assert LanguageEncoder.train_engine_calls == n0 and h2.grad is not None
    y = enc.head(hidden, 2)  # the published configuration: served by the engine, with live dropout
    assert LanguageEncoder.train_engine_calls == n0 + 1
    y2 = enc.head(hidden, 2)
    assert float((y - y2).abs().max()) > 1e-4  # a fresh mask every call


def test_engine_adam_for_the_head_is_torch_adam_arithmetic_and_survives_a_checkpoint():
    """``t2l_text_adam_step`` (what ``text2loc_amd.optim.Adam`` hands the head's trained parameters to): three steps on given gradients
    equal ``torch.optim.Adam`` on a deep copy element for element (same float32 arithmetic: 1e-7 relative), ``zero_grad`` clears the bound
    buffers in place, the moments travel through ``text_adam_state`` / ``set_text_adam_state`` into a second head that then takes the same
    fourth step, and the eval-mode weights of the engine head follow the stepped parameters (no stale fold)."""
    enc = _encoder(11)
    ref = copy.deepcopy(enc)
    named = enc.engine_optimizer_params()
    assert len(named) == 28 and sum(p.numel() for _, p in named) > 13_000_000
    ref_params = dict(ref.named_parameters())
    opt_ref = torch.optim.Adam([ref_params[n] for n, _ in named], lr=3e-4, betas=(0.9, 0.999), eps=1e-8)
    gen = torch.Generator(device="cuda").manual_seed(1)

    def give_grads(step):
        for n, p in named:
            g = torch.randn(p.shape, device="cuda", generator=gen) * (10.0 ** (step - 2))
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            ref_params[n].grad = g.clone()

    for step in range(3):
        give_grads(step)
        enc.engine_adam_step(3e-4, (0.9, 0.999), 1e-8)
        opt_ref.step()
    torch.cuda.synchronize()
    for n, p in named:
        a, b = p.detach(), ref_params[n].detach()
        assert float((a - b).abs().max()) <= 2e-7 * (1.0 + float(b.abs().max())), n
    # zero_grad in place: same storage, all zeros
    ptrs = [p.grad.data_ptr() for _, p in named]
    enc.engine_zero_grad()
    torch.cuda.synchronize()
    assert all(float(p.grad.abs().max()) == 0.0 for _, p in named) and ptrs == [p.grad.data_ptr() for _, p in named]
    # checkpoint: moments + step into a fresh head holding the same parameters
    m, v, st = enc._th_train_engine.text_adam_state()
    assert st == 3 and m.numel() == sum(p.numel() for _, p in named)
    twin = _encoder(11)
    twin.load_state_dict(enc.state_dict())
    twin._bind_text_train(torch.device("cuda", 0))
    twin._th_train_engine.set_text_adam_state(m, v, st)
    tw = dict(twin.named_parameters())
    give_grads(3)
    for n, p in named:
        tw[n].grad.copy_(p.grad)
    enc.engine_adam_step(3e-4)
    twin.engine_adam_step(3e-4)
    opt_ref.step()
    torch.cuda.synchronize()
    for n, p in named:
        assert torch.equal(p.detach(), tw[n].detach()), n
        assert float((p.detach() - ref_params[n].detach()).abs().max()) <= 3e-7 * (1.0 + float(ref_params[n].detach().abs().max())), n
    # the eval-mode engine head folds the STEPPED weights (the step happened behind torch's version counters)
    enc.eval()
    ref.eval()
    ref.use_engine_head = False
    hidden = torch.from_numpy(synth.make_t5_hidden(4 * 6, 9, seed=5)).cuda()
    with torch.no_grad():
        a, b = enc.head(hidden, 4), ref.head(hidden, 4)
    assert float((a - b).abs().max()) < 2e-5 * max(1.0, float(b.abs().max()))


def test_optimizer_routes_the_head_to_the_engine_and_steps_like_torch():
    """``text2loc_amd.optim.Adam`` over a CellRetrievalNetwork with the real LanguageEncoder head: the head's 28 trained tensors form the
    "text_head" group (stepped by the engine), and two training steps follow a twin whose head is stepped by torch.optim.Adam
    (``text_engine=False``): same losses, same parameters to float32 rounding of Adam's normalised update."""
    from tests.test_gpu_train_loop import _args
    from tests.test_host_logic import make_objects
    from text2loc_amd.cell_retrieval import CellRetrievalNetwork
    from text2loc_amd.losses import ContrastiveLoss
    from text2loc_amd.optim import Adam

    B, S, L = 16, 6, 8
    cells = synth.make_cells(B, seed=3)
    objects = make_objects(cells, 3)

    def build(text_engine):
        enc = _encoder(7)
        _no_dropout(enc)
        model = CellRetrievalNetwork(synth.KNOWN_CLASS, synth.COLOR_NAMES, _args(), language_encoder=enc)
        model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_object_branch_weights(8).items()}, strict=False)
        for layer in model.obj_inter_module:
            layer.dropout.p = layer.dropout1.p = layer.dropout2.p = 0.0
            layer.self_attn.dropout = 0.0
        model = model.to("cuda").train()
        return model, Adam(model, lr=2e-4, text_engine=text_engine)

    (ma, oa), (mb, ob) = build(True), build(False)
    kinds = [g["t2l_engine"] for g in oa.param_groups]
    assert kinds[0] is True and kinds[-1] == "text_head" and len(oa.param_groups[-1]["params"]) == 28
    assert all(g["t2l_engine"] != "text_head" for g in ob.param_groups)
    crit = ContrastiveLoss(0.1)
    la, lb = [], []
    for step in range(2):
        hidden = torch.from_numpy(synth.make_t5_hidden(B * S, L, seed=70 + step)).cuda()
        for model, opt, acc in ((ma, oa, la), (mb, ob, lb)):
            opt.zero_grad()
            anchor = torch.nn.functional.normalize(model.language_encoder.head(hidden, B))
            loss = crit(anchor, model.encode_objects(objects))
            loss.backward()
            opt.step()
            acc.append(float(loss.detach()))
    torch.cuda.synchronize()
    assert np.allclose(la, lb, rtol=1e-5), (la, lb)
    pb = dict(mb.named_parameters())
    for n, p in ma.named_parameters():
        if n.startswith("language_encoder.") and p.requires_grad and p.grad is not None:
            err = (p.detach() - pb[n].detach()).abs()
            # Adam's first steps are lr * g / (|g| + eps): the two runs' gradients differ in the last bits (float atomics), and an element
            # whose gradient is rounding noise (whole families here: biases in front of a BatchNorm / a softmax) may move by up to 2 lr per
            # step in either direction — bounded everywhere, tight in the median of the weight matrices
            assert float(err.max()) <= 4.2 * 2e-4, n
            if p.dim() == 2:
                assert float(err.median()) < 2e-6, (n, float(err.median()))
    sd = oa.state_dict()
    assert sd["t2l_text_engine"]["step"] == 2 and sd["t2l_text_engine"]["exp_avg"].numel() > 13_000_000


def test_optimizer_checkpoints_migrate_between_the_two_head_layouts():
    """A checkpoint written with the head inside ``torch.optim.Adam`` (``text_engine=False`` — also the layout of every checkpoint older
    than the engine-side head) resumes under ``text_engine=True`` with the head's moments and step moved into the engine, and the
    other way round; the resumed optimizer takes the same next step as the one that wrote the checkpoint. A checkpoint of another
    model raises instead of restarting Adam silently."""
    from tests.test_gpu_train_loop import _args
    from tests.test_host_logic import make_objects
    from text2loc_amd.cell_retrieval import CellRetrievalNetwork
    from text2loc_amd.losses import ContrastiveLoss
    from text2loc_amd.optim import Adam

    B, S, L = 16, 6, 8
    objects = make_objects(synth.make_cells(B, seed=3), 3)
    crit = ContrastiveLoss(0.1)

    def build(text_engine):
        enc = _encoder(7)
        _no_dropout(enc)
        model = CellRetrievalNetwork(synth.KNOWN_CLASS, synth.COLOR_NAMES, _args(), language_encoder=enc)
        model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_object_branch_weights(8).items()}, strict=False)
        for layer in model.obj_inter_module:
            layer.dropout.p = layer.dropout1.p = layer.dropout2.p = 0.0
            layer.self_attn.dropout = 0.0
        model = model.to("cuda").train()
        return model, Adam(model, lr=2e-4, text_engine=text_engine)

    def one_step(model, opt, seed):
        hidden = torch.from_numpy(synth.make_t5_hidden(B * S, L, seed=seed)).cuda()
        opt.zero_grad()
        loss = crit(torch.nn.functional.normalize(model.language_encoder.head(hidden, B)), model.encode_objects(objects))
        loss.backward()
        opt.step()
        return float(loss.detach())

    for writer_engine in (False, True):
        mw, ow = build(writer_engine)
        for step in range(2):
            one_step(mw, ow, 90 + step)
        # a decayed learning rate (the reference's ExponentialLR / StepLR, training/coarse.py:272-275): the resumed head must step at
        # the DECAYED rate whichever group list the checkpoint had ([obj, rest] against [obj, rest, text] / [obj, text])
        sched = torch.optim.lr_scheduler.ExponentialLR(ow, 0.5)
        sched.step()
        assert all(abs(g["lr"] - 1e-4) < 1e-12 for g in ow.param_groups)
        ckpt_model = {k: v.detach().clone() for k, v in mw.state_dict().items()}
        ckpt_opt = ow.state_dict()
        assert (ckpt_opt["t2l_text_engine"] is not None) == writer_engine
        mr, orr = build(not writer_engine)
        kinds_before = [g["t2l_engine"] for g in orr.param_groups]
        mr.load_state_dict(ckpt_model)
        orr.load_state_dict(ckpt_opt)
        assert [g["t2l_engine"] for g in orr.param_groups] == kinds_before          # markers are the reader's own
        assert all(abs(g["lr"] - 1e-4) < 1e-12 for g in orr.param_groups), [g["lr"] for g in orr.param_groups]
        assert [g["t2l_engine"] for g in orr.state_dict()["param_groups"]] == kinds_before
        # the head's moments arrived on the other side, element for element
        head = [("language_encoder." + n, p) for n, p in mw.language_encoder.engine_optimizer_params()]
        if writer_engine:  # engine -> torch state
            flat_m = ckpt_opt["t2l_text_engine"]["exp_avg"]
            rest_names = [n for n, _, k in orr._non_obj if k == "rest"]
            off = 0
            for n, p in head:
                st = orr._torch.state[orr._torch.param_groups[0]["params"][rest_names.index(n)]]
                assert float(st["step"]) == 2.0 and torch.equal(st["exp_avg"].reshape(-1).cpu(), flat_m[off:off + p.numel()].cpu()), n
                off += p.numel()
        else:              # torch state -> engine
            m, v, st = mr.language_encoder._th_train_engine.text_adam_state()
            assert st == 2
            names_w = [n for n, _, _ in ow._non_obj]
            want = torch.cat([ow._torch.state[ow._torch.param_groups[0]["params"][names_w.index(n)]]["exp_avg"].reshape(-1) for n, _ in head])
            assert torch.equal(m.cpu(), want.cpu()) and float(m.abs().max()) > 0
        # and the next step is the writer's next step
        lw, lr_ = one_step(mw, ow, 99), one_step(mr, orr, 99)
        torch.cuda.synchronize()
        assert abs(lw - lr_) <= 1e-5 * max(1.0, abs(lw))
        pr = dict(mr.named_parameters())
        for n, p in mw.named_parameters():
            if n.startswith("language_encoder.") and p.requires_grad and p.dim() == 2:
                assert float((p.detach() - pr[n].detach()).abs().median()) < 2e-6, n
    # a checkpoint that is not this model's: loud
    bad = ow.state_dict()
    bad["t2l_text_engine"] = dict(bad["t2l_text_engine"], exp_avg=bad["t2l_text_engine"]["exp_avg"][:-5], exp_avg_sq=bad["t2l_text_engine"]["exp_avg_sq"][:-5])
    with pytest.raises(ValueError):
        build(True)[1].load_state_dict(bad)
