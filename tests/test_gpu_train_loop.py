"""GPU tier: the drop-in TRAINING surface end to end — ``CellRetrievalNetwork.train()`` + ``encode_objects`` (engine
forward) + ``ContrastiveLoss`` + ``loss.backward()`` (engine backward) + ``text2loc_amd.optim.Adam`` — against a plain
PyTorch fp32 replica of the same graph built from the SAME nn.Modules (torch autograd + torch.optim.Adam), i.e. what
the reference's train_epoch computes (training/coarse.py:31-58), for several steps, then eval-mode agreement."""
import argparse
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from text2loc_amd import packing, synth
from tests.test_host_logic import make_objects

pytestmark = pytest.mark.gpu


class TableText(torch.nn.Module):
    """Trainable stand-in for the text branch: one embedding row per description id."""

    def __init__(self, n, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.table = torch.nn.Parameter(torch.randn(n, 256, generator=g))

    def forward(self, idx):
        return self.table[torch.as_tensor(idx, device=self.table.device)]

    @property
    def device(self):
        return self.table.device


def _args(**kw):
    a = argparse.Namespace(coarse_embed_dim=256, object_size=28, object_inter_module_num_heads=4,
                           object_inter_module_num_layers=2, hungging_model=None, fixed_embedding=True,
                           intra_module_num_layers=1, intra_module_num_heads=4, inter_module_num_layers=1,
                           inter_module_num_heads=4, class_embed=True, color_embed=True,
                           use_features=["class", "color", "position", "num"], ranking_loss="contrastive",
                           top_k=[1, 3, 5], threshs=[5, 10, 15], batch_size=16, temperature=0.1)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def torch_encode_objects(model, packed, pn=None):
    """models/object_encoder.py:66-153 + models/cell_retrieval.py:65-110 with torch ops on the model's own modules."""
    oe = model.object_encoder
    dev = model.device
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in packed.items()}
    a = model.args
    emb = []
    if a.class_embed:
        emb.append(F.normalize(oe.class_embedding(t["class_idx"].long()), dim=-1))
    else:
        emb.append(F.normalize(oe.mlp_pointnet(pn), dim=-1))
    if a.color_embed:
        emb.append(F.normalize(oe.color_embedding(t["color_idx"].long()), dim=-1))
    else:
        emb.append(F.normalize(oe.color_encoder(t["rgb"]), dim=-1))
    emb.append(F.normalize(oe.pos_encoder(t["center"]), dim=-1))
    emb.append(F.normalize(oe.num_encoder((t["n_pts"].unsqueeze(-1) - 1826.6844940968194) / 2516.8905096993817), dim=-1))
    e = F.normalize(oe.mlp_merge(torch.cat(emb, dim=-1)), dim=-1)
    B = len(packed["counts"])
    x = torch.zeros(B, 28, 256, device=dev)
    for i in range(B):
        lo, n = int(packed["offsets"][i]), min(int(packed["counts"][i]), 28)
        x[i, :n] = e[lo:lo + n]
    x = x.permute(1, 0, 2).contiguous()
    for layer in model.obj_inter_module:
        x = layer(x)
    return F.normalize(x.max(dim=0)[0])


def torch_contrastive(im, s, t=0.1):
    im = im / torch.norm(im, dim=1, keepdim=True)
    s = s / torch.norm(s, dim=1, keepdim=True)
    sim = im @ s.t()
    num = torch.exp(torch.diag(sim) / t)
    den = torch.exp(sim / t)
    return torch.mean(-torch.log(num / den.sum(0)) - torch.log(num / den.sum(1)))


@pytest.mark.parametrize("embed", [True, False])
def test_training_loop_tracks_the_torch_replica(embed):
    from text2loc_amd.cell_retrieval import CellRetrievalNetwork
    from text2loc_amd.losses import ContrastiveLoss
    from text2loc_amd.optim import Adam

    B, STEPS, LR = 16, 6, 1e-3
    args = _args(class_embed=embed, color_embed=embed)
    batches = []
    for s in range(2):
        cells = synth.make_cells(B, seed=40 + s, with_pn_feat=True)
        batches.append((cells, make_objects(cells, 40 + s)))
    model = CellRetrievalNetwork(synth.KNOWN_CLASS, synth.COLOR_NAMES, args, language_encoder=TableText(2 * B, 5))
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_object_branch_weights(6).items()},
                          strict=False)
    for layer in model.obj_inter_module:  # torch's dropout stream cannot be replayed: compare with the sites off
        layer.dropout.p = layer.dropout1.p = layer.dropout2.p = 0.0
        layer.self_attn.dropout = 0.0
    model = model.to("cuda")
    ref = copy.deepcopy(model)
    opt = Adam(model, lr=LR)
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, 0.9)  # the reference's default scheduler drives it unchanged
    opt_ref = torch.optim.Adam(ref.parameters(), lr=LR)
    sched_ref = torch.optim.lr_scheduler.ExponentialLR(opt_ref, 0.9)
    crit = ContrastiveLoss(temperature=0.1)
    model.train()
    ref.train()
    losses, losses_ref = [], []
    for step in range(STEPS):
        cells, objects = batches[step % 2]
        texts = list(range((step % 2) * B, (step % 2) * B + B))
        pn = [torch.from_numpy(cells["pn_feat"][cells["offsets"][i]:cells["offsets"][i + 1]]) for i in range(B)]
        opt.zero_grad()
        loss = crit(model.encode_text(texts), model.encode_objects(objects, None if embed else pn))
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))

        opt_ref.zero_grad()
        packed = packing.pack_cells(objects, model.object_encoder.known_classes, model.object_encoder.known_colors)
        pos = torch_encode_objects(ref, packed, None if embed else torch.from_numpy(cells["pn_feat"]).cuda())
        loss_ref = torch_contrastive(ref.encode_text(texts), pos)
        loss_ref.backward()
        opt_ref.step()
        losses_ref.append(float(loss_ref.detach()))
        if step == 2:
            sched.step()
            sched_ref.step()
    # Step 0 starts from identical state: tight. Afterwards the two float32 trainings drift apart slowly — Adam's
    # first steps are +-lr*sign(g), so every parameter whose true gradient is 0 (Linear biases in front of a BatchNorm,
    # key biases of the attention) random-walks on rounding noise, and ~0.02 % of the other elements flip sign — so the
    # curves are compared to 1 % and the parameters statistically.
    assert abs(losses[0] - losses_ref[0]) < 2e-5
    assert np.allclose(losses, losses_ref, rtol=1e-2, atol=0), (losses, losses_ref)
    assert losses[-1] < 0.6 * losses[0]
    sd, sd_ref = model.state_dict(), ref.state_dict()
    for n, v in sd_ref.items():
        if v.dtype != torch.float32 or n.startswith("object_encoder.pointnet"):
            assert torch.equal(sd[n], v), n  # num_batches_tracked
            continue
        if (n.startswith("object_encoder.") and n.endswith((".0.bias", ".1.running_mean"))) or n.endswith("in_proj_bias"):
            continue  # (partly) zero true gradient, see above; running_mean carries the drifting Linear bias
        err = (sd[n] - v).abs()
        assert float((err < 1e-3 * (1 + v.abs())).float().mean()) > 0.95 and float((err / (1 + v.abs())).max()) < 2.5 * LR * STEPS, (n, float(err.max()))
    # eval mode after training: the parameters and running statistics the ENGINE updated in place must reach the fused
    # eval encoder (re-folded BatchNorm) — compare with torch evaluating the model's own modules on the same memory
    model.eval()
    cells, objects = batches[0]
    pn = [torch.from_numpy(cells["pn_feat"][cells["offsets"][i]:cells["offsets"][i + 1]]) for i in range(B)]
    with torch.no_grad():
        out = model.encode_objects(objects, None if embed else pn)
        packed = packing.pack_cells(objects, model.object_encoder.known_classes, model.object_encoder.known_colors)
        out_ref = torch_encode_objects(model, packed, None if embed else torch.from_numpy(cells["pn_feat"]).cuda())
    assert float((out - out_ref).abs().max()) < 2e-5


def test_train_epoch_runs_and_pn_feature_gradient_flows():
    from text2loc_amd.cell_retrieval import CellRetrievalNetwork
    from text2loc_amd.coarse import collate_fn, train_epoch
    from text2loc_amd.losses import ContrastiveLoss
    from text2loc_amd.optim import Adam

    B = 8
    args = _args(class_embed=False, color_embed=False)
    cells = synth.make_cells(2 * B, seed=3, with_pn_feat=True)
    objects = make_objects(cells, 3)
    model = CellRetrievalNetwork(synth.KNOWN_CLASS, synth.COLOR_NAMES, args, language_encoder=TableText(2 * B, 1)).to("cuda")
    feats = [torch.from_numpy(cells["pn_feat"][cells["offsets"][i]:cells["offsets"][i + 1]]).cuda() for i in range(2 * B)]

    class Ds(torch.utils.data.Dataset):
        def __len__(self):
            return 2 * B

        def __getitem__(self, i):
            return {"texts": i, "objects": objects[i], "object_points": feats[i]}

    dl = torch.utils.data.DataLoader(Ds(), batch_size=B, collate_fn=collate_fn, shuffle=False)
    opt = Adam(model, lr=1e-3)
    torch.manual_seed(0)
    l0, _ = train_epoch(model, dl, args, opt, ContrastiveLoss(0.1))  # dropout on (p = 0.1)
    l1, _ = train_epoch(model, dl, args, opt, ContrastiveLoss(0.1))
    assert np.isfinite(l0) and np.isfinite(l1) and l1 < l0
    bn = model.object_encoder.mlp_merge[0][1]
    assert int(bn.num_batches_tracked) == 4
    # gradient w.r.t. the PointNet features (the seam a PyTorch PointNet++ would hang on)
    model.train()
    pn = [f.clone().requires_grad_(True) for f in feats[:B]]
    out = model.encode_objects(objects[:B], pn)
    out.square().sum().backward()  # unit rows: constant loss, so the gradient must vanish
    assert all(p.grad is not None for p in pn)
    assert max(float(p.grad.abs().max()) for p in pn) < 1e-5
    out = model.encode_objects(objects[:B], pn)
    (out * torch.randn_like(out)).sum().backward()
    assert max(float(p.grad.abs().max()) for p in pn) > 1e-6
    with pytest.raises(Exception, match="stale"):
        a = model.encode_objects(objects[:B], pn)
        model.encode_objects(objects[:B], pn)
        a.sum().backward()


def test_adam_state_survives_rebinds_and_checkpoints():
    """The engine owns exp_avg / exp_avg_sq / step of the object branch: a re-bind of the same model (new .grad storage)
    keeps them, and optimizer.state_dict() / load_state_dict() round-trip them (resume gives the same next step)."""
    from text2loc_amd.cell_retrieval import CellRetrievalNetwork
    from text2loc_amd.losses import ContrastiveLoss
    from text2loc_amd.optim import Adam

    B = 8
    args = _args()
    cells = synth.make_cells(B, seed=77)
    objects = make_objects(cells, 77)
    sd0 = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_object_branch_weights(6).items()}

    def fresh():
        m = CellRetrievalNetwork(synth.KNOWN_CLASS, synth.COLOR_NAMES, args, language_encoder=TableText(B, 5))
        m.load_state_dict(sd0, strict=False)
        for layer in m.obj_inter_module:
            layer.dropout.p = layer.dropout1.p = layer.dropout2.p = 0.0
            layer.self_attn.dropout = 0.0
        return m.to("cuda").train()

    crit = ContrastiveLoss(temperature=0.1)

    def one_step(m, opt):
        opt.zero_grad()
        loss = crit(m.encode_text(list(range(B))), m.encode_objects(objects, None))
        loss.backward()
        opt.step()

    model = fresh()
    opt = Adam(model, lr=1e-3)
    one_step(model, opt)
    one_step(model, opt)
    m2, v2, step2 = model.train_engine().adam_state()
    assert step2 == 2 and float(v2.abs().max()) > 0
    # re-bind: hand one parameter a NEW gradient buffer (what an external .grad assignment does)
    p = model.obj_inter_module[0].linear1.weight
    p.grad = torch.zeros_like(p)
    eng = model.train_engine()
    m2b, v2b, step2b = eng.adam_state()
    assert step2b == 2 and torch.equal(m2, m2b) and torch.equal(v2, v2b)
    # checkpoint after 2 steps -> resume in a new model/optimizer -> the third step matches the uninterrupted run
    ckpt_model = {k: v.detach().clone() for k, v in model.state_dict().items()}
    ckpt_opt = opt.state_dict()
    one_step(model, opt)
    want = {k: v.detach().clone() for k, v in model.state_dict().items()}
    resumed = fresh()
    resumed.load_state_dict(ckpt_model)
    opt_r = Adam(resumed, lr=1e-3)
    opt_r.load_state_dict(ckpt_opt)
    assert resumed.train_engine().adam_state()[2] == 2
    one_step(resumed, opt_r)
    for k in ("obj_inter_module.1.linear2.weight", "object_encoder.mlp_merge.0.0.weight", "object_encoder.class_embedding.weight"):
        assert torch.allclose(resumed.state_dict()[k], want[k], rtol=0, atol=2e-6), k  # float atomics in the gradients
    # a DIFFERENT model on a fresh engine starts from zero
    assert fresh().train_engine().adam_state()[2] == 0
