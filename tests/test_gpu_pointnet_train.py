"""GPU tier: the PointNet++ object backbone under model.train() (SURVEY.md §8 rows a3 + a9) through the C ABI —
t2l_pointnet_features_train / t2l_pointnet_backward — against the float64 oracle oracle/t2l_oracle_pointnet_train.py
(per-cell BatchNorm statistics, sequential running-statistics updates, max aggregation with arg-max replay).
PARITY UNPINNED like the eval path (the index structure is this build's reading of torch_cluster / PyG): what is checked
here is self-consistency of the HIP kernels with the build's own restatement, whose backward is checked against central
differences in tests/test_oracle_train.py."""
import numpy as np
import pytest
import torch

from oracle import t2l_oracle_pointnet_train as OPT
from text2loc_amd import synth

pytestmark = pytest.mark.gpu

P = "object_encoder.pointnet."


@pytest.fixture(scope="module")
def eng():
    from text2loc_amd.engine import Engine

    e = Engine(0)
    yield e
    e.close()


def bind_all(eng, sd_obj, sd_pn):
    tensors = {}
    for k, v in list(sd_obj.items()) + list(sd_pn.items()):
        if k.endswith("num_batches_tracked") or k.endswith("_embedding.weight") or "classifier" in k:
            continue
        t = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).cuda()
        tensors[k] = (t, None if "running_" in k else torch.zeros_like(t))
    eng.train_bind(tensors, class_embed=False, color_embed=False)
    return tensors


def check(tensors, name, exp, med=1e-2, fro=0.03):
    """-> True when the tensor agrees to float32 rounding (1e-4 of its norm)."""
    got = tensors[name][1].cpu().numpy().astype(np.float64).ravel()
    exp = np.asarray(exp, dtype=np.float64).ravel()
    rms = max(np.sqrt((exp ** 2).mean()), 1e-30)
    err = np.abs(got - exp)
    if name.endswith(".0.bias") and "lin" not in name:  # Linear bias in front of a BatchNorm: true gradient 0
        scale = np.abs(tensors[name.replace(".0.bias", ".1.bias")][1].cpu().numpy()).max()
        assert np.abs(got).max() < 2e-3 * max(1.0, scale), (name, np.abs(got).max(), scale)
        return False
    # float32 MFMA kernels against float64. Measured: ~1e-6 of the norm on every tensor, EXCEPT upstream of a discrete decision
    # (arg-max row, ReLU sign, 32nd neighbour) that falls the other way within float32 rounding: one such flip in SA3 of the
    # ragged case moves everything upstream by ~1% of its norm. A wrong formula is O(1) everywhere.
    assert np.median(err) < med * rms + 1e-9, (name, np.median(err), rms)
    ratio = np.sqrt((err ** 2).sum()) / max(np.sqrt((exp ** 2).sum()), 1e-30)
    assert ratio < fro, (name, ratio)
    return ratio < 1e-4


@pytest.mark.parametrize("n_cells,min_obj,max_obj,self_loops", [(2, 2, 2, 1), (3, 1, 4, 1), (2, 3, 3, 0)])
def test_pointnet_train_forward_backward_match_the_float64_oracle(eng, n_cells, min_obj, max_obj, self_loops):
    cells = synth.make_cells(n_cells, seed=3 + n_cells, min_obj=min_obj, max_obj=max_obj)
    pos, rgb = synth.make_sampled_points(cells, 3)
    sd_pn = synth.make_pointnet_weights(1)
    sd_obj = synth.make_object_branch_weights(2)
    offs = np.asarray(cells["offsets"], dtype=np.int32)
    n = pos.shape[0]
    eng.set_option("pointnet_pyg_self_loops", self_loops)
    tensors = bind_all(eng, sd_obj, sd_pn)
    rng = np.random.default_rng(0)
    R = rng.standard_normal((n, 256))
    f2_ref, info = OPT.forward_backward(pos, rgb, offs, sd_pn, grad_f2=R, pyg_self_loops=bool(self_loops))

    dpos, drgb = torch.from_numpy(pos).cuda(), torch.from_numpy(rgb).cuda()
    f2 = eng.pointnet_features_train(dpos, drgb, offs)
    torch.cuda.synchronize()
    got = f2.cpu().numpy().astype(np.float64)
    scale = np.abs(f2_ref).max()
    assert np.abs(got - f2_ref).max() < 2e-4 * scale, (np.abs(got - f2_ref).max(), scale)
    # running statistics: one momentum update per cell, in cell order
    for k, v in info["running"].items():
        r = tensors[k][0].cpu().numpy().astype(np.float64)
        assert np.abs(r - v).max() < 2e-5 * max(1.0, np.abs(v).max()), (k, np.abs(r - v).max())
        assert np.abs(v - np.asarray(sd_pn[k], dtype=np.float64)).max() > 0  # they did move
    eng.zero_grad()
    eng.pointnet_backward(torch.from_numpy(R.astype(np.float32)).cuda())
    torch.cuda.synchronize()
    assert sorted(info["grads"].keys()) == sorted(k for k in tensors if k.startswith(P) and tensors[k][1] is not None)
    tight = sum(check(tensors, name, g) for name, g in info["grads"].items())
    assert tight >= 10, tight  # (28 tensors are not Linear biases in front of a BatchNorm; a flip leaves its downstream side tight)
    eng.set_option("pointnet_pyg_self_loops", 1)


def test_pointnet_train_backward_accumulates_and_repeats(eng):
    """Two backward calls double the gradients (they ADD, as autograd does); a second forward gives identical features."""
    cells = synth.make_cells(2, seed=9, min_obj=2, max_obj=3)
    pos, rgb = synth.make_sampled_points(cells, 1)
    tensors = bind_all(eng, synth.make_object_branch_weights(2), synth.make_pointnet_weights(4))
    offs = np.asarray(cells["offsets"], dtype=np.int32)
    dpos, drgb = torch.from_numpy(pos).cuda(), torch.from_numpy(rgb).cuda()
    g = torch.randn(pos.shape[0], 256, device="cuda")
    a = eng.pointnet_features_train(dpos, drgb, offs).clone()
    eng.zero_grad()
    eng.pointnet_backward(g)
    one = {k: v[1].clone() for k, v in tensors.items() if k.startswith(P) and v[1] is not None}
    eng.pointnet_backward(g)
    for k, v in one.items():
        two = tensors[k][1]
        if k.endswith(".0.bias") and "lin" not in k:  # in front of a BatchNorm: atomics-ordered float32 noise around 0
            continue
        assert torch.allclose(two, 2 * v, rtol=1e-3, atol=1e-5 * float(v.abs().max()) + 1e-9), k
    b = eng.pointnet_features_train(dpos, drgb, offs)
    assert torch.equal(a, b)  # batch statistics do not depend on the running statistics; every kernel is deterministic in forward


def test_pointnet_train_needs_the_backbone_bound(eng):
    from text2loc_amd.engine import T2LError

    sd = synth.make_object_branch_weights(2)
    tensors = {}
    for k, v in sd.items():
        if k.endswith("num_batches_tracked") or k.endswith("_embedding.weight"):
            continue
        t = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).cuda()
        tensors[k] = (t, None if "running_" in k else torch.zeros_like(t))
    eng.train_bind(tensors, class_embed=False, color_embed=False)
    pos = torch.zeros(1, 256, 3, device="cuda")
    with pytest.raises(T2LError, match="bind the object_encoder.pointnet"):
        eng.pointnet_features_train(pos, pos, np.array([0, 1], dtype=np.int32))
    # a partial binding is refused
    pn = synth.make_pointnet_weights(1)
    k = P + "lin2.weight"
    t = torch.from_numpy(pn[k]).cuda()
    tensors[k] = (t, torch.zeros_like(t))
    with pytest.raises(T2LError, match="completely"):
        eng.train_bind(tensors, class_embed=False, color_embed=False)
