"""GPU tier: fused ContrastiveLoss fwd+bwd kernel (C ABI) vs the reference's golden loss/grads and the oracle.
Tolerance (fp32, exp of |x| <= 10): 3e-5 relative on the loss, 3e-6 absolute on gradients (|g| <= ~0.2)."""
import numpy as np
import pytest
import torch

from oracle import t2l_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from text2loc_amd.engine import Engine

    e = Engine(0)
    yield e
    e.close()


def _run(eng, a, p, t):
    import torch

    loss, ga, gp = eng.contrastive_loss(torch.from_numpy(a).cuda(), torch.from_numpy(p).cuda(), t)
    torch.cuda.synchronize()
    return float(loss.item()), ga.cpu().numpy(), gp.cpu().numpy()


def test_reference_golden(eng, golden):
    g = golden("loss")
    loss, ga, gp = _run(eng, g["anchor"], g["positive"], float(g["temperature"]))
    assert abs(loss - float(g["loss"])) < 3e-5 * max(1.0, abs(float(g["loss"])))
    assert np.abs(ga - g["grad_anchor"]).max() < 3e-6
    assert np.abs(gp - g["grad_positive"]).max() < 3e-6


@pytest.mark.parametrize("B", [1, 2, 31, 33, 64, 100, 128])
def test_batches_vs_oracle(eng, B):
    rng = np.random.default_rng(B)
    a = rng.standard_normal((B, 256)).astype(np.float32)
    p = (a + 0.8 * rng.standard_normal((B, 256))).astype(np.float32)
    rl, rga, rgp = O.contrastive_loss(a, p, 0.1, dtype=np.float64)
    loss, ga, gp = _run(eng, a, p, 0.1)
    assert abs(loss - rl) < 3e-5 * max(1.0, abs(rl))
    assert np.abs(ga - rga).max() < 3e-6
    assert np.abs(gp - rgp).max() < 3e-6


@pytest.mark.parametrize("B", [1, 31, 33, 64, 65, 100, 128])
def test_odd_batches_and_forward_only_calls(eng, B):
    """Batches that are not a multiple of the 32-row tile, scaled inputs, and forward-only calls (no gradient buffers: one workgroup)
    give the oracle's loss and gradients."""
    rng = np.random.default_rng(100 + B)
    a = (3.0 * rng.standard_normal((B, 256))).astype(np.float32)
    p = (0.5 * a + 0.4 * rng.standard_normal((B, 256))).astype(np.float32)
    rl, rga, rgp = O.contrastive_loss(a, p, 0.1, dtype=np.float64)
    loss, ga, gp = _run(eng, a, p, 0.1)
    lf, _, _ = eng.contrastive_loss(torch.from_numpy(a).cuda(), torch.from_numpy(p).cuda(), 0.1, need_grad=False)
    assert abs(float(lf.item()) - loss) < 1e-6 * max(1.0, abs(rl))
    assert abs(loss - rl) < 3e-5 * max(1.0, abs(rl))
    assert np.abs(ga - rga).max() < 1e-4 * np.abs(rga).max() + 3e-6 and np.abs(gp - rgp).max() < 1e-4 * np.abs(rgp).max() + 3e-6  # (B = 1: all gradients are 0)


def test_autograd_function(eng):
    """text2loc_amd.losses.ContrastiveLoss plugs into autograd like the reference's nn.Module."""
    import torch
    from text2loc_amd.losses import ContrastiveLoss

    rng = np.random.default_rng(0)
    a = torch.from_numpy(rng.standard_normal((64, 256)).astype(np.float32)).cuda().requires_grad_()
    p = torch.from_numpy(rng.standard_normal((64, 256)).astype(np.float32)).cuda().requires_grad_()
    loss = ContrastiveLoss(temperature=0.1)(a, p)
    (2.0 * loss).backward()
    rl, rga, rgp = O.contrastive_loss(a.detach().cpu().numpy(), p.detach().cpu().numpy(), 0.1, dtype=np.float64)
    assert abs(loss.item() - rl) < 3e-5 * abs(rl)
    assert np.abs(a.grad.cpu().numpy() - 2 * rga).max() < 6e-6
    assert np.abs(p.grad.cpu().numpy() - 2 * rgp).max() < 6e-6


@pytest.mark.parametrize("B", [129, 200, 512, 1024])
def test_global_batches_beyond_the_fused_kernel(B):
    """B > 128 — the all-gathered W x B batch of data-parallel training (8 x 64 = 512) — runs as a chain of launches over a
    [B][B] matrix in HBM: same value and gradients as the float64 restatement of training/losses.py:269-283."""
    from text2loc_amd.engine import Engine

    rs = np.random.default_rng(B)
    a = rs.standard_normal((B, 256)).astype(np.float32)
    p = (a + 0.7 * rs.standard_normal((B, 256))).astype(np.float32) * rs.uniform(0.5, 2.0, size=(B, 1)).astype(np.float32)
    eng = Engine(0)
    loss, ga, gp = eng.contrastive_loss(torch.from_numpy(a).cuda(), torch.from_numpy(p).cuda(), 0.1)
    rl, rga, rgp = O.contrastive_loss(a, p, 0.1, dtype=np.float64)
    assert abs(float(loss.item()) - rl) < 3e-5 * abs(rl)
    assert np.abs(ga.cpu().numpy() - rga).max() < 3e-6 and np.abs(gp.cpu().numpy() - rgp).max() < 3e-6
    l2, g2, _ = eng.contrastive_loss(torch.from_numpy(a).cuda(), torch.from_numpy(p).cuda(), 0.1, need_grad=False)
    assert g2 is None and abs(float(l2.item()) - float(loss.item())) < 1e-7
    with pytest.raises(Exception, match="1024"):
        eng.contrastive_loss(torch.zeros(1025, 256, device="cuda"), torch.zeros(1025, 256, device="cuda"), 0.1)
    eng.close()
