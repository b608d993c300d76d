"""dev: t2l_contrastive_loss (forward + analytic backward), the 4*nb-workgroup kernel against the single-workgroup one: wall time per
call in a stream-ordered loop and the largest difference of loss / gradients against a float64 torch evaluation.  python tools/loss_probe.py"""
import sys
import time

import torch

sys.path.insert(0, ".")
from text2loc_amd.engine import Engine  # noqa: E402


def ref(a, p, temp):
    a = a.double().requires_grad_(True)
    p = p.double().requires_grad_(True)
    an, pn = torch.nn.functional.normalize(a, dim=1), torch.nn.functional.normalize(p, dim=1)
    sim = an @ pn.T / temp
    e = sim.exp()
    loss = (e.sum(0).log() + e.sum(1).log() - 2 * sim.diag()).mean()
    loss.backward()
    return loss.detach(), a.grad, p.grad


eng = Engine(0)
for B in (64, 32, 1, 33, 100, 128):
    torch.manual_seed(B)
    a = torch.randn(B, 256, device="cuda") * 3
    p = (a + 0.7 * torch.randn(B, 256, device="cuda")) * 0.5
    rl, ra, rp = ref(a, p, 0.1)
    line = f"B={B:4d}"
    for single in (1, 0):
        eng.set_option("loss_single_wg", single)
        for _ in range(20):
            l, ga, gp = eng.contrastive_loss(a, p, 0.1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            eng.contrastive_loss(a, p, 0.1)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 300 * 1e6
        l, ga, gp = eng.contrastive_loss(a, p, 0.1)
        lf, _, _ = eng.contrastive_loss(a, p, 0.1, need_grad=False)
        err = max(float((ga - ra).abs().max() / ra.abs().max()), float((gp - rp).abs().max() / rp.abs().max()))
        line += f" | {'single' if single else 'tiles '}: {us:6.1f} us  loss err {abs(float(l) - float(rl)):.2e} ({abs(float(lf) - float(rl)):.2e} fwd-only)  grad err {err:.2e}"
    print(line)
