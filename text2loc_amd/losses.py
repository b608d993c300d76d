"""ContrastiveLoss with the reference's constructor/forward signature (training/losses.py:255-283),
computed by the fused HIP forward+backward kernel (libt2l.so: t2l_contrastive_loss)."""
from __future__ import annotations

import torch

from .engine import Engine

_engines = {}


def _engine_for(device: torch.device) -> Engine:
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _engines:
        _engines[idx] = Engine(idx)
    return _engines[idx]


class _ContrastiveFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, im, s, temperature):
        eng = _engine_for(im.device)
        need = im.requires_grad or s.requires_grad
        loss, ga, gp = eng.contrastive_loss(im.detach().contiguous().float(), s.detach().contiguous().float(),
                                            temperature, need_grad=need)
        if need:
            ctx.save_for_backward(ga, gp)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, grad_out):
        ga, gp = ctx.saved_tensors
        return grad_out * ga, grad_out * gp, None


class ContrastiveLoss(torch.nn.Module):
    """Symmetric InfoNCE; ``forward(im, s)`` as in the reference (both inputs are re-normalised inside)."""

    def __init__(self, temperature: float = 1.0):
        super().__init__()
        self.temperature = temperature

    def forward(self, im, s):
        return _ContrastiveFn.apply(im, s, float(self.temperature))
