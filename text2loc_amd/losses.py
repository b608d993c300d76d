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


class _GatherRowsFn(torch.autograd.Function):
    """all_gather of the ranks' [B,D] rows into [W*B,D] (rank-major), differentiable: the backward hands every rank the slice
    of the incoming gradient that belongs to ITS rows — no communication. With a loss every rank evaluates identically on the
    gathered rows, that slice is d(global loss)/d(local rows); parameter gradients then only need a SUM over the ranks
    (``optim.Adam(group=..., grad_reduce="sum")``)."""

    @staticmethod
    def forward(ctx, x, group):
        import torch.distributed as dist

        from .sharded import _all_gather

        world, rank = dist.get_world_size(group), dist.get_rank(group)
        ctx.lo, ctx.n = rank * x.shape[0], x.shape[0]
        out = x.new_empty((world * x.shape[0],) + tuple(x.shape[1:]))
        _all_gather(dist, out, x.detach().contiguous(), group)
        return out

    @staticmethod
    def backward(ctx, grad):
        return grad[ctx.lo:ctx.lo + ctx.n].contiguous(), None


def gather_rows_with_grad(x, group=None):
    """[B,D] per rank -> [W*B,D] on every rank, gradients flowing back to the local rows (every rank must hold B rows)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return x
    return _GatherRowsFn.apply(x, group)


class ContrastiveLoss(torch.nn.Module):
    """Symmetric InfoNCE; ``forward(im, s)`` as in the reference (both inputs are re-normalised inside).

    Data-parallel training (not in the reference, which is single-process): with ``group`` (or
    ``gather=True`` for the default process group) every rank all-gathers the [B,256] text and cell embeddings of ALL ranks
    — two collectives of W*B*1 KiB — and evaluates the loss of the GLOBAL W*B batch (every cell of every rank is a negative
    for every text), exactly the value one process would compute on the concatenated batch; the backward returns
    d(global loss)/d(local rows). Sum the parameter gradients over the ranks (``optim.Adam(model, group=...)``) and the step
    equals the single-process step on the W*B batch, except that BatchNorm statistics stay per rank."""

    def __init__(self, temperature: float = 1.0, group=None, gather: bool = False):
        super().__init__()
        self.temperature = temperature
        self.group = group
        self.gather = bool(gather) or group is not None

    def forward(self, im, s):
        if self.gather:
            im, s = gather_rows_with_grad(im, self.group), gather_rows_with_grad(s, self.group)
        return _ContrastiveFn.apply(im, s, float(self.temperature))


def _cosine_scores(im, s):
    im = im / torch.norm(im, dim=1, keepdim=True)
    s = s / torch.norm(s, dim=1, keepdim=True)
    return im @ s.t()


class PairwiseRankingLoss(torch.nn.Module):
    """--ranking_loss pairwise (training/losses.py:178-216, the reference's args default): sum of all in-batch hinge
    violations in both directions, divided by the batch size. Host-side bookkeeping on the [B,B] cosine matrix (2 MFLOP):
    plain torch ops on whatever device the embeddings live on; gradients reach the object branch through the engine's
    backward like any other loss on ``encode_objects``' output."""

    def __init__(self, margin: float = 1.0):
        super().__init__()
        self.margin = margin

    def forward(self, im, s):
        scores = _cosine_scores(im, s)
        diag = scores.diag()
        off = ~torch.eye(len(im), dtype=torch.bool, device=scores.device)
        cost_s = torch.clamp(self.margin - diag[:, None] + scores, min=0) * off   # vs the other cells of each text
        cost_im = torch.clamp(self.margin - diag[None, :] + scores, min=0) * off  # vs the other texts of each cell
        return (cost_s.sum() + cost_im.sum()) / len(im)


class HardestRankingLoss(torch.nn.Module):
    """--ranking_loss hardest, the definition in force (the module's SECOND HardestRankingLoss, training/losses.py:286-355,
    shadows the first): the pairwise hinge costs, but only the largest violation of every row counts, times ``scale``."""

    def __init__(self, margin: float = 1.0, scale: float = 64.0):
        super().__init__()
        self.margin = margin
        self.scale = scale

    def forward(self, images, captions):
        scores = _cosine_scores(images, captions)
        diag = scores.diag()
        off = ~torch.eye(len(images), dtype=torch.bool, device=scores.device)
        cost_s = torch.clamp(self.margin - diag[None, :] + scores, min=0) * off   # [i][j]: text j's own cell vs cell i
        cost_im = torch.clamp(self.margin - diag[:, None] + scores, min=0) * off  # [i][j]: cell i's own text vs text j
        return (cost_im.max(dim=1)[0].mean() + cost_s.max(dim=1)[0].mean()) * self.scale
