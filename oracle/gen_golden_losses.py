"""TEST INFRASTRUCTURE — build-container only.

Golden vectors for the reference's other in-batch ranking losses (training/losses.py:178-253): PairwiseRankingLoss and
HardestRankingLoss (selected by --ranking_loss pairwise|hardest, training/coarse.py:263-266) — value and autograd
gradients on seeded embeddings -> tests/golden/loss_ranking.npz. (PairwiseRankingLoss calls .cuda() on a zeros tensor;
there is no GPU in the build container, so Tensor.cuda is an identity while the reference runs here.)
"""
from __future__ import annotations

import os.path as osp
import sys

import numpy as np

HERE = osp.dirname(osp.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as H  # noqa: E402

H.setup_reference_imports()

import torch  # noqa: E402

torch.Tensor.cuda = lambda self, *a, **k: self  # the reference hard-codes .cuda() (losses.py:203,208)

from training.losses import HardestRankingLoss, PairwiseRankingLoss  # noqa: E402


def main():
    rng = np.random.default_rng(11)
    B, D = 24, 256
    base = rng.standard_normal((B, D))
    im = (0.45 * base + rng.standard_normal((B, D))).astype(np.float32) * 1.7  # un-normalised on purpose
    s = (0.45 * base + rng.standard_normal((B, D))).astype(np.float32) * 0.6
    out = {"im": im, "s": s, "margin": np.float32(0.35)}
    for name, cls in (("pairwise", PairwiseRankingLoss), ("hardest", HardestRankingLoss)):
        a = torch.tensor(im, requires_grad=True)
        b = torch.tensor(s, requires_grad=True)
        loss = cls(margin=0.35)(a, b)
        loss.backward()
        out[name + "_loss"] = np.float64(loss.item())
        out[name + "_grad_im"] = a.grad.numpy()
        out[name + "_grad_s"] = b.grad.numpy()
        print(name, loss.item())
    np.savez_compressed(osp.join(H.REPO, "tests", "golden", "loss_ranking.npz"), **out)


if __name__ == "__main__":
    main()
