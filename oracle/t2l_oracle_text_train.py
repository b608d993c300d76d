"""TEST INFRASTRUCTURE — CPU oracle for the TRAINING step of the text head (SURVEY.md §8 f-4 / the text half of a9).

Restates, in numpy (float64 by default), ``LanguageEncoder.forward`` downstream of the frozen T5's hidden states under
``model.train()`` (models/language_encoder.py:127-147 as run by training/coarse.py:44,55-56) and the gradients autograd produces
for it: TransformerEncoderLayer(1024, 4 heads, 4096) over the tokens of every sentence (:127-131) -> max over the tokens (:133) ->
inter_mlp = Linear(1024 -> 256) + BatchNorm1d in batch-statistics mode, no ReLU (:135; get_mlp2 :43-74) -> view [B, S, 256] (:136) ->
x += TransformerEncoderLayer(256, 4 heads, 1024)(x) over the S sentences (:141-144) -> max over the sentences (:146).
Built from the blocks of oracle/t2l_oracle_train.py (the same post-norm layer with its four dropout sites; the layer index selects
the dropout sites: 0 -> sites 0-3, 1 -> sites 4-7, the masks are this build's counter-based ones).

Only ``tests/`` may import this module. Pinning: tests/test_oracle_train.py checks it against tests/golden/train_step_text.npz (the
imported reference's own forward / backward with the dropout sites at p = 0, oracle/gen_golden_text_train.py) and against central
differences of its own forward.
"""
from __future__ import annotations

import numpy as np

from . import t2l_oracle_train as OT


def text_head_train(hidden: np.ndarray, sd: dict, n_desc: int, grad_out=None, p_drop: float = 0.0, seed: int = 0, dtype=np.float64,
                    prefix: str = "language_encoder."):
    """hidden f32[n_sent, L, 1024] (description-major) -> (out [n_desc, 256] — not normalised —, info). With ``grad_out``
    [n_desc, 256]: info["grads"][name] = dLoss/dparameter, info["bn_stats"]["<prefix>inter_mlp.0.1"] = (mean, biased var, n)."""
    t = OT._Tape(sd, dtype)
    w = t.w
    x = np.asarray(hidden).astype(dtype)
    n_sent, L, _ = x.shape
    S = n_sent // n_desc
    x2, c1 = OT._layer_fwd(t, x, prefix + "intra_module.0", 4, p_drop, seed, 0)            # language_encoder.py:127-131
    tok_arg = x2.argmax(axis=1)                                                              # :133 (first maximal token wins)
    pooled = np.take_along_axis(x2, tok_arg[:, None, :], axis=1)[:, 0]
    mp = prefix + "inter_mlp.0"
    y = pooled @ w[mp + ".0.weight"].T + w[mp + ".0.bias"]                                  # :135
    z, bc = OT._bn_fwd(y, w[mp + ".1.weight"], w[mp + ".1.bias"])
    t.bn_stats[mp + ".1"] = (bc[2], bc[3], y.shape[0])
    xi = z.reshape(n_desc, S, -1)                                                            # :136
    y2, c2 = OT._layer_fwd(t, xi, prefix + "inter_module.0", 4, p_drop, seed, 1)            # :141-144
    tot = xi + y2
    sent_arg = tot.argmax(axis=1)                                                            # :146
    out = np.take_along_axis(tot, sent_arg[:, None, :], axis=1)[:, 0]
    info = {"pooled": pooled, "sentence_vectors": z, "bn_stats": t.bn_stats}
    if grad_out is None:
        return out, info
    g = np.asarray(grad_out).astype(dtype)
    dtot = np.zeros_like(tot)
    np.put_along_axis(dtot, sent_arg[:, None, :], g[:, None, :], axis=1)
    dxi = dtot + OT._layer_bwd(t, dtot, prefix + "inter_module.0", 4, c2)
    dz = dxi.reshape(n_sent, -1)
    dy, dg, db = OT._bn_bwd(dz, w[mp + ".1.weight"], bc)
    t.add(mp + ".1.weight", dg)
    t.add(mp + ".1.bias", db)
    t.add(mp + ".0.weight", dy.T @ pooled)
    t.add(mp + ".0.bias", dy.sum(0))
    dpooled = dy @ w[mp + ".0.weight"]
    dx2 = np.zeros_like(x2)
    np.put_along_axis(dx2, tok_arg[:, None, :], dpooled[:, None, :], axis=1)
    OT._layer_bwd(t, dx2, prefix + "intra_module.0", 4, c1)
    info["grads"] = t.grads
    return out, info
