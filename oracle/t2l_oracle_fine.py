"""TEST INFRASTRUCTURE — CPU oracle for the fine stage: ``CrossMatch.forward`` in eval mode
(models/cross_matcher.py:86-135) downstream of the text branch — ObjectEncoder at fine_embed_dim + F.normalize, the
cascaded cross-attention decoder layers (``nn.TransformerDecoderLayer``, post-norm, ReLU, no masks), max over the hints,
``mlp_offsets``. numpy, float32 by default.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import this module.
Pinned by ``tests/golden/fine_{embed,pn}.npz`` = the imported reference's own CrossMatch on seeded inputs
(oracle/gen_golden_fine.py), checked in tests/test_oracle_golden.py.
"""
from __future__ import annotations

import numpy as np

from . import t2l_oracle as O

F32 = np.float32


def fine_object_encodings(cells: dict, sd: dict, class_embed: bool, color_embed: bool, pad_size: int = 16,
                          use_features=("class", "color", "position", "num")) -> np.ndarray:
    """[n_cells, pad_size, D]: ObjectEncoder.forward (object_encoder.py:66-153) then reshape + F.normalize
    (cross_matcher.py:103-104). Every cell must already hold exactly ``pad_size`` objects (eval.py:147-156)."""
    feats = O.encode_object_features(cells, sd, class_embed, color_embed, use_features)
    return O.l2_normalize(feats).reshape(-1, pad_size, feats.shape[1])


def _mha(q_in, kv_in, sd, prefix, n_heads):
    """nn.MultiheadAttention (batch_first=False semantics, evaluated per sample): q_in [P,Tq,D], kv_in [P,Tk,D]."""
    D = q_in.shape[-1]
    hd = D // n_heads
    w, b = sd[prefix + ".in_proj_weight"], sd[prefix + ".in_proj_bias"]
    q = q_in @ w[:D].T + b[:D]
    k = kv_in @ w[D:2 * D].T + b[D:2 * D]
    v = kv_in @ w[2 * D:].T + b[2 * D:]

    def heads(t):
        P, T, _ = t.shape
        return t.reshape(P, T, n_heads, hd).transpose(0, 2, 1, 3)

    q, k, v = heads(q), heads(k), heads(v)
    s = (q @ k.transpose(0, 1, 3, 2)) / F32(np.sqrt(hd))
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    p = p / p.sum(-1, keepdims=True, dtype=p.dtype)
    o = (p @ v).transpose(0, 2, 1, 3).reshape(q_in.shape)
    return o @ sd[prefix + ".out_proj.weight"].T + sd[prefix + ".out_proj.bias"]


def decoder_layer(x, mem, sd, prefix, n_heads):
    """torch.nn.TransformerDecoderLayer(d, nhead, 4d), norm_first=False, eval: x = norm1(x + SA(x));
    x = norm2(x + MHA(x, mem, mem)); x = norm3(x + linear2(relu(linear1(x))))."""
    x = O.layer_norm(x + _mha(x, x, sd, prefix + ".self_attn", n_heads), sd[prefix + ".norm1.weight"], sd[prefix + ".norm1.bias"])
    x = O.layer_norm(x + _mha(x, mem, sd, prefix + ".multihead_attn", n_heads), sd[prefix + ".norm2.weight"], sd[prefix + ".norm2.bias"])
    ff = np.maximum(x @ sd[prefix + ".linear1.weight"].T + sd[prefix + ".linear1.bias"], F32(0)) @ sd[prefix + ".linear2.weight"].T + sd[prefix + ".linear2.bias"]
    return O.layer_norm(x + ff, sd[prefix + ".norm3.weight"], sd[prefix + ".norm3.bias"])


def cross_match(obj_enc: np.ndarray, hint_enc: np.ndarray, sd: dict, n_layers: int = 2, n_heads: int = 4) -> np.ndarray:
    """obj_enc [P,n_obj,D] (unit rows), hint_enc [P,n_hints,D] -> offsets [P,2] (cross_matcher.py:109-131, the
    ``len(cross_hints) == len(cross_objects)`` branch the published configuration takes; ``n_layers == 0`` = the
    ``fine_num_decoder_layers == 0`` construction, cross_matcher.py:75-79 / :119-120: one ``cross_hints`` layer, hints attend the raw objects)."""
    d0, d1 = obj_enc.astype(F32), hint_enc.astype(F32)
    sd = {k: np.asarray(v, dtype=F32) for k, v in sd.items() if np.asarray(v).dtype.kind == "f"}
    if n_layers == 0:
        d1 = decoder_layer(d1, d0, sd, "cross_hints", n_heads)
    for i in range(n_layers):
        d0 = decoder_layer(d0, d1, sd, f"cross_objects.{i}", n_heads)
        d1 = decoder_layer(d1, d0, sd, f"cross_hints.{i}", n_heads)
    h = d1.max(axis=1)
    h = np.maximum(h @ sd["mlp_offsets.0.weight"].T + sd["mlp_offsets.0.bias"], F32(0))
    return h @ sd["mlp_offsets.2.weight"].T + sd["mlp_offsets.2.bias"]
