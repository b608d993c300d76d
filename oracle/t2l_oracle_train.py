"""TEST INFRASTRUCTURE — CPU oracle for the TRAINING step of the object branch.

Restates, in numpy (float64 by default), what the reference's ``train_epoch`` body does to the object branch
(training/coarse.py:31-58): ``model.train()`` forward of ``CellRetrievalNetwork.encode_objects``
(models/cell_retrieval.py:65-110; ObjectEncoder models/object_encoder.py:66-153 with BatchNorm1d in batch-statistics
mode; ``nn.TransformerEncoderLayer`` post-norm with its four dropout sites), the gradients autograd produces for it,
the BatchNorm running-statistics update, and ``torch.optim.Adam``'s update (training/coarse.py:258).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import this module.

Pinning: ``tests/test_oracle_train.py`` checks forward, loss gradients, running statistics and the Adam step
against ``tests/golden/train_step_{embed,pn}.npz`` (the imported reference run with dropout p=0, see
oracle/gen_golden_train.py) and checks the analytic backward against central differences of the forward.
Dropout masks are this build's own counter-based masks (``dropout_keep``): torch's RNG stream is not reproducible,
so with p>0 the oracle pins the build's arithmetic, not the reference's random draw.
"""
from __future__ import annotations

import numpy as np

BN_EPS = 1e-5
LN_EPS = 1e-5
BN_MOMENTUM = 0.1
NORM_EPS = 1e-12
NUM_MEAN = 1826.6844940968194  # models/object_encoder.py:43
NUM_STD = 2516.8905096993817  # models/object_encoder.py:44
FEATURES = ("class", "color", "position", "num")


# ----------------------------------------------------------------------------------------------------------------
# dropout masks: counter-based (seed, site, flat element index) -> keep bit. The HIP kernels use the same function.
# ----------------------------------------------------------------------------------------------------------------
def dropout_keep(seed: int, site: int, n: int, p: float) -> np.ndarray:
    """bool[n]; element i is kept iff top 24 bits of lowbias32(i*0x9E3779B1 + (seed ^ site*0x85EBCA77)) >= p*2^24."""
    if p <= 0.0:
        return np.ones(n, dtype=bool)
    M = np.uint64(0xFFFFFFFF)
    x = (np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B1) + np.uint64((seed ^ (site * 0x85EBCA77)) & 0xFFFFFFFF)) & M
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & M
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & M
    x ^= x >> np.uint64(16)
    thr = np.uint64(int(p * (1 << 24)))
    return (x >> np.uint64(8)) >= thr


def _drop(x, seed, site, p):
    if p <= 0.0:
        return x, None
    keep = dropout_keep(seed, site, x.size, p).reshape(x.shape)
    scale = keep / (1.0 - p)
    return x * scale, scale


# ----------------------------------------------------------------------------------------------------------------
# building blocks with backward
# ----------------------------------------------------------------------------------------------------------------
def _normalize_fwd(x):
    n = np.maximum(np.sqrt((x * x).sum(-1, keepdims=True)), NORM_EPS)
    return x / n, n


def _normalize_bwd(dy, y, n):
    # F.normalize: y = x / max(||x||, eps). For ||x|| > eps: dx = (dy - y (y.dy)) / ||x||; clamped rows: dx = dy / eps
    clamped = n <= NORM_EPS
    dx = (dy - y * (y * dy).sum(-1, keepdims=True)) / n
    return np.where(clamped, dy / NORM_EPS, dx)


def _bn_fwd(x, g, b):
    mean = x.mean(0)
    var = x.var(0)  # biased, used for the normalisation (torch BatchNorm1d training mode)
    rstd = 1.0 / np.sqrt(var + BN_EPS)
    xhat = (x - mean) * rstd
    return xhat * g + b, (xhat, rstd, mean, var)


def _bn_bwd(dy, g, cache):
    xhat, rstd, _, _ = cache
    n = dy.shape[0]
    dg = (dy * xhat).sum(0)
    db = dy.sum(0)
    dx = (g * rstd / n) * (n * dy - db - xhat * dg)
    return dx, dg, db


def _ln_fwd(z, g, b):
    mu = z.mean(-1, keepdims=True)
    rstd = 1.0 / np.sqrt(((z - mu) ** 2).mean(-1, keepdims=True) + LN_EPS)
    xhat = (z - mu) * rstd
    return xhat * g + b, (xhat, rstd)


def _ln_bwd(dy, g, cache):
    xhat, rstd = cache
    dg = (dy * xhat).reshape(-1, xhat.shape[-1]).sum(0)
    db = dy.reshape(-1, xhat.shape[-1]).sum(0)
    dxh = dy * g
    dz = rstd * (dxh - dxh.mean(-1, keepdims=True) - xhat * (dxh * xhat).mean(-1, keepdims=True))
    return dz, dg, db


class _Tape:
    """Weights in the working dtype + gradient accumulators + BatchNorm batch statistics seen in forward."""

    def __init__(self, sd, dtype):
        self.w = {k: np.asarray(v).astype(dtype) for k, v in sd.items() if np.asarray(v).dtype.kind == "f"}
        self.grads = {}
        self.bn_stats = {}  # prefix -> (mean, biased var, n)

    def add(self, name, g):
        self.grads[name] = self.grads.get(name, 0) + g


def _mlp_fwd(t: _Tape, x, prefix, n_layers):
    """get_mlp (models/language_encoder.py:16-41): [Linear, BatchNorm1d(batch stats), ReLU] per layer."""
    caches = []
    for i in range(n_layers):
        w, b = t.w[f"{prefix}.{i}.0.weight"], t.w[f"{prefix}.{i}.0.bias"]
        y = x @ w.T + b
        z, bc = _bn_fwd(y, t.w[f"{prefix}.{i}.1.weight"], t.w[f"{prefix}.{i}.1.bias"])
        t.bn_stats[f"{prefix}.{i}.1"] = (bc[2], bc[3], y.shape[0])
        out = np.maximum(z, 0)
        caches.append((x, bc, z > 0))
        x = out
    return x, caches


def _mlp_bwd(t: _Tape, d, prefix, caches, need_dx=False):
    for i in reversed(range(len(caches))):
        x, bc, pos = caches[i]
        d = d * pos
        d, dg, db = _bn_bwd(d, t.w[f"{prefix}.{i}.1.weight"], bc)
        t.add(f"{prefix}.{i}.1.weight", dg)
        t.add(f"{prefix}.{i}.1.bias", db)
        t.add(f"{prefix}.{i}.0.weight", d.T @ x)
        t.add(f"{prefix}.{i}.0.bias", d.sum(0))
        if i > 0 or need_dx:
            d = d @ t.w[f"{prefix}.{i}.0.weight"]
    return d


def _layer_fwd(t: _Tape, x, prefix, H, p, seed, layer):
    """nn.TransformerEncoderLayer (post-norm, ReLU) in train mode, x [B,S,D] (token order does not matter)."""
    B, S, D = x.shape
    hd = D // H
    w = t.w
    qkv = x @ w[prefix + ".self_attn.in_proj_weight"].T + w[prefix + ".self_attn.in_proj_bias"]

    def heads(a):
        return a.reshape(B, S, H, hd).transpose(0, 2, 1, 3)  # [B,H,S,hd]

    q, k, v = heads(qkv[..., :D]), heads(qkv[..., D:2 * D]), heads(qkv[..., 2 * D:])
    s = (q @ k.transpose(0, 1, 3, 2)) / np.sqrt(hd)
    s = s - s.max(-1, keepdims=True)
    pr = np.exp(s)
    pr = pr / pr.sum(-1, keepdims=True)
    prd, m0 = _drop(pr, seed, layer * 4 + 0, p)  # attention-probability dropout (MultiheadAttention.dropout)
    o = (prd @ v).transpose(0, 2, 1, 3).reshape(B, S, D)
    a = o @ w[prefix + ".self_attn.out_proj.weight"].T + w[prefix + ".self_attn.out_proj.bias"]
    ad, m1 = _drop(a, seed, layer * 4 + 1, p)  # dropout1
    x1, ln1 = _ln_fwd(x + ad, w[prefix + ".norm1.weight"], w[prefix + ".norm1.bias"])
    h = np.maximum(x1 @ w[prefix + ".linear1.weight"].T + w[prefix + ".linear1.bias"], 0)
    hdp, m2 = _drop(h, seed, layer * 4 + 2, p)  # dropout
    f = hdp @ w[prefix + ".linear2.weight"].T + w[prefix + ".linear2.bias"]
    fd, m3 = _drop(f, seed, layer * 4 + 3, p)  # dropout2
    x2, ln2 = _ln_fwd(x1 + fd, w[prefix + ".norm2.weight"], w[prefix + ".norm2.bias"])
    return x2, (x, q, k, v, pr, prd, m0, o, m1, ln1, x1, h, hdp, m2, m3, ln2)


def _layer_bwd(t: _Tape, d, prefix, H, cache):
    x, q, k, v, pr, prd, m0, o, m1, ln1, x1, h, hdp, m2, m3, ln2 = cache
    B, S, D = x.shape
    hd = D // H
    w = t.w
    dz2, dg, db = _ln_bwd(d, w[prefix + ".norm2.weight"], ln2)
    t.add(prefix + ".norm2.weight", dg)
    t.add(prefix + ".norm2.bias", db)
    dfd = dz2 if m3 is None else dz2 * m3
    t.add(prefix + ".linear2.weight", dfd.reshape(-1, D).T @ hdp.reshape(-1, hdp.shape[-1]))
    t.add(prefix + ".linear2.bias", dfd.reshape(-1, D).sum(0))
    dh = dfd @ w[prefix + ".linear2.weight"]
    if m2 is not None:
        dh = dh * m2
    dh = dh * (h > 0)
    t.add(prefix + ".linear1.weight", dh.reshape(-1, dh.shape[-1]).T @ x1.reshape(-1, D))
    t.add(prefix + ".linear1.bias", dh.reshape(-1, dh.shape[-1]).sum(0))
    dx1 = dz2 + dh @ w[prefix + ".linear1.weight"]
    dz1, dg, db = _ln_bwd(dx1, w[prefix + ".norm1.weight"], ln1)
    t.add(prefix + ".norm1.weight", dg)
    t.add(prefix + ".norm1.bias", db)
    da = dz1 if m1 is None else dz1 * m1
    t.add(prefix + ".self_attn.out_proj.weight", da.reshape(-1, D).T @ o.reshape(-1, D))
    t.add(prefix + ".self_attn.out_proj.bias", da.reshape(-1, D).sum(0))
    do = (da @ w[prefix + ".self_attn.out_proj.weight"]).reshape(B, S, H, hd).transpose(0, 2, 1, 3)
    dv = prd.transpose(0, 1, 3, 2) @ do
    dprd = do @ v.transpose(0, 1, 3, 2)
    dpr = dprd if m0 is None else dprd * m0
    ds = pr * (dpr - (dpr * pr).sum(-1, keepdims=True)) / np.sqrt(hd)
    dq = ds @ k
    dk = ds.transpose(0, 1, 3, 2) @ q

    def unheads(a):
        return a.transpose(0, 2, 1, 3).reshape(B, S, D)

    dqkv = np.concatenate([unheads(dq), unheads(dk), unheads(dv)], axis=-1)
    t.add(prefix + ".self_attn.in_proj_weight", dqkv.reshape(-1, 3 * D).T @ x.reshape(-1, D))
    t.add(prefix + ".self_attn.in_proj_bias", dqkv.reshape(-1, 3 * D).sum(0))
    return dz1 + dqkv @ w[prefix + ".self_attn.in_proj_weight"]


# ----------------------------------------------------------------------------------------------------------------
# forward + backward of encode_objects in training mode
# ----------------------------------------------------------------------------------------------------------------
def encode_cells_train(cells: dict, sd: dict, class_embed: bool, color_embed: bool, grad_out=None, p_drop: float = 0.0,
                       seed: int = 0, object_size: int = 28, n_heads: int = 4, n_layers: int = 2,
                       use_features=FEATURES, dtype=np.float64):
    """Returns (out [B,D], info). ``info['bn_stats']``: BatchNorm prefix -> (batch mean, biased var, n rows).
    With ``grad_out`` [B,D] (dLoss/d out) also ``info['grads']`` (state_dict names) and, in the PointNet mode,
    ``info['grad_pn_feat']``."""
    t = _Tape(sd, dtype)
    p = "object_encoder."
    counts, offsets = np.asarray(cells["counts"]), np.asarray(cells["offsets"])
    B = len(counts)
    branches = []  # (kind, payload) per feature, code order class -> color -> position -> num (object_encoder.py:102-145)
    emb = []
    for f in FEATURES:
        if f not in use_features:
            continue
        if f == "class" and class_embed:
            e = t.w[p + "class_embedding.weight"][cells["class_idx"]]
            branches.append(("emb", p + "class_embedding.weight", np.asarray(cells["class_idx"])))
        elif f == "class":
            e, c = _mlp_fwd(t, np.asarray(cells["pn_feat"]).astype(dtype), p + "mlp_pointnet", 1)
            branches.append(("mlp", p + "mlp_pointnet", c))
        elif f == "color" and color_embed:
            e = t.w[p + "color_embedding.weight"][cells["color_idx"]]
            branches.append(("emb", p + "color_embedding.weight", np.asarray(cells["color_idx"])))
        elif f == "color":
            e, c = _mlp_fwd(t, np.asarray(cells["rgb"]).astype(dtype), p + "color_encoder", 2)
            branches.append(("mlp", p + "color_encoder", c))
        elif f == "position":
            e, c = _mlp_fwd(t, np.asarray(cells["center"]).astype(dtype), p + "pos_encoder", 2)
            branches.append(("mlp", p + "pos_encoder", c))
        else:
            # the reference standardises in float32 on the host tensor (object_encoder.py:141-144)
            xin = ((np.asarray(cells["n_pts"], dtype=np.float32)[:, None] - np.float32(NUM_MEAN)) / np.float32(NUM_STD))
            e, c = _mlp_fwd(t, xin.astype(dtype), p + "num_encoder", 2)
            branches.append(("mlp", p + "num_encoder", c))
        y, n = _normalize_fwd(e)
        emb.append((y, n))
    cat = np.concatenate([y for y, _ in emb], axis=-1)
    if len(emb) > 1:
        feats, merge_c = _mlp_fwd(t, cat, p + "mlp_merge", 1)
    else:
        feats, merge_c = cat, None
    fy, fn = _normalize_fwd(feats)  # cell_retrieval.py:92
    D = fy.shape[1]
    x = np.zeros((B, object_size, D), dtype=dtype)
    for i in range(B):  # cell_retrieval.py:94-98
        n = min(int(counts[i]), object_size)
        x[i, :n] = fy[int(offsets[i]):int(offsets[i]) + n]
    layer_c = []
    for l in range(n_layers):
        x, c = _layer_fwd(t, x, f"obj_inter_module.{l}", n_heads, p_drop, seed, l)
        layer_c.append(c)
    arg = x.argmax(axis=1)  # [B,D] first maximal slot (cell_retrieval.py:107)
    pooled = np.take_along_axis(x, arg[:, None, :], axis=1)[:, 0, :]
    out, on = _normalize_fwd(pooled)  # cell_retrieval.py:108
    info = {"bn_stats": t.bn_stats, "features": feats}
    if grad_out is None:
        return out, info

    d = _normalize_bwd(np.asarray(grad_out).astype(dtype), out, on)
    dx = np.zeros_like(x)
    np.put_along_axis(dx, arg[:, None, :], d[:, None, :], axis=1)
    for l in reversed(range(n_layers)):
        dx = _layer_bwd(t, dx, f"obj_inter_module.{l}", n_heads, layer_c[l])
    dfy = np.zeros_like(fy)
    for i in range(B):
        n = min(int(counts[i]), object_size)
        dfy[int(offsets[i]):int(offsets[i]) + n] = dx[i, :n]
    dfeats = _normalize_bwd(dfy, fy, fn)
    dcat = _mlp_bwd(t, dfeats, p + "mlp_merge", merge_c, need_dx=True) if merge_c is not None else dfeats
    for j, ((kind, name, payload), (y, n)) in enumerate(zip(branches, emb)):
        de = _normalize_bwd(dcat[:, j * D:(j + 1) * D], y, n)
        if kind == "emb":
            g = np.zeros_like(t.w[name])
            np.add.at(g, payload, de)
            g[0] = 0  # padding_idx=0 (object_encoder.py:33,37): the row never receives gradient
            t.add(name, g)
        else:
            dxin = _mlp_bwd(t, de, name, payload, need_dx=name.endswith("mlp_pointnet"))
            if name.endswith("mlp_pointnet"):
                info["grad_pn_feat"] = dxin
    info["grads"] = t.grads
    return out, info


def bn_running_update(sd: dict, bn_stats: dict, momentum: float = BN_MOMENTUM) -> dict:
    """BatchNorm1d.train() side effect: running_mean/var <- (1-m) old + m (batch mean, UNBIASED batch var);
    num_batches_tracked += 1."""
    new = {}
    for prefix, (mean, var, n) in bn_stats.items():
        new[prefix + ".running_mean"] = (1 - momentum) * np.asarray(sd[prefix + ".running_mean"], dtype=np.float64) + momentum * mean
        new[prefix + ".running_var"] = (1 - momentum) * np.asarray(sd[prefix + ".running_var"], dtype=np.float64) + momentum * var * n / max(n - 1, 1)
        new[prefix + ".num_batches_tracked"] = int(np.asarray(sd.get(prefix + ".num_batches_tracked", 0))) + 1
    return new


def adam_step(param, grad, m, v, step: int, lr: float, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam defaults (no weight decay, no amsgrad): returns (param, m, v) after update number ``step``."""
    m = beta1 * m + (1 - beta1) * grad
    v = beta2 * v + (1 - beta2) * grad * grad
    denom = np.sqrt(v) / np.sqrt(1 - beta2 ** step) + eps
    return param - (lr / (1 - beta1 ** step)) * m / denom, m, v
