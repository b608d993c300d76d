"""TEST INFRASTRUCTURE — CPU oracle for the Text2Loc coarse-retrieval hot path (numpy).

This is a restatement of the reference's algorithm for the path SURVEY.md §8 scopes, written from
its observable behaviour; every function cites the reference file:line it follows. It is the
*checker*: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it. The product (``text2loc_amd``) never does, and fails loudly without its HIP library.

Pinning: checked against golden vectors produced by running the imported reference in the build
container (``oracle/gen_golden.py`` -> ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``).
NOT pinned: PointNet++ (third-party arithmetic absent) — "parity unpinned" there;
published-mode vectors start downstream of it (fixed ``features2`` inputs).
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
BN_EPS = 1e-5  # torch.nn.BatchNorm1d default
LN_EPS = 1e-5  # torch.nn.LayerNorm default (TransformerEncoderLayer)
NUM_MEAN = 1826.6844940968194  # models/object_encoder.py:43
NUM_STD = 2516.8905096993817  # models/object_encoder.py:44


# ----------------------------------------------------------------------------------------------
# a1 — per-object reductions (datapreparation/kitti360pose/imports.py:28-41)
# ----------------------------------------------------------------------------------------------
def object_reductions(xyz: np.ndarray, rgb: np.ndarray, colors: np.ndarray):
    """mean rgb (imports.py:28-31), nearest colour centre index (imports.py:33-38),
    mean xyz (imports.py:40-41), point count (object_encoder.py:141 ``len(obj.xyz)``)."""
    color_rgb = np.mean(rgb, axis=0)
    dists = np.linalg.norm(np.mean(rgb, axis=0) - colors, axis=1)
    return color_rgb, int(np.argmin(dists)), np.mean(xyz, axis=0), len(xyz)


# ----------------------------------------------------------------------------------------------
# building blocks
# ----------------------------------------------------------------------------------------------
def l2_normalize(x: np.ndarray, eps: float = 1e-12) -> np.ndarray:
    """torch.nn.functional.normalize(dim=-1): x / max(||x||_2, eps)."""
    n = np.sqrt(np.sum(x * x, axis=-1, keepdims=True, dtype=x.dtype))
    return x / np.maximum(n, F32(eps) if x.dtype == F32 else eps)


def linear(x, w, b):
    return x @ w.T + b


def batchnorm_eval(x, sd, prefix):
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    return (x - rm) / np.sqrt(rv + F32(BN_EPS)) * sd[prefix + ".weight"] + sd[prefix + ".bias"]


def mlp(x, sd, prefix, n_layers, last_relu=True):
    """get_mlp (models/language_encoder.py:16-41): [Linear, BatchNorm1d, ReLU] per layer with a
    trailing ReLU; get_mlp2 (:43-74) drops the last ReLU (last_relu=False). Eval-mode BN."""
    for i in range(n_layers):
        x = linear(x, sd[f"{prefix}.{i}.0.weight"], sd[f"{prefix}.{i}.0.bias"])
        x = batchnorm_eval(x, sd, f"{prefix}.{i}.1")
        if last_relu or i < n_layers - 1:
            x = np.maximum(x, F32(0))
    return x


def layer_norm(x, w, b):
    mu = x.mean(axis=-1, keepdims=True, dtype=x.dtype)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True, dtype=x.dtype)
    return (x - mu) / np.sqrt(var + F32(LN_EPS)) * w + b


def encoder_layer(x, sd, prefix, n_heads):
    """torch.nn.TransformerEncoderLayer, post-norm, ReLU, eval (dropout off), NO mask — exactly how
    models/cell_retrieval.py:35,102-103 and models/language_encoder.py:95,100,130-147 use it.
    x: [S, B, D] (batch_first=False). in_proj rows [0:D]=Wq, [D:2D]=Wk, [2D:3D]=Wv."""
    S, B, D = x.shape
    hd = D // n_heads
    wi, bi = sd[prefix + ".self_attn.in_proj_weight"], sd[prefix + ".self_attn.in_proj_bias"]
    qkv = linear(x, wi, bi)  # [S,B,3D]
    q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]

    def heads(t):  # [S,B,D] -> [B,H,S,hd]
        return t.reshape(S, B, n_heads, hd).transpose(1, 2, 0, 3)

    q, k, v = heads(q), heads(k), heads(v)
    scores = (q @ k.transpose(0, 1, 3, 2)) / F32(np.sqrt(hd))  # [B,H,S,S]
    scores = scores - scores.max(axis=-1, keepdims=True)
    p = np.exp(scores)
    p = p / p.sum(axis=-1, keepdims=True, dtype=p.dtype)
    o = (p @ v).transpose(2, 0, 1, 3).reshape(S, B, D)
    attn = linear(o, sd[prefix + ".self_attn.out_proj.weight"], sd[prefix + ".self_attn.out_proj.bias"])
    x = layer_norm(x + attn, sd[prefix + ".norm1.weight"], sd[prefix + ".norm1.bias"])
    ff = linear(np.maximum(linear(x, sd[prefix + ".linear1.weight"], sd[prefix + ".linear1.bias"]), F32(0)),
                sd[prefix + ".linear2.weight"], sd[prefix + ".linear2.bias"])
    return layer_norm(x + ff, sd[prefix + ".norm2.weight"], sd[prefix + ".norm2.bias"])


# ----------------------------------------------------------------------------------------------
# a2 — ObjectEncoder.forward (models/object_encoder.py:66-153), eval mode
# ----------------------------------------------------------------------------------------------
def encode_object_features(cells: dict, sd: dict, class_embed: bool, color_embed: bool,
                           use_features=("class", "color", "position", "num")) -> np.ndarray:
    """Per-object merged feature f32[total_objects, D] (before cell_retrieval.py:92's normalize).
    Feature order is fixed by code order class -> color -> position -> num (object_encoder.py:102-145)."""
    p = "object_encoder."
    emb = []
    if "class" in use_features:
        if class_embed:  # object_encoder.py:103-110
            e = sd[p + "class_embedding.weight"][cells["class_idx"]]
        else:  # object_encoder.py:86-99,112: PointNet features2 -> mlp_pointnet
            e = mlp(cells["pn_feat"].astype(F32), sd, p + "mlp_pointnet", 1)
        emb.append(l2_normalize(e))
    if "color" in use_features:
        if color_embed:  # object_encoder.py:116-120
            e = sd[p + "color_embedding.weight"][cells["color_idx"]]
        else:  # object_encoder.py:121-128
            e = mlp(cells["rgb"].astype(F32), sd, p + "color_encoder", 2)
        emb.append(l2_normalize(e))
    if "position" in use_features:  # object_encoder.py:130-136
        emb.append(l2_normalize(mlp(cells["center"].astype(F32), sd, p + "pos_encoder", 2)))
    if "num" in use_features:  # object_encoder.py:138-145
        x = ((cells["n_pts"].astype(F32)[:, None] - F32(NUM_MEAN)) / F32(NUM_STD)).astype(F32)
        emb.append(l2_normalize(mlp(x, sd, p + "num_encoder", 2)))
    if len(emb) > 1:  # object_encoder.py:148-149
        return mlp(np.concatenate(emb, axis=-1), sd, p + "mlp_merge", 1)
    return emb[0]


# ----------------------------------------------------------------------------------------------
# a4 — CellRetrievalNetwork.encode_objects (models/cell_retrieval.py:65-110), eval mode
# ----------------------------------------------------------------------------------------------
def encode_cells(cells: dict, sd: dict, class_embed: bool, color_embed: bool, object_size: int = 28,
                 n_heads: int = 4, n_layers: int = 2, use_features=("class", "color", "position", "num"),
                 return_stages: bool = False):
    feats = encode_object_features(cells, sd, class_embed, color_embed, use_features)
    emb = l2_normalize(feats)  # cell_retrieval.py:92
    B, D = len(cells["counts"]), emb.shape[1]
    x = np.zeros((B, object_size, D), dtype=F32)  # cell_retrieval.py:85
    for i in range(B):  # cell_retrieval.py:94-98 (first min(n,28) objects; rest silently dropped)
        lo = int(cells["offsets"][i])
        n = min(int(cells["counts"][i]), object_size)
        x[i, :n] = emb[lo:lo + n]
    x = np.ascontiguousarray(x.transpose(1, 0, 2))  # [S,B,D] cell_retrieval.py:101
    stages = [x.copy()] if return_stages else None
    for layer in range(n_layers):  # cell_retrieval.py:102-103 — no padding mask
        x = encoder_layer(x, sd, f"obj_inter_module.{layer}", n_heads)
        if return_stages:
            stages.append(x.copy())
    pooled = x.max(axis=0)  # cell_retrieval.py:107 — over ALL slots including pads
    out = l2_normalize(pooled)  # cell_retrieval.py:108
    if return_stages:
        return out, feats, stages
    return out


# ----------------------------------------------------------------------------------------------
# a5 (head only) — LanguageEncoder.forward after T5 (models/language_encoder.py:127-148) + normalize
# ----------------------------------------------------------------------------------------------
def text_head(hidden: np.ndarray, sd: dict, batch_size: int, n_heads_intra=4, n_heads_inter=4) -> np.ndarray:
    """hidden: T5 last_hidden_state f32[6B, L, 1024]. No padding mask anywhere (language_encoder.py:130-135)."""
    p = "language_encoder."
    x = np.ascontiguousarray(hidden.transpose(1, 0, 2))  # :128
    x = encoder_layer(x, sd, p + "intra_module.0", n_heads_intra)  # :130-131
    x = x.transpose(1, 0, 2).max(axis=1)  # :132-133 (max over tokens incl. pads)
    x = mlp(x, sd, p + "inter_mlp", 1, last_relu=False)  # :135
    x = x.reshape(batch_size, -1, x.shape[-1]).transpose(1, 0, 2)  # :136,141
    x = x + encoder_layer(np.ascontiguousarray(x), sd, p + "inter_module.0", n_heads_inter)  # :142-143
    x = x.max(axis=0)  # :145
    return l2_normalize(x)  # cell_retrieval.py:61


# ----------------------------------------------------------------------------------------------
# a6 — retrieval loop of eval_epoch (training/coarse.py:81-86,119-146)
# ----------------------------------------------------------------------------------------------
def retrieve_topk(cell_encodings: np.ndarray, text_encodings: np.ndarray, k: int):
    """Per query: float64 ``C @ t`` and a full descending argsort, first k kept (coarse.py:121-125).
    The reference's np.argsort is not stable; THIS oracle defines exact-score ties as lower row
    index first (SURVEY.md §7 'hard parts') and the fixtures are tie-free.
    Returns (idx i64[Q,k], scores f64[Q,k])."""
    C = np.asarray(cell_encodings, dtype=np.float64)  # coarse.py:81 (np.zeros -> float64)
    T = np.asarray(text_encodings, dtype=np.float64)
    Q = T.shape[0]
    k = min(k, C.shape[0])
    idx = np.zeros((Q, k), dtype=np.int64)
    sc = np.zeros((Q, k), dtype=np.float64)
    for q in range(Q):
        scores = C[:] @ T[q]
        order = np.argsort(-1.0 * scores, kind="stable")[0:k]
        idx[q], sc[q] = order, scores[order]
    return idx, sc


def eval_accuracies(top_idx, db_cell_ids, query_cell_ids, query_poses_xy, cell_centers_xy, cell_size, top_k):
    """hit@k (coarse.py:127-133) and close@k (coarse.py:135-146), averaged (coarse.py:148-150)."""
    acc = {k: [] for k in top_k}
    close = {k: [] for k in top_k}
    for q in range(len(top_idx)):
        ids = db_cell_ids[top_idx[q]]
        for k in top_k:
            acc[k].append(query_cell_ids[q] in ids[0:k])
        d = np.linalg.norm(query_poses_xy[q] - cell_centers_xy[top_idx[q]], axis=1)
        for k in top_k:
            close[k].append(np.any(d[0:k] <= cell_size / 2))
    return {k: np.mean(v) for k, v in acc.items()}, {k: np.mean(v) for k, v in close.items()}


def coarse_pose_accuracies(top_idx, pose_w_xy, pose_scene, cell_bbox_xy, cell_scene, cell_size, top_k, threshs):
    """run_coarse post-processing (evaluation/pipeline.py:70-83) + calc_sample_accuracies
    (evaluation/utils.py:31-54): predicted position = cell centre (0.5,0.5), cross-scene = inf."""
    out = {k: {t: [] for t in threshs} for k in top_k}
    for q in range(len(top_idx)):
        pred = cell_bbox_xy[top_idx[q]] + 0.5 * cell_size
        d = np.linalg.norm(pose_w_xy[q] - pred, axis=1)
        d[cell_scene[top_idx[q]] != pose_scene[q]] = np.inf
        for k in top_k:
            for t in threshs:
                out[k][t].append(np.min(d[0:k]) <= t)
    return {k: {t: np.mean(v) for t, v in d.items()} for k, d in out.items()}


# ----------------------------------------------------------------------------------------------
# a8 — ContrastiveLoss.forward (training/losses.py:269-283) and its analytic gradient
# ----------------------------------------------------------------------------------------------
def contrastive_loss(im: np.ndarray, s: np.ndarray, temperature: float, dtype=np.float32):
    """Returns (loss, d loss/d im, d loss/d s). No max-subtraction, exactly as the reference."""
    im = im.astype(dtype)
    s = s.astype(dtype)
    B = im.shape[0]
    ni = np.sqrt((im * im).sum(axis=1, keepdims=True))
    ns = np.sqrt((s * s).sum(axis=1, keepdims=True))
    a, p = im / ni, s / ns  # losses.py:271-272
    sim = a @ p.T  # :274
    t = dtype(temperature)
    e = np.exp(sim / t)  # :278
    num = np.exp(np.diag(sim) / t)  # :275,277
    col, row = e.sum(axis=0), e.sum(axis=1)
    losses = -np.log(num / col) - np.log(num / row)  # :280
    loss = losses.mean()  # :281
    # d loss / d sim[i,j] = (1/B) * ( e_ij/col_j + e_ij/row_i - 2*delta_ij ) / t
    g = (e / col[None, :] + e / row[:, None] - 2.0 * np.eye(B, dtype=dtype)) / (t * dtype(B))
    ga, gp = g @ p, g.T @ a
    # back through x / ||x||
    gim = (ga - a * (ga * a).sum(axis=1, keepdims=True)) / ni
    gs = (gp - p * (gp * p).sum(axis=1, keepdims=True)) / ns
    return loss, gim, gs
