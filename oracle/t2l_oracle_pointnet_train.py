"""TEST INFRASTRUCTURE — float64 oracle of the PointNet++ object backbone in TRAINING mode: forward with batch-statistics
BatchNorm, analytic backward, running-statistics update (SURVEY.md §8 rows a3 + a9).

PARITY UNPINNED, like the eval-mode restatement it extends (oracle/t2l_oracle_pointnet.py: FPS start / tie rule, ball query,
PyG's bipartite self-loop edge are this build's reading of absent third-party packages). What the reference's own code fixes
and this module follows:

* the backbone is called ONCE PER CELL (models/object_encoder.py:92-95: ``[self.pointnet(pyg_batch) for pyg_batch in
  object_points]``), so under ``model.train()`` every BatchNorm1d of it normalises with the statistics of THAT cell's rows
  (the edges of all its objects for the SetAbstraction MLPs, its objects' 32 remaining points for the global MLP) and updates
  its running statistics once per cell, in cell order (momentum 0.1, unbiased variance);
* structure: models/pointcloud/pointnet2.py:18-100 — SA(0.5, r) x3 with get_mlp([cin+3, h1, h2]) (Linear, BatchNorm, ReLU
  twice), max aggregation; global MLP on cat[x, pos] + max over the points; lin1 / lin2 with ReLU; ``features2`` is consumed.

The index structure (FPS order, neighbour lists, the extra self-loop source) comes from the eval oracle's own functions, so
both oracles see identical edges. Gradients: d loss / d every ``object_encoder.pointnet.*`` parameter the path uses, given
d loss / d features2 (the classifier heads are not on the path: object_encoder.py:60-61 uses features2).
Checked against central differences of its own forward in tests/test_oracle_train.py.
"""
from __future__ import annotations

import numpy as np

from . import t2l_oracle_pointnet as OP

BN_EPS = 1e-5
MOMENTUM = 0.1
P = "object_encoder.pointnet."


def _bn_train(h, gamma, beta):
    n = h.shape[0]
    mean = h.mean(axis=0)
    var = h.var(axis=0)  # biased
    rstd = 1.0 / np.sqrt(var + BN_EPS)
    xhat = (h - mean) * rstd
    return xhat * gamma + beta, (xhat, rstd, n, mean, var)


def _bn_bwd(dy, cache, gamma):
    xhat, rstd, n, _, _ = cache
    dgamma = (dy * xhat).sum(axis=0)
    dbeta = dy.sum(axis=0)
    dxhat = dy * gamma
    dx = rstd * (dxhat - dxhat.mean(axis=0) - xhat * (dxhat * xhat).mean(axis=0))
    return dx, dgamma, dbeta


class _Mlp:
    """get_mlp([cin, h1, h2]) in training mode over ONE BatchNorm domain (the rows of one cell)."""

    def __init__(self, sd, prefix):
        self.prefix = prefix
        self.w = [sd[f"{prefix}.{i}.0.weight"].astype(np.float64) for i in range(2)]
        self.b = [sd[f"{prefix}.{i}.0.bias"].astype(np.float64) for i in range(2)]
        self.g = [sd[f"{prefix}.{i}.1.weight"].astype(np.float64) for i in range(2)]
        self.be = [sd[f"{prefix}.{i}.1.bias"].astype(np.float64) for i in range(2)]

    def forward(self, x):
        cache = []
        for i in range(2):
            h = x @ self.w[i].T + self.b[i]
            y, bn = _bn_train(h, self.g[i], self.be[i])
            a = np.maximum(y, 0.0)
            cache.append((x, bn, a))
            x = a
        return x, cache

    def backward(self, da, cache, grads, need_dx=True):
        for i in (1, 0):
            x, bn, a = cache[i]
            dy = da * (a > 0)
            dh, dg, db = _bn_bwd(dy, bn, self.g[i])
            grads[f"{self.prefix}.{i}.1.weight"] = grads.get(f"{self.prefix}.{i}.1.weight", 0) + dg
            grads[f"{self.prefix}.{i}.1.bias"] = grads.get(f"{self.prefix}.{i}.1.bias", 0) + db
            grads[f"{self.prefix}.{i}.0.weight"] = grads.get(f"{self.prefix}.{i}.0.weight", 0) + dh.T @ x
            grads[f"{self.prefix}.{i}.0.bias"] = grads.get(f"{self.prefix}.{i}.0.bias", 0) + dh.sum(axis=0)
            da = dh @ self.w[i] if (need_dx or i == 1) else None
        return da


def build_edges(pos, cell_offsets, pyg_self_loops=True):
    """Index structure of the three SetAbstraction levels (shared by forward and backward): per level
    (sel [n_obj, nd], per object the list over centres of source-row indices into the CELL's flattened source array)."""
    n_obj = pos.shape[0]
    cur_pos = pos.astype(np.float32)
    levels = []
    for radius, _ in OP.LEVELS:
        ns = cur_pos.shape[1]
        nd = (ns + 1) // 2
        sel = np.zeros((n_obj, nd), dtype=np.int64)
        new_pos = np.zeros((n_obj, nd, 3), dtype=np.float32)
        src = [[None] * nd for _ in range(n_obj)]
        for c in range(len(cell_offsets) - 1):
            lo, hi = int(cell_offsets[c]), int(cell_offsets[c + 1])
            for o in range(lo, hi):
                sel[o] = OP.fps(cur_pos[o], nd)
                new_pos[o] = cur_pos[o][sel[o]]
                for t in range(nd):
                    nb = OP.ball_query(cur_pos[o], new_pos[o, t], radius) + (o - lo) * ns  # rows of the cell's source array
                    if pyg_self_loops:
                        nb = np.concatenate([nb, [(o - lo) * nd + t]])
                    src[o][t] = nb
        levels.append((sel, src, cur_pos, new_pos))
        cur_pos = new_pos
    return levels


def forward_backward(pos, rgb, cell_offsets, sd, grad_f2=None, pyg_self_loops=True):
    """pos, rgb f32[n_obj,256,3]. Returns (features2 f64[n_obj,256], info) with info["grads"] (when grad_f2 is given),
    info["running"] = name -> value after the per-cell sequential updates, info["features0"]."""
    sd64 = {k: np.asarray(v, dtype=np.float64) for k, v in sd.items() if np.asarray(v).dtype.kind == "f"}
    n_obj = pos.shape[0]
    n_cells = len(cell_offsets) - 1
    levels = build_edges(pos, cell_offsets, pyg_self_loops)
    running = {k: v.copy() for k, v in sd64.items() if k.startswith(P) and "running_" in k}

    def update_running(prefix, caches):  # caches: the two BatchNorm caches of one cell's call
        for i in range(2):
            _, _, n, mean, var = caches[i][1]
            running[f"{prefix}.{i}.1.running_mean"] = (1 - MOMENTUM) * running[f"{prefix}.{i}.1.running_mean"] + MOMENTUM * mean
            running[f"{prefix}.{i}.1.running_var"] = (1 - MOMENTUM) * running[f"{prefix}.{i}.1.running_var"] + MOMENTUM * var * n / max(n - 1, 1)

    x_levels = [rgb.astype(np.float64)]
    fwd = []  # per level, per cell: (mlp cache, row -> (object, centre), argmax rows)
    for li, (radius, name) in enumerate(OP.LEVELS):
        sel, src, src_pos, new_pos = levels[li]
        mlp = _Mlp(sd64, f"{P}{name}.point_conv.local_nn")
        x_prev = x_levels[-1]
        nd = sel.shape[1]
        h2 = mlp.w[1].shape[0]
        new_x = np.zeros((n_obj, nd, h2))
        per_cell = []
        for c in range(n_cells):
            lo, hi = int(cell_offsets[c]), int(cell_offsets[c + 1])
            xs = x_prev[lo:hi].reshape(-1, x_prev.shape[2])
            ps = src_pos[lo:hi].reshape(-1, 3).astype(np.float64)
            rows_src, rows_ctr = [], []
            for o in range(lo, hi):
                for t in range(nd):
                    rows_src.extend(src[o][t].tolist())
                    rows_ctr.extend([(o, t)] * len(src[o][t]))
            rows_src = np.array(rows_src, dtype=np.int64)
            ctr = np.array(rows_ctr, dtype=np.int64)
            xin = np.concatenate([xs[rows_src], ps[rows_src] - new_pos[ctr[:, 0], ctr[:, 1]].astype(np.float64)], axis=1)
            a, cache = mlp.forward(xin)
            update_running(mlp.prefix, cache)
            amax = {}
            start = 0
            for o in range(lo, hi):
                for t in range(nd):
                    k = len(src[o][t])
                    blk = a[start:start + k]
                    arg = blk.argmax(axis=0)  # first maximum
                    new_x[o, t] = blk[arg, np.arange(h2)]
                    amax[(o, t)] = start + arg
                    start += k
            per_cell.append((cache, rows_src, amax, lo, hi))
        fwd.append((mlp, per_cell))
        x_levels.append(new_x)
    # global abstraction, per cell
    ga = _Mlp(sd64, P + "ga.mlp")
    pos3 = levels[-1][3].astype(np.float64)
    x3 = x_levels[-1]
    f0 = np.zeros((n_obj, ga.w[1].shape[0]))
    ga_cells = []
    for c in range(n_cells):
        lo, hi = int(cell_offsets[c]), int(cell_offsets[c + 1])
        xin = np.concatenate([x3[lo:hi], pos3[lo:hi]], axis=2).reshape(-1, x3.shape[2] + 3)
        a, cache = ga.forward(xin)
        update_running(ga.prefix, cache)
        a3 = a.reshape(hi - lo, pos3.shape[1], -1)
        arg = a3.argmax(axis=1)
        f0[lo:hi] = np.take_along_axis(a3, arg[:, None, :], axis=1)[:, 0, :]
        ga_cells.append((cache, arg, lo, hi))
    w1, b1 = sd64[P + "lin1.weight"], sd64[P + "lin1.bias"]
    w2, b2 = sd64[P + "lin2.weight"], sd64[P + "lin2.bias"]
    f1 = np.maximum(f0 @ w1.T + b1, 0.0)
    f2 = np.maximum(f1 @ w2.T + b2, 0.0)
    info = {"running": running, "features0": f0, "features1": f1}
    if grad_f2 is None:
        return f2, info

    grads = {}
    d2 = np.asarray(grad_f2, dtype=np.float64) * (f2 > 0)
    grads[P + "lin2.weight"], grads[P + "lin2.bias"] = d2.T @ f1, d2.sum(axis=0)
    d1 = (d2 @ w2) * (f1 > 0)
    grads[P + "lin1.weight"], grads[P + "lin1.bias"] = d1.T @ f0, d1.sum(axis=0)
    df0 = d1 @ w1
    dx = np.zeros_like(x3)
    for cache, arg, lo, hi in ga_cells:
        npts = pos3.shape[1]
        da = np.zeros((hi - lo, npts, f0.shape[1]))
        np.put_along_axis(da, arg[:, None, :], df0[lo:hi][:, None, :], axis=1)
        dxin = ga.backward(da.reshape(-1, f0.shape[1]), cache, grads)
        dx[lo:hi] = dxin.reshape(hi - lo, npts, -1)[:, :, : x3.shape[2]]
    for li in (2, 1, 0):
        mlp, per_cell = fwd[li]
        x_prev = x_levels[li]
        dprev = np.zeros_like(x_prev)
        h2 = mlp.w[1].shape[0]
        for cache, rows_src, amax, lo, hi in per_cell:
            n_rows = cache[1][2].shape[0]
            da = np.zeros((n_rows, h2))
            for (o, t), rows in amax.items():
                np.add.at(da, (rows, np.arange(h2)), dx[o, t])
            dxin = mlp.backward(da, cache, grads, need_dx=li > 0)
            if li > 0:
                flat = dprev[lo:hi].reshape(-1, x_prev.shape[2])
                np.add.at(flat, rows_src, dxin[:, : x_prev.shape[2]])
                dprev[lo:hi] = flat.reshape(hi - lo, x_prev.shape[1], -1)
        dx = dprev
    info["grads"] = grads
    return f2, info
