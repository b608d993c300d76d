"""TEST INFRASTRUCTURE — CPU oracle for the PointNet++ object backbone, eval mode.

PARITY UNPINNED. The reference's arithmetic for this stage lives in third-party packages that are absent from
/root/reference and not installable here (torch_geometric==1.7.2, torch-cluster==1.6.0, torch-scatter==2.0.9;
requirements.txt:15-18); the reference has no tests or golden vectors for it and is stochastic there (FixedPoints
sampling, random FPS start). This module therefore restates the STRUCTURE the reference's own code fixes
(models/pointcloud/pointnet2.py:18-100) with the published semantics of those packages as this build understands them,
made deterministic — and the HIP kernels are tested against THIS restatement only:

* SetAbstractionLayer(ratio 0.5, radius r, get_mlp) (pointnet2.py:18-38), per object of a cell's PyG batch:
  - farthest point sampling of ceil(0.5 n) points — torch_cluster.fps; START = point 0 (the reference: random start);
    ties -> lowest index; squared distances in float32 as ((dx*dx + dy*dy) + dz*dz);
  - ball query — torch_cluster.radius(x, y, r, max_num_neighbors=32): for every sampled centre the FIRST 32 source
    points in index order with d^2 < r^2 (float32 r*r);
  - PointConv — message = local_nn(cat[x_j, pos_j - pos_i]) (get_mlp: Linear+BatchNorm(eval)+ReLU twice), max over the
    messages of a centre;
  - PyG 1.7 PointConv(add_self_loops=True) on a bipartite (pos_src, pos_dst) input removes edges with equal source and
    target index and then adds the edges (k -> k) for k < min(N_src, N_dst) = N_dst, indices counted over the WHOLE
    PyG batch (= all objects of the cell): centre k additionally receives the message of source node k, a point that in
    general belongs to an EARLIER object of the cell. ``pyg_self_loops=True`` (default) reproduces that; False drops it.
* GlobalAbstractionLayer (pointnet2.py:41-50): get_mlp([256+3,512,1024]) on cat[x, pos] of the 32 remaining points, max.
* lin1/lin2 with ReLU (pointnet2.py:86-89) -> features2 [n_objects, 256] (object_encoder.py:60-61 uses features2).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import this module.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
BN_EPS = 1e-5
LEVELS = ((0.2, "sa1"), (0.3, "sa2"), (0.4, "sa3"))
MAX_NEIGHBORS = 32


def _d2(p, q):
    d = (p - q).astype(F32)
    return ((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]).astype(F32) + d[..., 2] * d[..., 2]).astype(F32)


def fps(pos: np.ndarray, n_out: int) -> np.ndarray:
    """Selection ORDER of farthest point sampling from point 0 (lowest index wins ties)."""
    sel = np.zeros(n_out, dtype=np.int64)
    mind = _d2(pos, pos[0])
    for t in range(1, n_out):
        sel[t] = int(np.argmax(mind))  # numpy argmax returns the first maximum
        mind = np.minimum(mind, _d2(pos, pos[sel[t]]))
    return sel


def ball_query(pos_src: np.ndarray, centre: np.ndarray, radius: float) -> np.ndarray:
    r2 = F32(F32(radius) * F32(radius))
    idx = np.nonzero(_d2(pos_src, centre) < r2)[0]
    return idx[:MAX_NEIGHBORS]


def _mlp_eval(x, sd, prefix, n_layers=2):
    for i in range(n_layers):
        w, b = sd[f"{prefix}.{i}.0.weight"], sd[f"{prefix}.{i}.0.bias"]
        x = x @ w.T + b
        g, be = sd[f"{prefix}.{i}.1.weight"], sd[f"{prefix}.{i}.1.bias"]
        rm, rv = sd[f"{prefix}.{i}.1.running_mean"], sd[f"{prefix}.{i}.1.running_var"]
        x = np.maximum((x - rm) / np.sqrt(rv + F32(BN_EPS)) * g + be, F32(0))
    return x


def pointnet_features(pos: np.ndarray, rgb: np.ndarray, cell_offsets: np.ndarray, sd: dict, pyg_self_loops: bool = True,
                      return_levels: bool = False):
    """pos, rgb: f32[n_objects, 256, 3]; cell_offsets i32[n_cells+1] (objects per PyG batch = per cell).
    Returns features2 f32[n_objects, 256] (and, optionally, per level (pos, x, selection) lists)."""
    p = "object_encoder.pointnet."
    sd = {k: np.asarray(v, dtype=F32) for k, v in sd.items() if np.asarray(v).dtype.kind == "f"}
    n_obj = pos.shape[0]
    cur_pos, cur_x = pos.astype(F32), rgb.astype(F32)
    levels = []
    for radius, name in LEVELS:
        ns = cur_pos.shape[1]
        nd = (ns + 1) // 2
        h2 = sd[f"{p}{name}.point_conv.local_nn.1.0.weight"].shape[0]
        new_pos = np.zeros((n_obj, nd, 3), dtype=F32)
        new_x = np.zeros((n_obj, nd, h2), dtype=F32)
        sels = np.zeros((n_obj, nd), dtype=np.int64)
        for c in range(len(cell_offsets) - 1):
            lo, hi = int(cell_offsets[c]), int(cell_offsets[c + 1])
            src_pos_flat = cur_pos[lo:hi].reshape(-1, 3)  # the cell's PyG batch, object-major
            src_x_flat = cur_x[lo:hi].reshape(-1, cur_x.shape[2])
            for o in range(lo, hi):
                sel = fps(cur_pos[o], nd)
                sels[o] = sel
                new_pos[o] = cur_pos[o][sel]
                for t in range(nd):
                    nb = ball_query(cur_pos[o], new_pos[o, t], radius)
                    msg_in = np.concatenate([cur_x[o][nb], cur_pos[o][nb] - new_pos[o, t]], axis=1)
                    if pyg_self_loops:
                        k = (o - lo) * nd + t  # index of this centre in the cell's batch == index of the extra source node
                        extra = np.concatenate([src_x_flat[k], src_pos_flat[k] - new_pos[o, t]])[None, :]
                        msg_in = np.concatenate([msg_in, extra], axis=0)
                    new_x[o, t] = _mlp_eval(msg_in.astype(F32), sd, f"{p}{name}.point_conv.local_nn").max(axis=0)
        levels.append((new_pos, new_x, sels))
        cur_pos, cur_x = new_pos, new_x
    g = _mlp_eval(np.concatenate([cur_x, cur_pos], axis=2).reshape(-1, cur_x.shape[2] + 3), sd, p + "ga.mlp")
    f0 = g.reshape(n_obj, cur_pos.shape[1], -1).max(axis=1)
    f1 = np.maximum(f0 @ sd[p + "lin1.weight"].T + sd[p + "lin1.bias"], F32(0))
    f2 = np.maximum(f1 @ sd[p + "lin2.weight"].T + sd[p + "lin2.bias"], F32(0))
    if return_levels:
        return f2, levels, f0
    return f2


def _lowbias32(x):
    M = np.uint64(0xFFFFFFFF)
    x = x & M
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & M
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & M
    x ^= x >> np.uint64(16)
    return x


def sample_object_points(xyz: np.ndarray, rgb: np.ndarray, point_offsets: np.ndarray, seed: int, num: int = 256,
                         transform: str = "fixed", rotate_deg: float = 120.0):
    """FixedPoints(num) (indices with replacement) and, by ``transform``: "fixed" nothing else (`--no_pc_augment`, every published
    command: evaluation/pipeline.py:215-216, training/coarse.py:182-184), "normalize" NormalizeScale (centre on the sample mean, scale
    by 0.999999 / max |coord|; evaluation/pipeline.py:217-218), "rotate_normalize" RandomRotate(rotate_deg, axis=2) before it
    (training/coarse.py:185-192) — per object (dataloading/kitti360pose/utils.py:138-143), with the build's counter-based draws
    (include/t2l.h: t2l_sample_object_points) in place of numpy's global RNG."""
    n_obj = len(point_offsets) - 1
    pos = np.zeros((n_obj, num, 3), dtype=F32)
    col = np.zeros((n_obj, num, 3), dtype=F32)
    j = np.arange(num, dtype=np.uint64)
    for o in range(n_obj):
        p0, n = int(point_offsets[o]), int(point_offsets[o + 1] - point_offsets[o])
        okey = np.uint64((seed ^ ((o * 0x85EBCA77) & 0xFFFFFFFF)) & 0xFFFFFFFF)
        x = _lowbias32(j * np.uint64(0x9E3779B1) + okey)
        idx = (((x >> np.uint64(8)) * np.uint64(n)) >> np.uint64(24)).astype(np.int64)
        p = xyz[p0 + idx].astype(F32)
        if transform == "rotate_normalize":
            u = F32(int(_lowbias32(np.uint64(0xA5A5A5A5) + okey) >> np.uint64(8))) * F32(1.0 / 16777216.0)
            ang = F32(F32(rotate_deg) * F32(0.017453292519943295)) * (F32(2.0) * u - F32(1.0))
            c, sn = F32(np.cos(ang)), F32(np.sin(ang))
            p = np.stack([p[:, 0] * c - p[:, 1] * sn, p[:, 0] * sn + p[:, 1] * c, p[:, 2]], axis=1).astype(F32)
        if transform != "fixed":
            p = p - p.mean(axis=0, dtype=np.float64).astype(F32)
            p = p * (F32(1.0) / np.abs(p).max() * F32(0.999999))
        pos[o] = p
        col[o] = rgb[p0 + idx]
    return pos, col
