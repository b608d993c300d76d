"""Host-side packer: the reference's ``List[List[Object3d]]`` -> the engine's packed SoA (include/t2l.h).

This is the build's counterpart of the Python loops inside ``ObjectEncoder.forward``
(models/object_encoder.py:74-84,121-145): per object the class index, colour index, mean rgb, mean xyz
and point count. The reference recomputes these reductions on every forward call (62 % of its
``encode_objects`` wall time); here they are computed once per object and cached on it.

Objects are duck-typed exactly like the reference's ``Object3d`` (datapreparation/kitti360pose/imports.py:
8-83): ``.label``, ``.xyz [n,3]``, ``.rgb [n,3]``; if they carry the reference's own ``get_center`` /
``get_color_rgb`` / ``get_color_text`` methods those are NOT called — the reductions below are this
package's own (numpy, same formulas), so the packer works on plain containers too.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np

from .tables import COLOR_NAMES, COLORS, KNOWN_CLASS  # data tables (utils.py:48-69,210-231)


def class_table(known_classes: Sequence[str]) -> Dict[str, int]:
    """known_classes -> index, 0 reserved for '<unk>'/padding (models/object_encoder.py:31-33)."""
    t = {c: i + 1 for i, c in enumerate(known_classes)}
    t["<unk>"] = 0
    return t


def color_table() -> Dict[str, int]:
    """Colour name -> row of color_embedding. The reference ignores its ``known_colors`` argument and always
    enumerates COLOR_NAMES (object_encoder.py:35-37): the duplicate 'gray' collapses 1 -> 4 and
    'dark-green' shares row 0 with the padding index."""
    t = {c: i for i, c in enumerate(COLOR_NAMES)}
    t["<unk>"] = 0
    return t


def object_features(obj) -> tuple:
    """(mean rgb f64[3], COLORS argmin, mean xyz f64[3], n_points) — imports.py:28-41, cached on the object."""
    cached = getattr(obj, "_t2l_feat", None)
    if cached is not None and cached[4] == (id(obj.xyz), id(obj.rgb)):
        return cached[:4]
    rgb = np.mean(obj.rgb, axis=0)
    cidx = int(np.argmin(np.linalg.norm(rgb - COLORS, axis=1)))
    center = np.mean(obj.xyz, axis=0)
    out = (rgb, cidx, center, len(obj.xyz))
    try:
        obj._t2l_feat = out + ((id(obj.xyz), id(obj.rgb)),)
    except Exception:
        pass
    return out


def pack_cells(objects: List[List[object]], known_classes: Dict[str, int], known_colors: Optional[Dict[str, int]] = None,
               pn_feat: Optional[Sequence[np.ndarray]] = None) -> Dict[str, np.ndarray]:
    """objects: per cell, the cell's objects in dataset order. pn_feat: optional per-cell ``[n_i,256]`` arrays of
    PointNet++ ``features2`` (needed when class_embed is off). Returns numpy arrays laid out as t2l_packed_cells."""
    known_colors = known_colors or color_table()
    counts = np.array([len(o) for o in objects], dtype=np.int32)
    offsets = np.zeros(len(objects) + 1, dtype=np.int32)
    np.cumsum(counts, out=offsets[1:])
    total = int(offsets[-1])
    class_idx = np.empty(total, dtype=np.int32)
    color_idx = np.empty(total, dtype=np.int32)
    rgb = np.empty((total, 3), dtype=np.float32)
    center = np.empty((total, 3), dtype=np.float32)
    n_pts = np.empty(total, dtype=np.float32)
    i = 0
    for objs in objects:
        for o in objs:
            c_rgb, cidx, c_xyz, n = object_features(o)
            class_idx[i] = known_classes.get(o.label, 0)  # object_encoder.py:81
            color_idx[i] = known_colors[COLOR_NAMES[cidx]]  # object_encoder.py:83
            rgb[i] = c_rgb  # torch.tensor(..., dtype=torch.float), object_encoder.py:124-127
            center[i] = c_xyz  # object_encoder.py:133-134
            n_pts[i] = n  # object_encoder.py:141-143
            i += 1
    out = {"counts": counts, "offsets": offsets, "class_idx": class_idx, "color_idx": color_idx, "rgb": rgb,
           "center": center, "n_pts": n_pts}
    if pn_feat is not None:
        feats = [np.asarray(f, dtype=np.float32).reshape(-1, 256) for f in pn_feat]
        if [len(f) for f in feats] != counts.tolist():
            raise ValueError("pn_feat must hold one [n_i,256] array per cell, matching the object counts")
        out["pn_feat"] = np.concatenate(feats, axis=0) if feats else np.zeros((0, 256), np.float32)
    return out


def pack_cells_gpu(engine, objects: List[List[object]], known_classes: Dict[str, int],
                   known_colors: Optional[Dict[str, int]] = None, device="cuda") -> Dict[str, "torch.Tensor"]:
    """Same packed SoA as ``pack_cells`` but the per-object reductions over the raw points run on the GPU
    (t2l_reduce_objects, one HBM pass over 24 B/point) instead of ~3 numpy reductions per object on the host.
    The points are concatenated on the host once and copied over; class indices come from the labels."""
    import torch

    known_colors = known_colors or color_table()
    flat = [o for objs in objects for o in objs]
    counts = np.array([len(o) for o in objects], dtype=np.int32)
    offsets = np.zeros(len(objects) + 1, dtype=np.int32)
    np.cumsum(counts, out=offsets[1:])
    npts = np.array([len(o.xyz) for o in flat], dtype=np.int64)
    poff = np.zeros(len(flat) + 1, dtype=np.int64)
    np.cumsum(npts, out=poff[1:])
    xyz = np.concatenate([np.asarray(o.xyz, dtype=np.float32) for o in flat], axis=0) if flat else np.zeros((0, 3), np.float32)
    rgb = np.concatenate([np.asarray(o.rgb, dtype=np.float32) for o in flat], axis=0) if flat else np.zeros((0, 3), np.float32)
    rows = np.array([known_colors[c] for c in COLOR_NAMES], dtype=np.int32)
    red = engine.reduce_objects(torch.from_numpy(xyz).to(device), torch.from_numpy(rgb).to(device),
                                poff, COLORS, rows)
    red["offsets"] = torch.from_numpy(offsets).to(device)
    red["class_idx"] = torch.from_numpy(np.array([known_classes.get(o.label, 0) for o in flat], dtype=np.int32)).to(device)
    return red


class PackedCellSet:
    """Every cell of a database flattened ONCE on the host: object counts, labels, and the raw points of all objects concatenated
    (f32, 24 B/point) with per-object point offsets — the input of ``t2l_reduce_objects`` (a1) and ``t2l_sample_object_points``
    (the dataloader's FixedPoints batches) for the WHOLE dataset. The reference walks the pickled objects again on every forward
    call (models/object_encoder.py:74-84,121-145) and once more per item in the dataloader (dataloading/kitti360pose/utils.py:
    91-147); here the walk happens once per dataset, the device copies and the (weight-independent) per-object reductions are
    cached per GPU, and ``CellRetrievalNetwork.encode_cell_set`` encodes from them in chunks of thousands of cells.

    ``cells``: anything with ``.objects`` (each ``.label``, ``.xyz [n,3]``, ``.rgb [n,3]``) and ``.id``."""

    def __init__(self, cells):
        flat = [o for c in cells for o in c.objects]
        self.cell_ids = [c.id for c in cells]
        self.counts = np.array([len(c.objects) for c in cells], dtype=np.int32)
        self.offsets = np.zeros(len(self.counts) + 1, dtype=np.int32)
        np.cumsum(self.counts, out=self.offsets[1:])
        self.labels = [o.label for o in flat]
        npts = np.fromiter((len(o.xyz) for o in flat), dtype=np.int64, count=len(flat))
        self.point_offsets = np.zeros(len(flat) + 1, dtype=np.int64)
        np.cumsum(npts, out=self.point_offsets[1:])
        if flat:
            self.xyz = np.concatenate([o.xyz for o in flat], axis=0, dtype=np.float32, casting="same_kind")
            self.rgb = np.concatenate([o.rgb for o in flat], axis=0, dtype=np.float32, casting="same_kind")
        else:
            self.xyz = self.rgb = np.zeros((0, 3), np.float32)
        self._class: Dict[tuple, np.ndarray] = {}
        self._dev: Dict[str, dict] = {}

    @property
    def n_cells(self) -> int:
        return int(len(self.counts))

    @property
    def n_objects(self) -> int:
        return int(self.offsets[-1])

    @property
    def n_points(self) -> int:
        return int(self.point_offsets[-1])

    def class_idx(self, known_classes: Dict[str, int]) -> np.ndarray:
        """known_classes.get(label, 0) per object (models/object_encoder.py:81), cached per class table."""
        key = tuple(sorted(known_classes.items()))
        out = self._class.get(key)
        if out is None:
            get = known_classes.get
            out = self._class[key] = np.fromiter((get(lb, 0) for lb in self.labels), dtype=np.int32, count=len(self.labels))
        return out

    def on_device(self, device) -> Dict[str, "torch.Tensor"]:
        """The points, offsets (and later the reductions) as tensors on ``device``, copied once."""
        import torch

        key = str(torch.device(device))
        d = self._dev.get(key)
        if d is None:
            d = self._dev[key] = {"xyz": torch.from_numpy(self.xyz).to(device), "rgb": torch.from_numpy(self.rgb).to(device),
                                  "point_offsets": torch.from_numpy(self.point_offsets).to(device),
                                  "offsets": torch.from_numpy(self.offsets).to(device)}
        return d

    def reduced(self, engine, device, known_colors: Optional[Dict[str, int]] = None) -> Dict[str, "torch.Tensor"]:
        """a1 for every object of the dataset in one launch pair (t2l_reduce_objects): mean rgb, nearest colour row, mean xyz,
        point count — functions of the points alone, so cached beside the device copy."""
        d = self.on_device(device)
        if "red" not in d:
            known_colors = known_colors or color_table()
            rows = np.array([known_colors[c] for c in COLOR_NAMES], dtype=np.int32)
            d["red"] = engine.reduce_objects(d["xyz"], d["rgb"], self.point_offsets, COLORS, rows)
        return d["red"]

    def class_idx_on(self, device, known_classes: Dict[str, int]):
        import torch

        d = self.on_device(device)
        key = ("class", tuple(sorted(known_classes.items())))
        if key not in d:
            d[key] = torch.from_numpy(self.class_idx(known_classes)).to(device)
        return d[key]

    def release_device(self, device=None):
        """Drop the device copies (all devices when ``device`` is None): the raw points are 24 B each."""
        import torch

        if device is None:
            self._dev.clear()
        else:
            self._dev.pop(str(torch.device(device)), None)


# transform name -> t2l_sample_object_points flags (include/t2l.h: T2L_SAMPLE_NORMALIZE = 1, T2L_SAMPLE_ROTATE = 2)
POINT_TRANSFORMS = {"fixed": 0, "normalize": 1, "rotate_normalize": 3}


def point_transform_from_args(args, train: bool = False, fine: bool = False) -> str:
    """The transform the reference's scripts build from their arguments: ``--no_pc_augment`` (``--no_pc_augment_fine`` for the
    fine dataset) -> FixedPoints only — every published command passes it (README.md:89,107,125-126,139-140); without it
    FixedPoints + NormalizeScale for evaluation / validation and + RandomRotate(120, axis=2) for training
    (evaluation/pipeline.py:215-223, training/coarse.py:182-193)."""
    flag = bool(getattr(args, "no_pc_augment_fine" if fine else "no_pc_augment", False))
    if flag:
        return "fixed"
    return "rotate_normalize" if train else "normalize"


def sample_object_points(objects: List[List[object]], num: int = 256, rng: Optional[np.random.Generator] = None,
                         transform: str = "fixed", rotate_deg: float = 120.0):
    """Per cell, the point batch the reference's dataloader hands PointNet++ (dataloading/kitti360pose/utils.py:91-147):
    ``FixedPoints(num)`` — ``num`` indices drawn with replacement — and, by ``transform`` (see ``point_transform_from_args``):
    "fixed" nothing else (`--no_pc_augment`, the published configuration: positions stay in the cell-normalised frame),
    "normalize" ``NormalizeScale`` (centre on the mean, scale by 0.999999 / max|coordinate|), "rotate_normalize"
    ``RandomRotate(rotate_deg, axis=2)`` then ``NormalizeScale``. Returns a list of dicts
    ``{"pos": f32[n_i*num,3], "x": f32[n_i*num,3]}`` (object-major), the duck type ``encode_objects`` accepts."""
    if transform not in POINT_TRANSFORMS:
        raise ValueError(f"transform must be one of {sorted(POINT_TRANSFORMS)}, got {transform!r}")
    rng = rng or np.random.default_rng()
    out = []
    for objs in objects:
        pos, x = [], []
        for o in objs:
            xyz, rgb = np.asarray(o.xyz, dtype=np.float32), np.asarray(o.rgb, dtype=np.float32)
            sel = rng.integers(0, len(xyz), size=num)
            p = xyz[sel]
            if transform == "rotate_normalize":
                ang = np.deg2rad(rotate_deg) * rng.uniform(-1.0, 1.0)
                c, sn = np.float32(np.cos(ang)), np.float32(np.sin(ang))
                p = p @ np.array([[c, sn, 0], [-sn, c, 0], [0, 0, 1]], dtype=np.float32)
            if transform != "fixed":
                p = p - p.mean(axis=0, keepdims=True)
                p = p * (np.float32(0.999999) / max(float(np.abs(p).max()), 1e-12))
            pos.append(p.astype(np.float32))
            x.append(rgb[sel])
        out.append({"pos": np.concatenate(pos, axis=0), "x": np.concatenate(x, axis=0)})
    return out


def sample_object_points_gpu(engine, objects: List[List[object]], device="cuda", seed: int = 0, transform: str = "fixed",
                             rotate_deg: float = 120.0):
    """``sample_object_points`` on the GPU (t2l_sample_object_points): the raw points are concatenated and copied over once,
    FixedPoints(256) and the transform run as one kernel (counter-based draw, so equal to the host sampler in distribution,
    not in the indices). Returns the same per-cell ``{"pos", "x"}`` batches, as CUDA tensors."""
    import torch

    flat = [o for objs in objects for o in objs]
    npts = np.array([len(o.xyz) for o in flat], dtype=np.int64)
    poff = np.zeros(len(flat) + 1, dtype=np.int64)
    np.cumsum(npts, out=poff[1:])
    xyz = np.concatenate([np.asarray(o.xyz, dtype=np.float32) for o in flat], axis=0)
    rgb = np.concatenate([np.asarray(o.rgb, dtype=np.float32) for o in flat], axis=0)
    pos, col = engine.sample_object_points(torch.from_numpy(xyz).to(device), torch.from_numpy(rgb).to(device),
                                           torch.from_numpy(poff).to(device), seed, transform=transform, rotate_deg=rotate_deg)
    out, lo = [], 0
    for objs in objects:
        hi = lo + len(objs)
        out.append({"pos": pos[lo:hi].reshape(-1, 3), "x": col[lo:hi].reshape(-1, 3)})
        lo = hi
    return out


def to_device(packed: Dict[str, np.ndarray], device) -> Dict[str, "torch.Tensor"]:
    import torch

    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(device, non_blocking=True) for k, v in packed.items()
            if k != "counts"}


__all__ = ["KNOWN_CLASS", "COLOR_NAMES", "class_table", "color_table", "object_features", "pack_cells", "PackedCellSet",
           "pack_cells_gpu", "sample_object_points", "sample_object_points_gpu", "to_device", "POINT_TRANSFORMS",
           "point_transform_from_args"]
