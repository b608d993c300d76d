"""Reader for the reference's on-disk KITTI360Pose data — ``<base>/cells/<scene>.pkl`` and ``<base>/poses/<scene>.pkl``
(dataloading/kitti360pose/base.py:40-48) — WITHOUT the reference's classes on ``sys.path``.

The pickles hold instances of ``datapreparation.kitti360pose.imports.{Cell, Object3d, Pose, DescriptionBestCell,
DescriptionPoseCell}`` (plain attribute bags, imports.py:8-245). Unpickling normally imports those modules (and with them
cv2, easydict, ...). ``KittiUnpickler.find_class`` instead resolves every class of that package to a small record type
defined here, which receives the pickled ``__dict__`` unchanged: same attribute names, nothing else. The records are what
the rest of this package duck-types against (``.label/.xyz/.rgb``, ``.id/.objects/.cell_size/.bbox_w/.get_center()``,
``.pose_w/.cell_id/.descriptions[*].direction/.object_color_text/.object_label``).

On top of that, the dataset surface ``evaluation.pipeline`` / ``training.coarse.eval_epoch`` consume
(dataloading/kitti360pose/cells.py:36-205), with the training-time augmentations (``shuffle_hints``, ``flip_poses``):

    ds = Kitti360PoseDataset(base_path, scene_names)           # ~ Kitti360CoarseDatasetMulti(..., transform, shuffle_hints, flip_poses)
    ds.all_cells, ds.all_poses, ds.get_cell_dataset(), ds[i] -> {"poses","cells","objects","object_points","texts","cell_ids",...}
    dl = DataLoader(ds, batch_size=..., collate_fn=Kitti360PoseDataset.collate_fn)
    db = CellDatabase.build(model, ds.get_cell_dataset())

``object_points``: the reference hands PointNet++ a PyG batch of per-object FixedPoints(256) samples per cell — and nothing
else under ``--no_pc_augment``, which every published command passes; NormalizeScale (and RandomRotate in training) only
without the flag (dataloading/kitti360pose/utils.py:91-147, evaluation/pipeline.py:215-223, training/coarse.py:182-193).
``object_points="sample"`` builds those batches with ``packing.sample_object_points`` (host, seeded) under ``transform``
("fixed" by default = the published configuration); ``None`` skips them (class_embed mode does not read them).
"""
from __future__ import annotations

import io
import os.path as osp
import pickle
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch.utils.data

from . import packing

_REF_PACKAGE = "datapreparation.kitti360pose"


class Record:
    """Attribute bag standing in for a pickled reference object (pickle restores ``__dict__`` through ``__setstate__``
    or, without one, by updating ``__dict__`` — both land here)."""

    _ref_class = "object"

    def __setstate__(self, state):
        if isinstance(state, tuple) and len(state) == 2:  # (dict, slots) form
            for part in state:
                if part:
                    self.__dict__.update(part)
        elif state:
            self.__dict__.update(state)

    def __repr__(self):
        return f"<{self._ref_class} {getattr(self, 'id', getattr(self, 'label', ''))}>"


class ObjectRecord(Record):
    """imports.py:8-83 ``Object3d``: id, instance_id, xyz f64[n,3], rgb [n,3], label."""

    _ref_class = "Object3d"

    def get_center(self):
        return np.mean(self.xyz, axis=0)

    def get_color_rgb(self):
        return np.mean(self.rgb, axis=0)

    def get_color_text(self):
        return packing.COLOR_NAMES[int(np.argmin(np.linalg.norm(np.mean(self.rgb, axis=0) - packing.COLORS, axis=1)))]


class CellRecord(Record):
    """imports.py:221-245 ``Cell``: id, scene_name, objects, cell_size, bbox_w."""

    _ref_class = "Cell"

    def get_center(self):
        b = np.asarray(self.bbox_w)
        return 0.5 * (b[0:3] + b[3:6])


class PoseRecord(Record):
    """imports.py:175-218 ``Pose``: pose, pose_w, cell_id, scene_name, descriptions, described_by."""

    _ref_class = "Pose"


class HintRecord(Record):
    """imports.py:86-172 ``DescriptionPoseCell`` / ``DescriptionBestCell``: direction, object_label, object_color_text, ..."""

    _ref_class = "Description"


_CLASS_MAP = {"Object3d": ObjectRecord, "Cell": CellRecord, "Pose": PoseRecord, "DescriptionBestCell": HintRecord,
              "DescriptionPoseCell": HintRecord, "Description": HintRecord}


# Exact (module, name) pairs a KITTI360Pose pickle may resolve besides the reference's own record classes: what numpy
# arrays / scalars and plain containers need to rebuild themselves (both numpy 1.x and 2.x module spellings). Nothing else
# — in particular nothing of ``builtins`` that can call or import (eval, exec, getattr, __import__) — is reachable, so a
# crafted dataset pickle cannot execute code through ``load_pickle``.
_SAFE_GLOBALS = {
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
    ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer"),
    ("numpy", "ndarray"), ("numpy", "dtype"),
    ("_codecs", "encode"),
    ("collections", "OrderedDict"),
} | {("builtins", n) for n in ("list", "dict", "set", "tuple", "str", "int", "float", "bool", "bytes", "bytearray", "complex",
                                 "frozenset", "slice", "range", "object")}


class KittiUnpickler(pickle.Unpickler):
    """Resolves classes of the reference's ``datapreparation.kitti360pose`` package to the records above and the exact
    numpy / container reconstructors of ``_SAFE_GLOBALS``; every other global raises ``UnpicklingError``. Unknown classes
    of the reference's package become generic ``Record``s."""

    def find_class(self, module, name):
        if module == _REF_PACKAGE or module.startswith(_REF_PACKAGE + ".") or module.startswith("datapreparation.kitti360."):
            return _CLASS_MAP.get(name, Record)  # dataloading/__init__.py:8-10 aliases the old package name
        if (module, name) in _SAFE_GLOBALS:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"refusing to resolve {module}.{name} while reading a KITTI360Pose pickle")


def load_pickle(path: str):
    with open(path, "rb") as f:
        return KittiUnpickler(io.BufferedReader(f)).load()


def load_scene(base_path: str, scene_name: str):
    """(cells, poses) of one scene, as base.py:40-48 reads them (ids must be unique inside the scene)."""
    cells = load_pickle(osp.join(base_path, "cells", f"{scene_name}.pkl"))
    poses = load_pickle(osp.join(base_path, "poses", f"{scene_name}.pkl"))
    ids = [c.id for c in cells]
    if len(set(ids)) != len(ids):
        raise ValueError(f"{scene_name}: cell ids repeat")
    return cells, poses


def hint_sentences(pose) -> List[str]:
    """base.py:60-68: one template sentence per description."""
    return [f"The pose is {d.direction} of a {d.object_color_text} {d.object_label}." for d in pose.descriptions]


def _swap_words(text: str, a: str, b: str) -> str:
    return text.replace(a, a + "-flipped").replace(b, a).replace(a + "-flipped", b)


def flip_pose_in_cell(pose, cell, text: str, direction: int):
    """dataloading/kitti360pose/utils.py:15-88 (the three-value form the coarse dataset uses): mirror the cell about x = 0.5
    (``direction`` +1, "horizontally": east <-> west in the text) or y = 0.5 (-1, "vertically": north <-> south). Pose and
    cell are copied first; object points, the pose-in-cell and every description's ``closest_point`` are mirrored, the
    descriptions' ``direction`` fields are NOT (as in the reference: only the text changes)."""
    import copy

    if direction not in (-1, 1):
        raise ValueError("direction must be +1 (horizontal) or -1 (vertical)")
    pose, cell = copy.deepcopy(pose), copy.deepcopy(cell)
    ax = 0 if direction == 1 else 1
    pose.pose[ax] = 1.0 - pose.pose[ax]
    for obj in cell.objects:
        obj.xyz[:, ax] = 1 - obj.xyz[:, ax]
        if hasattr(obj, "_t2l_feat"):
            del obj._t2l_feat  # cached reductions of the un-mirrored points
    for d in pose.descriptions:
        d.closest_point[ax] = 1.0 - d.closest_point[ax]
    text = _swap_words(text, "east", "west") if direction == 1 else _swap_words(text, "north", "south")
    return pose, cell, text


class CellOnlyDataset:
    """cells.py:187-205 ``Kitti360CoarseCellOnlyDataset``: one item per database cell, in ``all_cells`` order (never flipped
    or shuffled, as in the reference)."""

    def __init__(self, cells: Sequence[CellRecord], object_points: Optional[str] = None, seed: int = 0, transform: str = "fixed",
                 packed_holder: Optional[dict] = None):
        self.cells = list(cells)
        self._points = object_points
        self._seed = seed
        self._transform = "normalize" if transform == "rotate_normalize" else transform  # val transform: no rotation
        self._packed_holder = packed_holder if packed_holder is not None else {}

    def __len__(self):
        return len(self.cells)

    # ---- the engine-sized path: every cell of the database flattened ONCE (packing.PackedCellSet) ---------------------------
    # ``coarse.eval_epoch`` / ``CellDatabase.build`` encode through it when it is there (CellRetrievalNetwork.encode_cell_set):
    # one host pass over the objects per DATASET instead of one per forward call, the per-object reductions, FixedPoints and
    # PointNet++ on the GPU over chunks of thousands of cells whatever ``args.batch_size`` says. The holder is shared with the
    # parent ``Kitti360PoseDataset`` (``get_cell_dataset`` builds a new CellOnlyDataset per call: the three validation passes per
    # epoch of training/coarse.py:283-287 must not flatten the database three times).
    @property
    def wants_points(self) -> bool:
        return self._points is not None

    @property
    def point_transform(self) -> str:
        return self._transform

    @property
    def point_seed(self) -> int:
        return int(self._seed)

    def packed(self):
        ps = self._packed_holder.get("set")
        if ps is None:
            ps = self._packed_holder["set"] = packing.PackedCellSet(self.cells)
        return ps

    def _object_points(self, cell, idx):
        if self._points is None:
            return None
        rng = np.random.default_rng([self._seed, idx])
        return packing.sample_object_points([cell.objects], 256, rng, transform=self._transform)[0]

    def __getitem__(self, idx):
        cell = self.cells[idx]
        return {"cells": cell, "cell_ids": cell.id, "objects": cell.objects, "object_points": self._object_points(cell, idx)}


class Kitti360PoseDataset:
    """One item per pose over several scenes (cells.py:36-185 ``Kitti360CoarseDataset`` / ``...Multi``).

    ``transform``: the point transform of ``object_points="sample"`` — "fixed" (FixedPoints(256) only: `--no_pc_augment`,
    what every published command passes and therefore the default), "normalize", "rotate_normalize"
    (``packing.point_transform_from_args(args, train=...)`` picks it the way the reference's scripts do).
    ``shuffle_hints`` / ``flip_poses``: the training-time augmentations of cells.py:80-91 (training/coarse.py:196-202 turns
    both on): the hint sentences are permuted, and with probability 1/2 each the cell is mirrored horizontally and
    vertically (``flip_pose_in_cell``). Their draws come from ``aug_rng`` — a ``numpy.random.RandomState`` consumed with the
    reference's own call sequence (one ``choice`` without replacement over the hints, then two ``choice((True, False))``), so
    ``RandomState(s)`` here reproduces the reference under ``np.random.seed(s)`` item for item; ``None`` = numpy's global
    stream, i.e. exactly what the reference draws from."""

    def __init__(self, base_path: str, scene_names: Sequence[str], object_points: Optional[str] = None, seed: int = 0,
                 transform: str = "fixed", shuffle_hints: bool = False, flip_poses: bool = False, aug_rng=None):
        if object_points not in (None, "sample"):
            raise ValueError("object_points must be None or 'sample'")
        if transform not in packing.POINT_TRANSFORMS:
            raise ValueError(f"transform must be one of {sorted(packing.POINT_TRANSFORMS)}")
        self.scene_names = list(scene_names)
        self._points, self._seed, self._transform = object_points, seed, transform
        self.shuffle_hints, self.flip_poses = bool(shuffle_hints), bool(flip_poses)
        self._aug = aug_rng if aug_rng is not None else np.random
        self._epoch_draw = 0
        self._epoch = 0
        self.all_cells: List[CellRecord] = []
        self.all_poses: List[PoseRecord] = []
        self._pose_scene: List[str] = []
        self._packed_holder: dict = {}
        for s in self.scene_names:
            cells, poses = base_path[s] if isinstance(base_path, dict) else load_scene(base_path, s)
            self.all_cells.extend(cells)
            self.all_poses.extend(poses)
            self._pose_scene.extend([s] * len(poses))
        ids = [c.id for c in self.all_cells]
        if len(set(ids)) != len(ids):
            raise ValueError("cell ids repeat across scenes")  # cells.py:137-138
        self.cells_dict: Dict[str, CellRecord] = {c.id: c for c in self.all_cells}
        self._cell_row = {c.id: i for i, c in enumerate(self.all_cells)}
        self.hint_descriptions = [hint_sentences(p) for p in self.all_poses]

    @classmethod
    def from_records(cls, cells: Sequence[CellRecord], poses: Sequence[PoseRecord], scene_name: str = "in_memory", **kw):
        """The same dataset over records that are already in memory (one scene): what ``load_scene`` would have returned."""
        return cls({scene_name: (list(cells), list(poses))}, [scene_name], **kw)

    def __len__(self):
        return len(self.all_poses)

    def eval_texts(self) -> Optional[List[str]]:
        """Every item's ``texts`` in item order WITHOUT building the items — None when items are augmented (then the texts are a
        function of the draw and only ``__getitem__`` knows them). ``coarse.eval_epoch`` reads the queries from here instead of
        iterating the DataLoader: an evaluation item also carries its cell's point batch (cells.py:91-107), which the query side
        never looks at."""
        if self.shuffle_hints or self.flip_poses:
            return None
        return [" ".join(h) for h in self.hint_descriptions]

    def __getitem__(self, idx):
        pose = self.all_poses[idx]
        cell = self.cells_dict[pose.cell_id]
        hints = self.hint_descriptions[idx]
        if self.shuffle_hints:  # cells.py:80-81
            hints = self._aug.choice(hints, size=len(hints), replace=False)
        text = " ".join(hints)
        if self.flip_poses:     # cells.py:86-90
            if self._aug.choice((True, False)):
                pose, cell, text = flip_pose_in_cell(pose, cell, text, 1)
            if self._aug.choice((True, False)):
                pose, cell, text = flip_pose_in_cell(pose, cell, text, -1)
        pts = None
        if self._points is not None:
            if self._transform == "rotate_normalize":  # a fresh rotation / draw every time an item is fetched
                # keyed on (seed, cell, epoch, DataLoader worker, per-process counter): a worker process starts from a COPY of the
                # parent's counter every epoch, so the counter alone repeated the draws across epochs and collided across workers
                self._epoch_draw += 1
                wi = torch.utils.data.get_worker_info()
                rng = np.random.default_rng([self._seed, self._cell_row[cell.id], self._epoch, 0 if wi is None else wi.id + 1, self._epoch_draw])
            else:
                rng = np.random.default_rng([self._seed, self._cell_row[cell.id]])
            pts = packing.sample_object_points([cell.objects], 256, rng, transform=self._transform)[0]
        return {"poses": pose, "cells": cell, "objects": cell.objects, "object_points": pts, "texts": text,
                "cell_ids": pose.cell_id, "scene_names": self._pose_scene[idx], "debug_hint_descriptions": hints}

    def set_epoch(self, epoch: int):
        """Call once per epoch (as DistributedSampler.set_epoch): part of the key of the augmentation draws under
        transform="rotate_normalize" (DataLoader workers restart from a copy of this object every epoch)."""
        self._epoch = int(epoch)

    def get_known_classes(self):
        return list(packing.KNOWN_CLASS)

    def get_cell_dataset(self) -> CellOnlyDataset:
        return CellOnlyDataset(self.all_cells, self._points, self._seed, self._transform, self._packed_holder)

    @staticmethod
    def collate_fn(data):
        return {key: [d[key] for d in data] for key in data[0].keys()}
