"""Reader for the reference's on-disk KITTI360Pose data — ``<base>/cells/<scene>.pkl`` and ``<base>/poses/<scene>.pkl``
(dataloading/kitti360pose/base.py:40-48) — WITHOUT the reference's classes on ``sys.path``.

The pickles hold instances of ``datapreparation.kitti360pose.imports.{Cell, Object3d, Pose, DescriptionBestCell,
DescriptionPoseCell}`` (plain attribute bags, imports.py:8-245). Unpickling normally imports those modules (and with them
cv2, easydict, ...). ``KittiUnpickler.find_class`` instead resolves every class of that package to a small record type
defined here, which receives the pickled ``__dict__`` unchanged: same attribute names, nothing else. The records are what
the rest of this package duck-types against (``.label/.xyz/.rgb``, ``.id/.objects/.cell_size/.bbox_w/.get_center()``,
``.pose_w/.cell_id/.descriptions[*].direction/.object_color_text/.object_label``).

On top of that, the dataset surface ``evaluation.pipeline`` / ``training.coarse.eval_epoch`` consume
(dataloading/kitti360pose/cells.py:36-205), eval-time only (no flipping / hint shuffling augmentation):

    ds = Kitti360PoseDataset(base_path, scene_names)           # ~ Kitti360CoarseDatasetMulti
    ds.all_cells, ds.all_poses, ds.get_cell_dataset(), ds[i] -> {"poses","cells","objects","object_points","texts","cell_ids",...}
    dl = DataLoader(ds, batch_size=..., collate_fn=Kitti360PoseDataset.collate_fn)
    db = CellDatabase.build(model, ds.get_cell_dataset())

``object_points``: the reference hands PointNet++ a PyG batch of FixedPoints(256)+NormalizeScale samples per cell
(dataloading/kitti360pose/utils.py:91-147). Here ``object_points="sample"`` builds the same batches with
``packing.sample_object_points`` (host, seeded), ``None`` skips them (class_embed mode does not read them).
"""
from __future__ import annotations

import io
import os.path as osp
import pickle
from typing import Dict, List, Optional, Sequence

import numpy as np

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


class CellOnlyDataset:
    """cells.py:187-205 ``Kitti360CoarseCellOnlyDataset``: one item per database cell, in ``all_cells`` order."""

    def __init__(self, cells: Sequence[CellRecord], object_points: Optional[str] = None, seed: int = 0):
        self.cells = list(cells)
        self._points = object_points
        self._seed = seed

    def __len__(self):
        return len(self.cells)

    def _object_points(self, cell, idx):
        if self._points is None:
            return None
        rng = np.random.default_rng([self._seed, idx])
        return packing.sample_object_points([cell.objects], 256, rng)[0]

    def __getitem__(self, idx):
        cell = self.cells[idx]
        return {"cells": cell, "cell_ids": cell.id, "objects": cell.objects, "object_points": self._object_points(cell, idx)}


class Kitti360PoseDataset:
    """One item per pose over several scenes (cells.py:36-185), evaluation settings."""

    def __init__(self, base_path: str, scene_names: Sequence[str], object_points: Optional[str] = None, seed: int = 0):
        if object_points not in (None, "sample"):
            raise ValueError("object_points must be None or 'sample'")
        self.scene_names = list(scene_names)
        self._points, self._seed = object_points, seed
        self.all_cells: List[CellRecord] = []
        self.all_poses: List[PoseRecord] = []
        self._pose_scene: List[str] = []
        for s in self.scene_names:
            cells, poses = load_scene(base_path, s)
            self.all_cells.extend(cells)
            self.all_poses.extend(poses)
            self._pose_scene.extend([s] * len(poses))
        ids = [c.id for c in self.all_cells]
        if len(set(ids)) != len(ids):
            raise ValueError("cell ids repeat across scenes")  # cells.py:137-138
        self.cells_dict: Dict[str, CellRecord] = {c.id: c for c in self.all_cells}
        self._cell_row = {c.id: i for i, c in enumerate(self.all_cells)}
        self.hint_descriptions = [hint_sentences(p) for p in self.all_poses]

    def __len__(self):
        return len(self.all_poses)

    def __getitem__(self, idx):
        pose = self.all_poses[idx]
        cell = self.cells_dict[pose.cell_id]
        hints = self.hint_descriptions[idx]
        pts = None
        if self._points is not None:
            rng = np.random.default_rng([self._seed, self._cell_row[cell.id]])
            pts = packing.sample_object_points([cell.objects], 256, rng)[0]
        return {"poses": pose, "cells": cell, "objects": cell.objects, "object_points": pts, "texts": " ".join(hints),
                "cell_ids": pose.cell_id, "scene_names": self._pose_scene[idx], "debug_hint_descriptions": hints}

    def get_known_classes(self):
        return list(packing.KNOWN_CLASS)

    def get_cell_dataset(self) -> CellOnlyDataset:
        return CellOnlyDataset(self.all_cells, self._points, self._seed)

    @staticmethod
    def collate_fn(data):
        return {key: [d[key] for d in data] for key in data[0].keys()}
