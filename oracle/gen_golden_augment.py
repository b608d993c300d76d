"""TEST INFRASTRUCTURE — build-container only.

Training-time dataset augmentations of the reference (dataloading/kitti360pose/cells.py:80-91: ``shuffle_hints`` /
``flip_poses``; dataloading/kitti360pose/utils.py:15-88 ``flip_pose_in_cell``) run on the committed ``k360_tiny``
directory with the reference's OWN ``Kitti360CoarseDatasetMulti(..., shuffle_hints=True, flip_poses=True)`` under
``np.random.seed(s)``; what every fetched item looks like (text, pose-in-cell, the cell's object points, the
descriptions' closest points) goes to ``tests/golden/k360_augment.npz``. ``text2loc_amd.kitti360pose`` must reproduce it
with ``aug_rng = np.random.RandomState(s)`` (tests/test_kitti_reader.py). Also the four deterministic flip combinations of
one item through ``flip_pose_in_cell`` directly. Re-run: ``python oracle/gen_golden_augment.py`` (deterministic).
"""
from __future__ import annotations

import os.path as osp
import sys

import numpy as np

HERE = osp.dirname(osp.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as H  # noqa: E402

H.setup_reference_imports()

BASE = osp.join(H.REPO, "tests", "golden", "k360_tiny")
SCENES = ["2013_05_28_drive_0010_sync", "2013_05_28_drive_0003_sync"]
SEEDS = [0, 7]


def item_arrays(item):
    pose, cell = item["poses"], item["cells"]
    return (np.asarray(pose.pose, dtype=np.float64), np.concatenate([o.xyz for o in cell.objects]).astype(np.float64),
            np.array([d.closest_point for d in pose.descriptions], dtype=np.float64))


def main():
    import torch_geometric.transforms as T
    from dataloading.kitti360pose.cells import Kitti360CoarseDatasetMulti
    from dataloading.kitti360pose.utils import flip_pose_in_cell

    arrays = {"seeds": np.array(SEEDS), "scenes": np.array(SCENES)}
    ds = Kitti360CoarseDatasetMulti(BASE, SCENES, T.FixedPoints(256), shuffle_hints=True, flip_poses=True)
    for s in SEEDS:
        np.random.seed(s)
        texts, poses, xyz, cps, cnt = [], [], [], [], []
        for i in range(len(ds)):
            it = ds[i]
            p, x, c = item_arrays(it)
            texts.append(it["texts"]); poses.append(p); xyz.append(x); cps.append(c); cnt.append(len(x))
        arrays[f"seed{s}_texts"] = np.array(texts)
        arrays[f"seed{s}_pose"] = np.array(poses)
        arrays[f"seed{s}_xyz"] = np.concatenate(xyz)
        arrays[f"seed{s}_xyz_counts"] = np.array(cnt, dtype=np.int64)
        arrays[f"seed{s}_closest"] = np.concatenate(cps)
    plain = Kitti360CoarseDatasetMulti(BASE, SCENES, T.FixedPoints(256), shuffle_hints=False, flip_poses=False)
    it = plain[1]
    for name, dirs in (("h", [1]), ("v", [-1]), ("hv", [1, -1])):
        pose, cell, text = it["poses"], it["cells"], it["texts"]
        for d in dirs:
            pose, cell, text = flip_pose_in_cell(pose, cell, text, d)
        p, x, c = item_arrays({"poses": pose, "cells": cell})
        arrays[f"flip_{name}_text"] = np.array(text)
        arrays[f"flip_{name}_pose"], arrays[f"flip_{name}_xyz"], arrays[f"flip_{name}_closest"] = p, x, c
    np.savez(osp.join(H.REPO, "tests", "golden", "k360_augment.npz"), **arrays)
    print("wrote k360_augment.npz:", len(ds), "items per seed")


if __name__ == "__main__":
    main()
