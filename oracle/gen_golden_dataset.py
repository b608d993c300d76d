"""TEST INFRASTRUCTURE — build-container only.

Writes a tiny KITTI360Pose directory (two scenes) with the reference's OWN classes — ``Cell / Object3d / Pose /
DescriptionBestCell`` pickled exactly as ``datapreparation`` leaves them on disk (dataloading/kitti360pose/base.py:40-48
reads them back) — into ``tests/golden/k360_tiny/{cells,poses}/<scene>.pkl``, and next to them ``k360_tiny.npz``: the same
content as plain arrays plus what the reference's own ``Kitti360BaseDataset`` derives from the pickles (hint sentences,
item texts, item order). The pickles are DATA (object graphs naming the reference's classes, no code); the test
(tests/test_kitti_reader.py) reads them with text2loc_amd.kitti360pose — no reference on sys.path — and must reproduce
the arrays. Re-run: ``python oracle/gen_golden_dataset.py`` (deterministic).
"""
from __future__ import annotations

import os.path as osp
import shutil
import sys

import numpy as np

HERE = osp.dirname(osp.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as H  # noqa: E402

H.setup_reference_imports()

from text2loc_amd import synth  # noqa: E402

OUT = osp.join(H.REPO, "tests", "golden", "k360_tiny")
SCENES = ["2013_05_28_drive_0010_sync", "2013_05_28_drive_0003_sync"]


def small_cells(n, seed):
    c = synth.make_cells(n, seed=seed, min_obj=3, max_obj=7)
    c["n_pts"] = np.minimum(c["n_pts"], 60).astype(np.float32)  # keep the fixture small
    return c


def main():
    from dataloading.kitti360pose.base import Kitti360BaseDataset

    shutil.rmtree(OUT, ignore_errors=True)
    arrays = {"scenes": np.array(SCENES)}
    for si, scene in enumerate(SCENES):
        cells_np = small_cells(5 + si, seed=40 + si)
        objects = H.build_objects(cells_np, seed=40 + si)
        cells, poses = H.write_dataset(OUT, objects, seed=si, scene=scene, n_poses=4 + si, grid=3)
        ds = Kitti360BaseDataset(OUT, scene)  # the reference's own reader of what was just written
        p = f"s{si}_"
        arrays[p + "cell_ids"] = np.array([c.id for c in ds.cells])
        arrays[p + "cell_bbox_w"] = np.array([c.bbox_w for c in ds.cells], dtype=np.float64)
        arrays[p + "cell_size"] = np.array([c.cell_size for c in ds.cells], dtype=np.float64)
        arrays[p + "cell_center"] = np.array([c.get_center() for c in ds.cells], dtype=np.float64)
        arrays[p + "obj_counts"] = np.array([len(c.objects) for c in ds.cells], dtype=np.int64)
        flat = [o for c in ds.cells for o in c.objects]
        arrays[p + "obj_label"] = np.array([o.label for o in flat])
        arrays[p + "obj_id"] = np.array([o.id for o in flat], dtype=np.int64)
        arrays[p + "obj_npts"] = np.array([len(o.xyz) for o in flat], dtype=np.int64)
        arrays[p + "obj_xyz"] = np.concatenate([o.xyz for o in flat]).astype(np.float64)
        arrays[p + "obj_rgb"] = np.concatenate([o.rgb for o in flat]).astype(np.float32)
        arrays[p + "obj_color_text"] = np.array([o.get_color_text() for o in flat])
        arrays[p + "obj_center"] = np.array([o.get_center() for o in flat], dtype=np.float64)
        arrays[p + "pose_w"] = np.array([q.pose_w for q in ds.poses], dtype=np.float64)
        arrays[p + "pose_cell_id"] = np.array([q.cell_id for q in ds.poses])
        arrays[p + "pose_in_cell"] = np.array([q.pose for q in ds.poses], dtype=np.float64)
        arrays[p + "hints"] = np.array([h for hs in ds.hint_descriptions for h in hs])  # base.py:50-68
        arrays[p + "hints_per_pose"] = np.array([len(hs) for hs in ds.hint_descriptions], dtype=np.int64)
        arrays[p + "known_classes"] = np.array(ds.get_known_classes())
    np.savez(osp.join(osp.dirname(OUT), "k360_tiny.npz"), **arrays)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
