"""CPU tier, row f-2 (pickle ingestion): text2loc_amd.kitti360pose reads the reference's on-disk format
(dataloading/kitti360pose/base.py:40-48) with NO reference module importable — the fixture pickles were written by the
reference's own Cell / Object3d / Pose / Description classes (oracle/gen_golden_dataset.py), the expected arrays by the
reference's own Kitti360BaseDataset reading them back."""
import os.path as osp
import pickle
import sys

import numpy as np
import pytest

from tests.conftest import GOLDEN

BASE = osp.join(GOLDEN, "k360_tiny")


def test_reference_modules_are_not_importable_here():
    assert "datapreparation" not in sys.modules
    with pytest.raises(ModuleNotFoundError):
        pickle.load(open(osp.join(BASE, "cells", "2013_05_28_drive_0010_sync.pkl"), "rb"))  # what plain pickle would need


def test_pickles_load_into_records_equal_to_the_reference_view(golden):
    from text2loc_amd import kitti360pose as K

    g = golden("k360_tiny")
    for si, scene in enumerate(g["scenes"]):
        p = f"s{si}_"
        cells, poses = K.load_scene(BASE, str(scene))
        assert all(isinstance(c, K.CellRecord) for c in cells) and all(isinstance(q, K.PoseRecord) for q in poses)
        assert [c.id for c in cells] == g[p + "cell_ids"].tolist()
        assert np.array_equal(np.array([c.bbox_w for c in cells]), g[p + "cell_bbox_w"])
        assert np.array_equal(np.array([c.cell_size for c in cells], dtype=np.float64), g[p + "cell_size"])
        assert np.array_equal(np.array([c.get_center() for c in cells]), g[p + "cell_center"])
        assert [len(c.objects) for c in cells] == g[p + "obj_counts"].tolist()
        flat = [o for c in cells for o in c.objects]
        assert all(isinstance(o, K.ObjectRecord) for o in flat)
        assert [o.label for o in flat] == g[p + "obj_label"].tolist()
        assert [o.id for o in flat] == g[p + "obj_id"].tolist()
        assert np.array_equal(np.concatenate([o.xyz for o in flat]), g[p + "obj_xyz"])
        assert np.array_equal(np.concatenate([o.rgb for o in flat]), g[p + "obj_rgb"])
        assert [o.get_color_text() for o in flat] == g[p + "obj_color_text"].tolist()
        assert np.array_equal(np.array([o.get_center() for o in flat]), g[p + "obj_center"])
        assert np.array_equal(np.array([q.pose_w for q in poses]), g[p + "pose_w"])
        assert [q.cell_id for q in poses] == g[p + "pose_cell_id"].tolist()
        hints = [h for q in poses for h in K.hint_sentences(q)]
        assert hints == g[p + "hints"].tolist()
        assert [len(q.descriptions) for q in poses] == g[p + "hints_per_pose"].tolist()


def test_dataset_surface_feeds_the_packer_and_eval_bookkeeping(golden):
    """Kitti360PoseDataset: all_cells / all_poses / items / cell-only dataset as eval_epoch and CellDatabase.build consume
    them; the host packer's per-object features equal the reference's reductions stored in the fixture."""
    from text2loc_amd import kitti360pose as K
    from text2loc_amd import packing

    g = golden("k360_tiny")
    scenes = [str(s) for s in g["scenes"]]
    ds = K.Kitti360PoseDataset(BASE, scenes, object_points="sample", seed=3)
    n_cells = sum(len(g[f"s{i}_cell_ids"]) for i in range(len(scenes)))
    n_poses = sum(len(g[f"s{i}_pose_w"]) for i in range(len(scenes)))
    assert len(ds.all_cells) == n_cells and len(ds) == len(ds.all_poses) == n_poses
    assert sorted(ds.get_known_classes()) == sorted(g["s0_known_classes"].tolist())
    item = ds[0]
    assert item["cell_ids"] == g["s0_pose_cell_id"][0] and item["cells"].id == item["cell_ids"]
    assert item["texts"] == " ".join(g["s0_hints"][: int(g["s0_hints_per_pose"][0])].tolist())
    n_obj = len(item["objects"])
    assert item["object_points"]["pos"].shape == (n_obj * 256, 3) and item["object_points"]["x"].shape == (n_obj * 256, 3)
    batch = K.Kitti360PoseDataset.collate_fn([ds[0], ds[1]])
    assert set(batch) >= {"texts", "cell_ids", "objects", "object_points", "poses", "cells"} and len(batch["texts"]) == 2
    cds = ds.get_cell_dataset()
    assert len(cds) == n_cells and [cds[i]["cell_ids"] for i in range(n_cells)] == [c.id for c in ds.all_cells]
    assert cds.cells is not None and cds[0]["cells"].cell_size == g["s0_cell_size"][0]
    packed = packing.pack_cells([c.objects for c in ds.all_cells], packing.class_table(packing.KNOWN_CLASS))
    centers = np.concatenate([g[f"s{i}_obj_center"] for i in range(len(scenes))])
    assert np.array_equal(packed["center"], centers.astype(np.float32))
    names = np.concatenate([g[f"s{i}_obj_color_text"] for i in range(len(scenes))])
    assert packed["color_idx"].tolist() == [packing.color_table()[n] for n in names]
    assert packed["n_pts"].tolist() == np.concatenate([g[f"s{i}_obj_npts"] for i in range(len(scenes))]).tolist()


def test_unpickler_refuses_foreign_classes(tmp_path):
    from text2loc_amd import kitti360pose as K

    p = tmp_path / "evil.pkl"
    pickle.dump(osp.join, open(p, "wb"))  # a global from a module outside the whitelist
    with pytest.raises(pickle.UnpicklingError, match="refusing"):
        K.load_pickle(str(p))


def test_unpickler_is_an_allow_list(tmp_path):
    """A dataset pickle is untrusted input: a reduce through builtins.eval / os.system / getattr must not resolve."""
    from text2loc_amd import kitti360pose as K

    payloads = [b"cbuiltins\neval\n(V__import__('os').getpid()\ntR.", b"cos\nsystem\n(Vtrue\ntR.",
                b"cbuiltins\ngetattr\n(cbuiltins\nlist\nVappend\ntR.", b"cnumpy\nload\n(V/dev/null\ntR.",
                b"cbuiltins\n__import__\n(Vos\ntR."]
    for i, blob in enumerate(payloads):
        f = tmp_path / f"evil{i}.pkl"
        f.write_bytes(blob)
        with pytest.raises(pickle.UnpicklingError):
            K.load_pickle(str(f))
    # what the data needs still loads: arrays, scalars, containers
    f = tmp_path / "ok.pkl"
    f.write_bytes(pickle.dumps({"a": np.arange(6.0).reshape(2, 3), "b": np.float64(2.5), "c": [1, (2, 3)], "d": {4}}))
    ok = K.load_pickle(str(f))
    assert np.array_equal(ok["a"], np.arange(6.0).reshape(2, 3)) and ok["b"] == 2.5 and ok["c"] == [1, (2, 3)]


def _item_arrays(it):
    pose, cell = it["poses"], it["cells"]
    return (np.asarray(pose.pose, dtype=np.float64), np.concatenate([o.xyz for o in cell.objects]).astype(np.float64),
            np.array([d.closest_point for d in pose.descriptions], dtype=np.float64))


def test_training_augmentations_equal_the_reference_draw_for_draw(golden):
    """shuffle_hints / flip_poses (cells.py:80-91, utils.py:15-88): with aug_rng = RandomState(s) every fetched item equals
    what the reference's own Kitti360CoarseDatasetMulti(shuffle_hints=True, flip_poses=True) returned under
    np.random.seed(s) — texts, mirrored pose-in-cell, mirrored object points, mirrored closest points."""
    from text2loc_amd import kitti360pose as K

    g = golden("k360_augment")
    scenes = [str(s) for s in g["scenes"]]
    for s in g["seeds"].tolist():
        ds = K.Kitti360PoseDataset(BASE, scenes, shuffle_hints=True, flip_poses=True, aug_rng=np.random.RandomState(s))
        lo_x = lo_c = 0
        flipped = 0
        for i in range(len(ds)):
            it = ds[i]
            p, x, c = _item_arrays(it)
            assert it["texts"] == str(g[f"seed{s}_texts"][i])
            assert np.array_equal(p, g[f"seed{s}_pose"][i])
            n = int(g[f"seed{s}_xyz_counts"][i])
            assert np.array_equal(x, g[f"seed{s}_xyz"][lo_x:lo_x + n])
            assert np.array_equal(c, g[f"seed{s}_closest"][lo_c:lo_c + len(c)])
            lo_x, lo_c = lo_x + n, lo_c + len(c)
            flipped += int(not np.array_equal(p, np.asarray(ds.all_poses[i].pose, dtype=np.float64)))
        assert flipped > 0  # the fixture really exercises the flips
        # the stored records were copied, not modified (a second epoch starts from the same data)
        plain = K.Kitti360PoseDataset(BASE, scenes)
        for a, b in zip(ds.all_cells, plain.all_cells):
            assert all(np.array_equal(oa.xyz, ob.xyz) for oa, ob in zip(a.objects, b.objects))


def test_flip_pose_in_cell_both_directions(golden):
    from text2loc_amd import kitti360pose as K

    g = golden("k360_augment")
    ds = K.Kitti360PoseDataset(BASE, [str(s) for s in g["scenes"]])
    it = ds[1]
    for name, dirs in (("h", [1]), ("v", [-1]), ("hv", [1, -1])):
        pose, cell, text = it["poses"], it["cells"], it["texts"]
        for d in dirs:
            pose, cell, text = K.flip_pose_in_cell(pose, cell, text, d)
        p, x, c = _item_arrays({"poses": pose, "cells": cell})
        assert text == str(g[f"flip_{name}_text"])
        assert np.array_equal(p, g[f"flip_{name}_pose"]) and np.array_equal(x, g[f"flip_{name}_xyz"])
        assert np.array_equal(c, g[f"flip_{name}_closest"])
    with pytest.raises(ValueError):
        K.flip_pose_in_cell(it["poses"], it["cells"], it["texts"], 0)


def test_flipped_cells_are_repacked_from_the_mirrored_points():
    """The packer caches per-object reductions on the object; a mirrored copy must not inherit the un-mirrored centre."""
    from text2loc_amd import kitti360pose as K
    from text2loc_amd import packing

    ds = K.Kitti360PoseDataset(BASE, ["2013_05_28_drive_0010_sync"])
    it = ds[0]
    kc = packing.class_table(ds.get_known_classes())
    before = packing.pack_cells([it["objects"]], kc)
    pose, cell, _ = K.flip_pose_in_cell(it["poses"], it["cells"], it["texts"], 1)
    after = packing.pack_cells([cell.objects], kc)
    assert np.allclose(after["center"][:, 0], 1 - before["center"][:, 0], atol=1e-6)
    assert np.allclose(after["center"][:, 1:], before["center"][:, 1:], atol=1e-7)


def test_point_transform_follows_no_pc_augment():
    """--no_pc_augment (every published command) = FixedPoints only: raw cell-frame coordinates; without it NormalizeScale
    (+ RandomRotate in training) — evaluation/pipeline.py:215-223, training/coarse.py:182-193."""
    import argparse

    from text2loc_amd import kitti360pose as K
    from text2loc_amd import packing

    A = argparse.Namespace
    assert packing.point_transform_from_args(A(no_pc_augment=True)) == "fixed"
    assert packing.point_transform_from_args(A(no_pc_augment=True), train=True) == "fixed"
    assert packing.point_transform_from_args(A(no_pc_augment=False)) == "normalize"
    assert packing.point_transform_from_args(A(no_pc_augment=False), train=True) == "rotate_normalize"
    assert packing.point_transform_from_args(A(no_pc_augment=False, no_pc_augment_fine=True), fine=True) == "fixed"
    ds = K.Kitti360PoseDataset(BASE, ["2013_05_28_drive_0010_sync"], object_points="sample", seed=1)  # default = published
    it = ds[0]
    pos = it["object_points"]["pos"].reshape(-1, 256, 3)
    for o, p in zip(it["objects"], pos):  # every sampled point IS one of the object's raw points
        raw = np.asarray(o.xyz, dtype=np.float32)
        assert all((raw == q).all(axis=1).any() for q in p[:8])
    dn = K.Kitti360PoseDataset(BASE, ["2013_05_28_drive_0010_sync"], object_points="sample", seed=1, transform="normalize")
    pn = dn[0]["object_points"]["pos"].reshape(-1, 256, 3)
    assert np.abs(pn.mean(axis=1)).max() < 1e-5 and np.all(np.abs(pn).max(axis=(1, 2)) > 0.9999)
    dr = K.Kitti360PoseDataset(BASE, ["2013_05_28_drive_0010_sync"], object_points="sample", seed=1, transform="rotate_normalize")
    a, b = dr[0]["object_points"]["pos"], dr[0]["object_points"]["pos"]
    assert not np.array_equal(a, b)  # a fresh rotation per fetch
    assert dr.get_cell_dataset()._transform == "normalize"  # the cell-only (validation) side never rotates
    # rotation about z keeps z and the xy radius of the un-normalised points
    objs = [it["objects"][:1]]
    r0 = packing.sample_object_points(objs, 256, np.random.default_rng(5), transform="fixed")[0]["pos"]
    rng = np.random.default_rng(5)
    sel = rng.integers(0, len(objs[0][0].xyz), size=256)
    assert np.array_equal(r0, np.asarray(objs[0][0].xyz, dtype=np.float32)[sel])


class _Recorder:
    """a 'model' that records what train_epoch feeds it (the augmentation draws are what is under test)"""

    def __init__(self):
        self.seen = []

    def train(self):
        return self

    def encode_text(self, texts):
        import torch

        return torch.zeros(len(texts), 4, requires_grad=True)

    def encode_objects(self, objects, object_points):
        import torch

        self.seen.append(np.concatenate([np.asarray(p["pos"]).reshape(-1) for p in object_points]))
        return torch.zeros(len(objects), 4)


class _NoOpt:
    def zero_grad(self):
        pass

    def step(self):
        pass


def test_train_epoch_advances_the_augmentation_epoch_with_dataloader_workers():
    """DataLoader workers (non-persistent) restart from a COPY of the dataset every epoch: without train_epoch telling the dataset that an
    epoch has passed, the rotate_normalize draws of epoch 2 repeat epoch 1's exactly (advisor, round 4). Two epochs over two worker
    processes: same items, different draws; and the epoch counter moved."""
    import argparse

    import torch

    from text2loc_amd import kitti360pose as K
    from text2loc_amd.coarse import train_epoch

    ds = K.Kitti360PoseDataset(BASE, ["2013_05_28_drive_0010_sync"], object_points="sample", seed=1, transform="rotate_normalize")
    dl = torch.utils.data.DataLoader(ds, batch_size=4, shuffle=False, num_workers=2, collate_fn=K.Kitti360PoseDataset.collate_fn)
    rec = _Recorder()
    args = argparse.Namespace(ranking_loss="contrastive")
    crit = lambda a, p: (a.sum() + p.sum()) * 0.0  # noqa: E731
    e0 = ds._epoch
    train_epoch(rec, dl, args, _NoOpt(), crit)
    n1 = len(rec.seen)
    train_epoch(rec, dl, args, _NoOpt(), crit)
    assert ds._epoch == e0 + 2 and n1 > 0 and len(rec.seen) == 2 * n1
    for a, b in zip(rec.seen[:n1], rec.seen[n1:]):
        assert a.shape == b.shape and not np.array_equal(a, b)
