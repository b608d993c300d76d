"""GPU tier: the drop-in surface end to end — CellRetrievalNetwork.encode_objects (packer -> fused HIP encoder),
eval_epoch / run_coarse (HIP search) — against the goldens of the reference's own
evaluation.pipeline.run_coarse run on the same 64 cells x 64 queries."""
import argparse

import numpy as np
import pytest
import torch

from text2loc_amd import synth
from tests.test_host_logic import StubCell, StubPose, make_objects

pytestmark = pytest.mark.gpu


class PresetText(torch.nn.Module):
    """Text branch stand-in: returns precomputed (golden) text embeddings; 'descriptions' are row indices."""

    def __init__(self, table):
        super().__init__()
        self.table = torch.nn.Parameter(torch.from_numpy(table), requires_grad=False)

    def forward(self, idx):
        return self.table[torch.as_tensor(idx, device=self.table.device)]

    @property
    def device(self):
        return self.table.device


def _args(**kw):
    a = argparse.Namespace(coarse_embed_dim=256, object_size=28, object_inter_module_num_heads=4,
                           object_inter_module_num_layers=2, hungging_model=None, fixed_embedding=True,
                           intra_module_num_layers=1, intra_module_num_heads=4, inter_module_num_layers=1,
                           inter_module_num_heads=4, class_embed=True, color_embed=True,
                           use_features=["class", "color", "position", "num"], ranking_loss="contrastive",
                           top_k=[1, 3, 5], threshs=[5, 10, 15], batch_size=16)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def test_run_coarse_matches_the_reference_run(golden):
    from text2loc_amd.cell_retrieval import CellRetrievalNetwork
    from text2loc_amd.coarse import collate_fn, eval_epoch, run_coarse

    g = golden("retrieval_e2e")
    args = _args(top_k=[int(k) for k in g["top_k"]], threshs=[int(t) for t in g["threshs"]])
    cells_np = synth.make_cells(int(g["n_cells"]), seed=int(g["cell_seed"]))
    objects = make_objects(cells_np, int(g["cell_seed"]))
    model = CellRetrievalNetwork(synth.KNOWN_CLASS, synth.COLOR_NAMES, args,
                                 language_encoder=PresetText(g["text_encodings"]))
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_object_branch_weights(int(g["weight_seed"])).items()}
    model.load_state_dict(sd, strict=False)
    model = model.to("cuda").eval()

    cells = [StubCell(c, b, g["cell_size"]) for c, b in zip(g["db_cell_ids"], g["cell_bbox_w"])]
    poses = [StubPose(c, p) for c, p in zip(g["query_cell_ids"], g["query_pose_w"])]

    class CellDs:
        def __init__(self):
            self.cells = cells

        def __len__(self):
            return len(cells)

        def __getitem__(self, i):
            return {"cells": cells[i], "cell_ids": cells[i].id, "objects": objects[i], "object_points": None}

    class Ds:
        all_cells, all_poses = cells, poses

        def __len__(self):
            return len(poses)

        def __getitem__(self, i):
            return {"texts": i, "cell_ids": poses[i].cell_id}

        def get_cell_dataset(self):
            return CellDs()

    # the HOST packer (numpy, the reference's own arithmetic for the per-object means: imports.py:28-41) feeds the encoder,
    # so the only difference to the reference run is the fused encoder's float32 round-off
    from oracle import c_oracle
    from text2loc_amd import packing

    for objs in objects:
        for o in objs:
            packing.object_features(o)
    dl = torch.utils.data.DataLoader(Ds(), batch_size=16, collate_fn=collate_fn, shuffle=False)
    acc, close, retr, ce, te = eval_epoch(model, dl, args, return_encodings=True)
    ref_ce = g["cell_encodings"].astype(np.float64)
    assert np.abs(ce - ref_ce).max() < 2e-6
    assert np.abs(te - g["text_encodings"].astype(np.float64)).max() < 1e-6  # encode_text re-normalises the preset rows
    ids = g["db_cell_ids"]
    k = max(args.top_k)
    # (1) EVERY query: the retrieved ids are exactly the float64 ranking of the embeddings the engine produced
    ridx, _ = c_oracle.retrieve_topk(ce.astype(np.float32), te.astype(np.float32), k)
    for q in range(len(retr)):
        assert np.array_equal(retr[q], ids[ridx[q]])
    # (2) and they are the REFERENCE's ids for every query whose reference ranking the encoder round-off cannot touch:
    # a score moves by at most ||c - c_ref||_2 * ||t||_2, so two neighbours can swap only if their gap is below twice that
    delta = np.linalg.norm(ce - ref_ce, axis=1).max() * np.linalg.norm(te, axis=1).max()
    full = np.sort(ref_ce @ te.T, axis=0)[::-1]
    decided = np.abs(np.diff(full[: k + 1], axis=0)).min(axis=0) > 2 * delta
    assert decided.sum() >= 60, (decided.sum(), delta)
    for q in np.nonzero(decided)[0]:
        assert np.array_equal(retr[q], ids[g["top_rows"][q]])
    undecided = int((~decided).sum())
    assert np.abs(np.array([acc[kk] for kk in args.top_k]) - g["acc"]).max() <= undecided / 64 + 1e-12
    retrievals, at = run_coarse(model, dl, args)
    got = np.array([[at[kk][t] for t in args.threshs] for kk in args.top_k])
    assert np.abs(got - g["acc_thresh"]).max() <= undecided / 64 + 1e-12
    if undecided == 0 or all(np.array_equal(retr[q], ids[g["top_rows"][q]]) for q in range(64)):
        assert np.array_equal(np.array([acc[kk] for kk in args.top_k]), g["acc"]) and np.array_equal(got, g["acc_thresh"])
    assert len(retrievals) == 64 and retrievals[0].dtype.kind == "U"


def test_weights_resync_after_parameter_update():
    from text2loc_amd.cell_retrieval import CellRetrievalNetwork

    args = _args()
    model = CellRetrievalNetwork(synth.KNOWN_CLASS, synth.COLOR_NAMES, args,
                                 language_encoder=PresetText(np.zeros((1, 256), np.float32)))
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in
                           synth.make_object_branch_weights(0).items()}, strict=False)
    model = model.to("cuda").eval()
    objs = make_objects(synth.make_cells(3, seed=5), 5)
    a = model.encode_objects(objs, [None] * 3).cpu().numpy()
    with torch.no_grad():
        model.obj_inter_module[0].linear1.bias.add_(0.5)
    b = model.encode_objects(objs, [None] * 3).cpu().numpy()
    assert np.abs(a - b).max() > 1e-4  # the engine picked up the in-place update
    model.train()  # training mode is served by the engine's batch-statistics forward (tests/test_gpu_train_loop.py)
    c = model.encode_objects(objs, [None] * 3)
    assert c.requires_grad and c.shape == (3, 256)


def test_cell_database_build_save_load_search(golden, tmp_path):
    """f-2: encode once -> file -> HBM; retrieval through the persisted database equals eval_epoch's."""
    from text2loc_amd.cell_retrieval import CellRetrievalNetwork
    from text2loc_amd.db import CellDatabase

    g = golden("retrieval_e2e")
    args = _args(top_k=[int(k) for k in g["top_k"]])
    cells_np = synth.make_cells(int(g["n_cells"]), seed=int(g["cell_seed"]))
    objects = make_objects(cells_np, int(g["cell_seed"]))
    model = CellRetrievalNetwork(synth.KNOWN_CLASS, synth.COLOR_NAMES, args, language_encoder=PresetText(g["text_encodings"]))
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in
                           synth.make_object_branch_weights(int(g["weight_seed"])).items()}, strict=False)
    model = model.to("cuda").eval()
    cells = [StubCell(c, b, g["cell_size"]) for c, b in zip(g["db_cell_ids"], g["cell_bbox_w"])]

    class CellDs:
        def __len__(self):
            return len(cells)

        def __getitem__(self, i):
            return {"cells": cells[i], "cell_ids": cells[i].id, "objects": objects[i], "object_points": None}

    db = CellDatabase.build(model, CellDs(), batch_size=20)
    assert len(db) == 64 and db.embeddings.shape == (64, 256) and db.bbox_w.shape == (64, 6)
    assert np.abs(db.embeddings - g["cell_encodings"]).max() < 1e-4
    path = str(tmp_path / "cells.t2ldb.npz")
    db.save(path)
    db2 = CellDatabase.load(path)
    assert np.array_equal(db2.cell_ids, db.cell_ids) and np.array_equal(db2.embeddings, db.embeddings)
    assert db2.meta["class_embed"] is True and db2.fine_desc is None
    from oracle import t2l_oracle as O

    t = torch.from_numpy(g["text_encodings"]).cuda()
    idx, sc = db2.search(model.engine(), t, 5)
    ridx, rsc = O.retrieve_topk(db2.embeddings, g["text_encodings"], 5)
    assert np.array_equal(idx.cpu().numpy().astype(np.int64), ridx) and np.abs(sc.cpu().numpy() - rsc).max() < 1e-12
    with pytest.raises(ValueError, match="unique"):
        CellDatabase(["a", "a"], np.zeros((2, 256), np.float32))
