"""Default arithmetic, decided with data (round-5 verdict item 5): does the plain-f16 option (``encoder_f16``: ONE f16 MFMA product per
operand pair, embeddings within ~1e-4) change what a user of the reference SEES — the retrieved cell ids and the top-1/3/5/10
recall — against the default split-f16 form (three products per pair, embeddings within ~2e-7 of the f32 modules)?

The workload is BASELINE config 2's shape with a TRAINED model (an untrained encoder's rows are nearly parallel and would overstate
id disagreement): a synthetic KITTI360Pose-shaped dataset (``synth.make_k360_records``), the coarse model trained on it by
``train_epoch`` (the reference's own training script shape: training/coarse.py:31-58, README.md:87-99 — batch 64, contrastive
loss, temperature 0.1, Adam) for a few epochs, then the validation poses through ``eval_epoch`` three ways:

    exact      database and queries in the default arithmetic
    db_f16     database (PointNet++ -> cell encoder) under ``encoder_f16``, queries in the default arithmetic
    all_f16    database and the text head under ``encoder_f16``

Reported: recall@k of each, the fraction of queries whose top-k id LIST / id SET equals the exact run's, the largest embedding
difference. The search itself is float64-exact in every run (``t2l_search``'s contract), so every difference is the encoders'.
``python tools/arith_ab.py [--published] [--cells N] [--epochs E]``; ``bench.py`` calls ``measure()`` for its secondary record.
"""
from __future__ import annotations

import argparse
import contextlib
import io
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from text2loc_amd import synth  # noqa: E402


def _engines(model):
    engs = [model.engine()]
    he = model.language_encoder._head_engine(model.device) if hasattr(model.language_encoder, "_head_engine") else None
    return engs, he


def _eval(model, dl, args, db_f16: bool, text_f16: bool):
    from text2loc_amd.coarse import eval_epoch

    engs, he = _engines(model)
    for e in engs:
        e.set_option("encoder_f16", 1 if db_f16 else 0)
    if he is not None:
        he.set_option("encoder_f16", 1 if text_f16 else 0)
    tc = model.language_encoder.text_cache
    if tc is not None and hasattr(tc, "_vec"):  # the eval-mode memo of per-sentence vectors is keyed on the weights, not on the option
        tc._vec.clear()
    model.language_encoder._head_generation += 1
    extras = {}
    with contextlib.redirect_stdout(io.StringIO()):
        acc, acc_close, _, ce, te = eval_epoch(model, dl, args, return_encodings=True, _extras=extras)
    for e in engs:
        e.set_option("encoder_f16", 0)
    if he is not None:
        he.set_option("encoder_f16", 0)
    return {"acc": acc, "acc_close": acc_close, "ids": extras["top_idx"].copy(), "cells": ce, "texts": te}


def _agreement(a, b, ks):
    out = {}
    for k in ks:
        out[f"same_list_top{k}"] = float(np.mean((a[:, :k] == b[:, :k]).all(axis=1)))
        out[f"same_set_top{k}"] = float(np.mean([set(x[:k]) == set(y[:k]) for x, y in zip(a, b)]))
    return out


def measure(published: bool = False, n_cells: int = 11259, n_train: int = 8192, n_eval: int = 4096, epochs: int = 6, lr: float = 5e-4,
            seed: int = 0, log=None):
    from text2loc_amd.coarse import train_epoch
    from text2loc_amd.kitti360pose import Kitti360PoseDataset
    from text2loc_amd.losses import ContrastiveLoss
    from text2loc_amd.optim import Adam
    from text2loc_amd.text_cache import TextCache

    t0 = time.perf_counter()
    cells, poses = synth.make_k360_records(n_cells, n_train + n_eval, seed=seed)
    pts = "sample" if published else None
    ds_all = Kitti360PoseDataset.from_records(cells, poses, object_points=pts, seed=seed)
    ds_train = Kitti360PoseDataset.from_records(cells, poses[:n_train], object_points=pts, seed=seed)
    ds_eval = Kitti360PoseDataset.from_records(cells, poses[n_train:], object_points=pts, seed=seed)
    args = synth.coarse_args(class_embed=not published, color_embed=not published, batch_size=64, pointnet_freeze=False)
    model = synth.make_coarse_model(args, sentences=TextCache.sentences_of(ds_all), seed=seed)
    le = model.language_encoder
    le.cache_in_training = True  # --fixed_embedding: the frozen T5's hidden states come from the sentence cache in training too
    dl_train = torch.utils.data.DataLoader(ds_train, batch_size=64, collate_fn=ds_train.collate_fn, shuffle=True, drop_last=True,
                                           generator=torch.Generator().manual_seed(seed))
    dl_eval = torch.utils.data.DataLoader(ds_eval, batch_size=64, collate_fn=ds_eval.collate_fn, shuffle=False)
    opt = Adam(model, lr=lr)
    crit = ContrastiveLoss(0.1)
    torch.manual_seed(seed)
    losses = []
    for ep in range(epochs):
        t1 = time.perf_counter()
        loss, _ = train_epoch(model, dl_train, args, opt, crit)
        losses.append(loss)
        if log:
            log(f"epoch {ep}: loss {loss:.4f} ({time.perf_counter() - t1:.1f} s)")
    model.eval()
    t_train = time.perf_counter() - t0
    runs = {"exact": _eval(model, dl_eval, args, False, False), "db_f16": _eval(model, dl_eval, args, True, False),
            "all_f16": _eval(model, dl_eval, args, True, True)}
    again = _eval(model, dl_eval, args, False, False)  # the A/B harness itself: the exact run repeats bit for bit
    ks = [1, 3, 5, 10]
    ex = runs["exact"]
    out = {"workload": f"{'published (PointNet++)' if published else 'embedding'} feature mode, {n_cells} cells, {n_train} training poses x "
                       f"{epochs} epochs (batch 64, contrastive, T = 0.1, Adam lr {lr}), {n_eval} validation poses, top-10 float64-exact search",
           "train_losses": [round(float(x), 4) for x in losses], "train_s": t_train,
           "exact_run_repeats_bit_for_bit": bool(np.array_equal(ex["ids"], again["ids"]) and np.array_equal(ex["cells"], again["cells"])),
           "recall": {name: {f"top{k}": r["acc"][k] for k in ks} for name, r in runs.items()},
           "recall_close": {name: {f"top{k}": r["acc_close"][k] for k in ks} for name, r in runs.items()}}
    for name in ("db_f16", "all_f16"):
        r = runs[name]
        d = {"max_abs_cell_embedding_diff": float(np.abs(r["cells"] - ex["cells"]).max()),
             "max_abs_text_embedding_diff": float(np.abs(r["texts"] - ex["texts"]).max())}
        d.update(_agreement(ex["ids"], r["ids"], ks))
        d["recall_delta"] = {f"top{k}": r["acc"][k] - ex["acc"][k] for k in ks}
        # how close were the calls that flipped? the exact run's score gap at the first differing rank
        sc = np.einsum("qd,qkd->qk", ex["texts"], ex["cells"][ex["ids"]])
        flips = np.nonzero((r["ids"] != ex["ids"]).any(axis=1))[0]
        if len(flips):
            first = np.array([int(np.argmax(r["ids"][q] != ex["ids"][q])) for q in flips])
            gaps = np.array([sc[q, j] - sc[q, min(j + 1, sc.shape[1] - 1)] for q, j in zip(flips, first)])
            d["flipped_queries"] = int(len(flips))
            d["exact_score_gap_at_first_flip"] = {"median": float(np.median(gaps)), "max": float(gaps.max())}
        out[name + "_vs_exact"] = d
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--published", action="store_true")
    ap.add_argument("--cells", type=int, default=11259)
    ap.add_argument("--train-poses", type=int, default=8192)
    ap.add_argument("--eval-poses", type=int, default=4096)
    ap.add_argument("--epochs", type=int, default=6)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    rec = measure(a.published, a.cells, a.train_poses, a.eval_poses, a.epochs, log=lambda s: print(s, file=sys.stderr, flush=True))
    txt = json.dumps(rec, indent=1, default=float)
    print(txt)
    if a.out:
        with open(a.out, "w") as f:
            f.write(txt)
