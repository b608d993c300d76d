"""Throughput AT THE BOUNDARY: ``run_coarse(model, dataloader, args)`` — the reference's evaluation entry point
(evaluation/pipeline.py:41-87 -> training/coarse.py:63-157) — through the drop-in Python surface at BASELINE config 2's size:
a KITTI360Pose-shaped dataset of 11,259 cells / 4,096 poses (``synth.make_k360_records`` -> ``Kitti360PoseDataset``), the text
branch behind the sentence cache ("T5 embeddings precomputed"), at ``batch_size`` 1 (the reference's evaluation default,
evaluation/args.py:11) and 64, in the embedding feature mode and in the PUBLISHED mode (point batches -> PointNet++ in the
engine). Wall time with a breakdown, the GPU time of the stages beside it (``t2l_kernel_stats`` events), the legacy one-call-per-
batch loop (``args.engine_batching = False``) for the before/after, and the oracle's reference-style CPU path on a subsample.

``bench.py`` calls ``measure()`` for its SECONDARY line; ``python bench_e2e.py`` prints the full record.
"""
from __future__ import annotations

import contextlib
import io
import json
import sys
import time

import numpy as np
import torch

from text2loc_amd import synth


def _run(model, dl, args, reps):
    from text2loc_amd.coarse import eval_epoch, run_coarse

    best, timing, res = None, None, None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            res = run_coarse(model, dl, args)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, timing = dt, dict(eval_epoch.last_timing)
    return best, timing, res


def _stage_gpu_ms(model, dl, args):
    """GPU time of the engine stages of ONE warm run_coarse (event pairs around each engine call: t2l_kernel_stats)."""
    names = ("reduce_objects", "pointnet", "encode_cells", "text_head", "text_inter", "search_scan", "search_rerank", "search_small",
             "search_exact")
    engines = [model.engine()]
    he = getattr(model.language_encoder, "_th_engine", None)
    if he is not None and he is not engines[0]:
        engines.append(he)
    for e in engines:
        e.set_option("profile_events", 1)
        for n in names:
            e.kernel_stats(n)
    _run(model, dl, args, 1)
    out = {}
    for e in engines:
        for n in names:
            ms, cnt = e.kernel_stats(n)
            if cnt:
                out[n] = out.get(n, 0.0) + ms * cnt
        e.set_option("profile_events", 0)
    return out


def _cpu_reference_style(ds, model, args, published, n_cells_sample):
    """The oracle's restatement of the reference's CPU path on a subsample: per-object reductions + (published mode) PointNet++ +
    ObjectEncoder + set transformer per cell (numpy f32), seconds per cell; the retrieval loop is bench.py's cpu_baseline."""
    from oracle import t2l_oracle as O

    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    cells = ds.all_cells[:n_cells_sample]
    t0 = time.perf_counter()
    counts = np.array([len(c.objects) for c in cells], np.int32)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    flat = [o for c in cells for o in c.objects]
    packed = {"counts": counts, "offsets": offs, "class_idx": np.array([synth.KNOWN_CLASS.index(o.label) + 1 for o in flat], np.int32),
              "rgb": np.array([np.mean(o.rgb, axis=0) for o in flat], np.float32),
              "center": np.array([np.mean(o.xyz, axis=0) for o in flat], np.float32),
              "n_pts": np.array([len(o.xyz) for o in flat], np.float32)}
    packed["color_idx"] = synth.color_name_to_embed_index(synth.nearest_color_index(packed["rgb"])).astype(np.int32)
    if published:
        from oracle import t2l_oracle_pointnet as OP

        po = np.concatenate([[0], np.cumsum([len(o.xyz) for o in flat])]).astype(np.int64)
        xyz = np.concatenate([o.xyz for o in flat]).astype(np.float32)
        rgb = np.concatenate([o.rgb for o in flat]).astype(np.float32)
        pos, col = OP.sample_object_points(xyz, rgb, po, 0)
        packed["pn_feat"] = OP.pointnet_features(pos, col, offs, sd).astype(np.float32)
    O.encode_cells(packed, sd, not published, not published)
    return (time.perf_counter() - t0) / max(1, len(cells))


def measure(n_cells: int = 11259, n_poses: int = 4096, quick: bool = False, modes=("embed", "published")):
    from text2loc_amd.kitti360pose import Kitti360PoseDataset
    from text2loc_amd.text_cache import TextCache

    t0 = time.perf_counter()
    cells, poses = synth.make_k360_records(n_cells, n_poses, seed=0)
    n_objects = sum(len(c.objects) for c in cells)
    out = {"workload": f"run_coarse over a synthetic KITTI360Pose-shaped dataset: {n_cells} cells ({n_objects} objects, 24-64 raw points "
                       f"each; KITTI360Pose's mean is 1,827), {n_poses} poses x 6 hints, text branch behind the sentence cache "
                       "(T5 hidden states precomputed: synthetic), top_k [1,3,5,10]",
           "n_cells": n_cells, "n_poses": n_poses, "n_objects": n_objects, "dataset_generation_s": time.perf_counter() - t0}
    for mode in modes:
        published = mode == "published"
        ds = Kitti360PoseDataset.from_records(cells, poses, object_points="sample" if published else None, seed=0)
        args = synth.coarse_args(class_embed=not published, color_embed=not published)
        model = synth.make_coarse_model(args, sentences=TextCache.sentences_of(ds), seed=0)
        rec = {"feature_mode": "PointNet++ features2 (published: class_embed/color_embed off)" if published else "class/colour embeddings"}
        ids_ref = None
        for bs in (1, 64):
            args.batch_size = bs
            args.engine_batching = True
            dl = torch.utils.data.DataLoader(ds, batch_size=bs, collate_fn=ds.collate_fn, shuffle=False)
            if bs == 1:
                ds._packed_holder.clear()  # cold: the one-time flatten + upload + reductions of the dataset are inside
                model.language_encoder.text_cache._desc.clear()
                cold, tc, _ = _run(model, dl, args, 1)
                rec["first_call_s"] = cold
                rec["first_call_breakdown_s"] = tc
            # (best of a few warm runs: the embedding mode is 10 ms of mostly host time — one noisy run on a busy host tripled it once)
            warm, tw, res = _run(model, dl, args, (2 if quick else 3) if published else 7)
            ids = np.array(res[0])
            if ids_ref is None:
                ids_ref = ids
            rec[f"batch_size_{bs}"] = {"wall_s": warm, "breakdown_s": tw, "queries_per_s_end_to_end": n_poses / warm,
                                       "cells_per_s_end_to_end": n_cells / warm}
            if bs == 1:
                st = _stage_gpu_ms(model, dl, args)
                gpu_s = sum(st.values()) * 1e-3
                rec["gpu_stage_ms"] = st
                rec["gpu_stages_total_s"] = gpu_s
                rec["wall_over_gpu_stages"] = warm / gpu_s if gpu_s > 0 else None
        # before: the reference's loop shape, one encode call per args.batch_size items (round 5's eval_epoch)
        legacy = {}
        for bs in (64, 1):
            if published:
                nc, npz = (1024, 512) if not quick else (256, 128)
                sub = Kitti360PoseDataset.from_records(cells[:nc], [p for p in poses if int(p.cell_id.split("_")[1]) < nc][:npz],
                                                       object_points="sample", seed=0)
            else:
                nc, sub = n_cells, ds
                if quick and bs == 1:
                    nc = 1024
                    sub = Kitti360PoseDataset.from_records(cells[:nc], [p for p in poses if int(p.cell_id.split("_")[1]) < nc][:512], seed=0)
            args.batch_size = bs
            args.engine_batching = False
            dl = torch.utils.data.DataLoader(sub, batch_size=bs, collate_fn=sub.collate_fn, shuffle=False)
            w, tl, _ = _run(model, dl, args, 1)
            legacy[f"batch_size_{bs}"] = {"cells": len(sub.all_cells), "poses": len(sub.all_poses), "wall_s": w, "breakdown_s": tl,
                                          "us_per_cell_db_side": tl["encode_cells_s"] / len(sub.all_cells) * 1e6,
                                          "us_per_query_text_side": tl["encode_text_s"] / max(1, len(sub.all_poses)) * 1e6,
                                          "extrapolated_full_size_s": tl["encode_cells_s"] / len(sub.all_cells) * n_cells
                                          + tl["encode_text_s"] / max(1, len(sub.all_poses)) * n_poses}
        args.engine_batching = True
        rec["legacy_one_call_per_batch"] = legacy
        try:
            s_cell = _cpu_reference_style(ds, model, args, published, 2 if published else 64)
            rec["cpu_reference_style_encode"] = {"s_per_cell": s_cell, "cells_sampled": 2 if published else 64,
                                                 "extrapolated_db_side_s": s_cell * n_cells,
                                                 "what": "oracle (numpy f32 restatement of object_encoder.py:66-153 + cell_retrieval.py:65-110"
                                                         + (" + pointnet2.py:18-100" if published else "") + "), one thread"}
        except Exception as e:  # the oracle is the checker: its absence must not take the measurement down
            rec["cpu_reference_style_encode"] = {"error": repr(e)}
        out[mode] = rec
        del model
        torch.cuda.empty_cache()
    return out


def headline_brief(rec: dict) -> dict:
    """The few scalars of ``measure()``'s record that go into bench.py's one JSON line."""
    b = {}
    for mode in ("embed", "published"):
        r = rec.get(mode)
        if not r:
            continue
        b[mode] = {"bs1_wall_s": r["batch_size_1"]["wall_s"], "bs64_wall_s": r["batch_size_64"]["wall_s"],
                   "first_call_s": r.get("first_call_s"), "gpu_stages_s": r.get("gpu_stages_total_s"),
                   "wall_over_gpu_stages": r.get("wall_over_gpu_stages"),
                   "legacy_bs1_full_size_s": r["legacy_one_call_per_batch"]["batch_size_1"]["extrapolated_full_size_s"]}
    return b


if __name__ == "__main__":
    quick = "--quick" in sys.argv
    r = measure(quick=quick)
    print(json.dumps(r, indent=1, default=float))
    print("E2E " + json.dumps(headline_brief(r), default=float))
