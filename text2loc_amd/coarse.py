"""Drop-ins for ``training.coarse.eval_epoch`` (training/coarse.py:63-157) and
``evaluation.pipeline.run_coarse`` / ``evaluation.coarse.run_coarse`` (evaluation/pipeline.py:41-87).

Same arguments, same return structures — the difference is where the work runs:

* cells are encoded by the fused HIP encoder and stay in HBM as the search database (no per-batch
  ``.cpu().numpy()`` round trip, training/coarse.py:105-113);
* the per-query ``float64 matvec + full argsort`` host loop (training/coarse.py:119-125) is ONE fused search on
  the GPU (f32 MFMA scan -> float64 re-rank -> certificate), returning exactly the ids the float64 loop ranks;
* hit@k / close@k / threshold accuracies (training/coarse.py:127-150, evaluation/utils.py:31-54) are
  bookkeeping on [Q,K] integer arrays and run vectorised on the host.

Datasets / dataloaders are duck-typed as in the reference: ``dataloader.dataset`` provides
``get_cell_dataset()`` (-> ``.cells``, items with ``cell_ids, objects, object_points``), ``.all_poses``
(``.pose_w``, ``.cell_id``) and ``.all_cells`` (``.id``, ``.cell_size``, ``.bbox_w``, ``.get_center()``);
batches are dict-of-lists with ``texts`` and ``cell_ids``.
"""
from __future__ import annotations

import time
from typing import Callable, Optional

import numpy as np
import torch


def collate_fn(data):
    """dict-of-lists collate, as Kitti360BaseDataset.collate_fn (dataloading/kitti360pose/base.py:83-87)."""
    return {key: [d[key] for d in data] for key in data[0].keys()}


def train_epoch(model, dataloader, args, optimizer, criterion):
    """training/coarse.py:31-58 with the optimizer and criterion passed in (the reference reads them from module
    globals): text branch on PyTorch autograd, object branch forward/backward in the engine, both meeting in the
    contrastive loss kernel. Returns (mean loss, []) like the reference."""
    if getattr(args, "ranking_loss", "contrastive") == "triplet":
        # the reference's own triplet branch cannot run (it calls encode_objects with one argument, training/coarse.py:49,
        # and eval_epoch asserts it away, :65); there is nothing to mirror
        raise NotImplementedError("ranking_loss='triplet' is not runnable in the reference either (training/coarse.py:47-50)")
    model.train()
    losses = []
    # one more epoch of augmentation draws: the dataset keys its per-item RNG on the epoch (kitti360pose.set_epoch), and DataLoader
    # workers restart from a copy of the dataset every epoch — nobody else would tell them that an epoch has passed
    ds = getattr(dataloader, "dataset", None)
    if hasattr(ds, "set_epoch"):
        ds.set_epoch(int(getattr(ds, "_epoch", 0)) + 1)
    for batch in dataloader:
        optimizer.zero_grad()
        anchor = model.encode_text(batch["texts"])
        positive = model.encode_objects(batch["objects"], batch["object_points"])
        loss = criterion(anchor, positive)  # contrastive (fused kernel) | pairwise | hardest (text2loc_amd.losses)
        loss.backward()
        optimizer.step()
        losses.append(loss.detach())
    return float(torch.stack(losses).mean().item()) if losses else float("nan"), []


def _batches(dataset, batch_size):
    n = len(dataset)
    for lo in range(0, n, batch_size):
        yield collate_fn([dataset[i] for i in range(lo, min(n, lo + batch_size))])


def _is_plain_sequential(dataloader) -> bool:
    """A torch DataLoader that walks its dataset front to back in fixed-size batches (the reference's evaluation loaders:
    shuffle off, no sampler) — the only kind whose batches ``eval_epoch`` may rebuild from the dataset without iterating it."""
    import torch.utils.data as tud

    return (isinstance(dataloader, tud.DataLoader) and isinstance(getattr(dataloader, "sampler", None), tud.SequentialSampler)
            and isinstance(getattr(dataloader, "batch_sampler", None), tud.BatchSampler) and dataloader.batch_size is not None
            and not dataloader.drop_last)


class _StaticTables:
    """What the bookkeeping reads from the duck-typed cell / pose objects, gathered ONCE per dataset: the reference walks
    ``all_cells`` and ``all_poses`` with Python loops on every call (training/coarse.py:127-146, evaluation/pipeline.py:62-83) —
    15k attribute reads + ``get_center`` calls cost more than the GPU spends on encoding and searching the whole database."""

    def __init__(self, cells, poses):
        self.n_cells, self.n_poses = len(cells), len(poses)
        self.cell_ids = np.array([str(c.id) for c in cells], dtype="<U32")
        self.centers_xy = np.array([np.asarray(c.get_center(), dtype=np.float64)[0:2] for c in cells]).reshape(-1, 2)
        self.bbox_xy = np.array([np.asarray(c.bbox_w, dtype=np.float64)[0:2] for c in cells]).reshape(-1, 2)
        self.cell_size = np.array([float(c.cell_size) for c in cells], dtype=np.float64)
        self.cell_scene = np.array([str(c.id).split("_")[0] for c in cells])
        self.pose_xy = np.array([np.asarray(p.pose_w, dtype=np.float64)[0:2] for p in poses]).reshape(-1, 2)
        self.pose_cell_ids = np.array([str(p.cell_id) for p in poses], dtype="<U32")
        self.pose_scene = np.array([str(p.cell_id).split("_")[0] for p in poses])
        row = {cid: i for i, cid in enumerate(self.cell_ids.tolist())}
        self.unique_ids = len(row) == self.n_cells
        self.pose_rows = np.array([row.get(cid, -1) for cid in self.pose_cell_ids.tolist()], dtype=np.int64)  # -1: a cell outside the DB


def _static_tables(dataset) -> "_StaticTables":
    cells, poses = dataset.all_cells, dataset.all_poses
    key = (id(cells), len(cells), id(poses), len(poses))
    hit = getattr(dataset, "_t2l_static_tables", None)
    if hit is not None and hit[0] == key:
        return hit[1]
    t = _StaticTables(cells, poses)
    try:
        dataset._t2l_static_tables = (key, t)
    except Exception:  # (a dataset that refuses new attributes: gathered per call, as before)
        pass
    return t


def _engine_retrieve(model) -> Callable:
    def retrieve(cell_enc: torch.Tensor, text_enc: torch.Tensor, k: int):
        from .engine import MAX_TOPK

        if k > MAX_TOPK:
            raise ValueError(f"max(top_k)={k} exceeds the engine's T2L_MAX_TOPK={MAX_TOPK} (include/t2l.h)")
        eng = model.engine()
        layout = getattr(model.args, "shard_layout", None)
        if layout not in (None, "", "none", "off"):
            # OPT-IN (args.shard_layout = "auto" | "query" | "row"): N ranks, one per GPU, answer the search together — the database
            # replicated and the QUERIES split while it fits a quarter of one GPU's HBM (KITTI360Pose: 11 k rows of ~28 M), row shards
            # + top-k merge beyond; every rank gets the complete result. Collectives run inside: EVERY rank must call eval_epoch, with
            # the same cells and the same queries (AutoSearcher proves that with one small all_reduce and raises on every rank
            # otherwise). Without the option an initialised process group changes nothing: a DP training script that validates on
            # rank 0 only, or feeds each rank its own DistributedSampler slice, searches locally.
            import torch.distributed as dist

            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                from .sharded import AutoSearcher

                srch = AutoSearcher(eng, layout=str(layout))
                srch.set_db(cell_enc.contiguous())
                idx, sc = srch.search(text_enc.contiguous(), k)
                return idx.cpu().numpy().astype(np.int64), sc.cpu().numpy()
        eng.db_set(cell_enc.contiguous())
        idx, sc = eng.search(text_enc.contiguous(), k)
        return idx.cpu().numpy().astype(np.int64), sc.cpu().numpy()

    return retrieve


@torch.no_grad()
def eval_epoch(model, dataloader, args, return_encodings: bool = False, return_distance: bool = False,
               retrieve: Optional[Callable] = None, _extras: Optional[dict] = None):
    """Returns (accuracies{k: float}, accuracies_close{k: float}, top_retrievals{q: ndarray['<U32'][max(top_k)]})
    (+ encodings / dists / scores variants, training/coarse.py:152-157). ``retrieve`` replaces the search step
    (tests of the host bookkeeping pass the oracle here); by default it is the model's HIP engine on this process's GPU.
    Multi-GPU retrieval is opt-in through ``model.args.shard_layout`` ("auto" | "query" | "row"): then every rank of the
    default process group must call this function with the same dataset (see ``_engine_retrieve``)."""
    assert args.ranking_loss != "triplet"  # as the reference (training/coarse.py:65)
    model.eval()
    dataset = dataloader.dataset
    cells_dataset = dataset.get_cell_dataset()
    cells = cells_dataset.cells
    cell_size = cells[0].cell_size
    max_k = int(np.max(args.top_k))
    retrieve = retrieve or _engine_retrieve(model)

    # Engine-sized batching (default; ``args.engine_batching = False`` restores one call per ``args.batch_size`` items). The
    # reference encodes ``args.batch_size`` items per forward call — 1 by default in evaluation (evaluation/args.py:11) — which on
    # this engine is 11,259 launches of a kernel that wants thousands of cells. In eval mode a cell's embedding depends on that
    # cell alone, so the DATABASE side is re-batched freely (bit-identical rows). On the QUERY side the batching is part of the
    # result (padding to the batch's longest sentence, no padding mask): the batches stay, the per-batch Python round does not
    # (``encode_text_batches``).
    rebatch = bool(getattr(args, "engine_batching", True))
    chunk_cells = int(getattr(args, "engine_batch_cells", 4096))
    timing = {}

    # ---- query side (text path stays on PyTorch)
    t0 = time.time()
    text_parts, query_cell_ids, rerun = [], [], []
    le = getattr(model, "language_encoder", None)
    deferred = hasattr(le, "begin_deferred") and hasattr(le, "end_deferred")
    texts = None
    if rebatch and hasattr(dataset, "eval_texts") and hasattr(model, "encode_text_batches") and _is_plain_sequential(dataloader):
        texts = dataset.eval_texts()  # item order == all_poses order: what a sequential DataLoader would have delivered
    if deferred:  # no host sync per batch: the head's overflow flags are collected on the device and read ONCE behind the loop
        le.begin_deferred()
    flagged = []
    try:
        if texts is not None:
            bs = int(dataloader.batch_size)
            text_parts.append(model.encode_text_batches(texts, bs).detach().float())
            rerun.append(lambda: model.encode_text_batches(texts, bs))
            query_cell_ids.extend(p.cell_id for p in dataset.all_poses)
        else:
            for batch in dataloader:
                text_parts.append(model.encode_text(batch["texts"]).detach().float())
                query_cell_ids.extend(batch["cell_ids"])
                if deferred:
                    rerun.append(lambda b=batch["texts"]: model.encode_text(b))
    finally:  # whatever the loop raised, the head must leave deferred mode (or every later call skips its overflow check)
        if deferred:
            n_flags = len(getattr(le, "_deferred", None) or [])
            flagged = le.end_deferred()
    if deferred:
        if flagged and n_flags != len(text_parts):  # (a custom encode_text that calls the head twice, or not at all: the flags no longer index batches)
            raise RuntimeError(f"deferred overflow check: {n_flags} head calls for {len(text_parts)} text batches — cannot tell which batch overflowed")
        for i in flagged:  # a batch whose activations left the f16 range: again, on the PyTorch modules
            keep, le.use_engine_head = le.use_engine_head, False
            try:
                text_parts[i] = rerun[i]().detach().float()
            finally:
                le.use_engine_head = keep
    text_enc = torch.cat(text_parts, dim=0)
    timing["encode_text_s"] = time.time() - t0
    print(f"Encoded {len(text_enc)} query texts in {time.time() - t0:0.2f}.")

    # ---- database side: every cell once, resident on the device
    t0 = time.time()
    if rebatch and hasattr(cells_dataset, "packed") and hasattr(model, "encode_cell_set"):
        t1 = time.time()
        cell_set = cells_dataset.packed()  # flattened once per dataset (cached on it); the reductions once per dataset and GPU
        timing["pack_s"] = time.time() - t1
        cell_enc = model.encode_cell_set(cell_set, points=bool(getattr(cells_dataset, "wants_points", False)),
                                         transform=getattr(cells_dataset, "point_transform", "fixed"),
                                         seed=int(getattr(cells_dataset, "point_seed", 0)), chunk_cells=chunk_cells).float()
        db_cell_ids = list(cell_set.cell_ids)
    else:
        cell_parts, db_cell_ids = [], []
        for batch in _batches(cells_dataset, max(int(args.batch_size), chunk_cells) if rebatch else args.batch_size):
            cell_parts.append(model.encode_objects(batch["objects"], batch["object_points"]).detach().float())
            db_cell_ids.extend(batch["cell_ids"])
        cell_enc = torch.cat(cell_parts, dim=0)
    timing["encode_cells_s"] = time.time() - t0  # (host time of issuing the work: the device is read behind the search)
    assert len(cell_enc) == len(dataset.all_cells)  # training/coarse.py:122
    db_cell_ids = np.array(db_cell_ids, dtype="<U32")
    query_cell_ids = np.array(query_cell_ids, dtype="<U32")

    # ---- retrieval: identical to float64 `cell_encodings @ t` + stable descending argsort, first max(top_k)
    # (a database smaller than max(top_k) yields that many columns, like the reference's `sorted_indices[0:max_k]`;
    # the engine marks missing ranks with id -1, which must never be used as an index)
    k_eff = min(max_k, len(cell_enc))
    t0 = time.time()
    top_idx, top_scores = retrieve(cell_enc, text_enc, k_eff)
    timing["search_s"] = time.time() - t0  # (the first device read of the run: everything enqueued above drains here)
    t0 = time.time()
    if (np.asarray(top_idx) < 0).any():
        raise RuntimeError("retrieval returned an empty rank although k <= number of cells")

    # ---- accuracies (host bookkeeping on [Q,K] arrays; the per-dataset tables are gathered once, _StaticTables)
    tabs = _static_tables(dataset)
    same_db = tabs.unique_ids and len(db_cell_ids) == tabs.n_cells and np.array_equal(db_cell_ids, tabs.cell_ids)
    retrieved_ids = db_cell_ids[top_idx]  # [Q,K]
    if same_db and len(query_cell_ids) == tabs.n_poses and np.array_equal(query_cell_ids, tabs.pose_cell_ids):
        hits = top_idx == tabs.pose_rows[:, None]  # ids are unique: equal ids <=> equal rows (integer compare instead of 41k strings)
    else:
        hits = retrieved_ids == query_cell_ids[:, None]
    if same_db:
        centers, query_poses_w = tabs.centers_xy, tabs.pose_xy
    else:
        centers = np.array([c.get_center()[0:2] for c in cells], dtype=np.float64)
        query_poses_w = np.array([p.pose_w[0:2] for p in dataset.all_poses], dtype=np.float64)
    dists = np.linalg.norm(query_poses_w[:, None, :] - centers[top_idx], axis=2)  # [Q,K]
    accuracies = {k: float(np.mean(hits[:, :k].any(axis=1))) for k in args.top_k}
    accuracies_close = {k: float(np.mean((dists[:, :k] <= cell_size / 2).any(axis=1))) for k in args.top_k}
    top_retrievals = dict(enumerate(retrieved_ids))
    timing["bookkeeping_s"] = time.time() - t0
    eval_epoch.last_timing = timing  # breakdown of the most recent call (bench_e2e.py)
    if _extras is not None:
        _extras.update(top_idx=np.asarray(top_idx), same_db=same_db, tables=tabs, retrieved_ids=retrieved_ids)

    if return_encodings or return_distance:
        ce = cell_enc.cpu().numpy().astype(np.float64)
        te = text_enc.cpu().numpy().astype(np.float64)
        if return_encodings:
            return accuracies, accuracies_close, top_retrievals, ce, te
        return accuracies, accuracies_close, top_retrievals, ce, te, dists, top_scores
    return accuracies, accuracies_close, top_retrievals


def calc_sample_accuracies(pose, top_cells, pos_in_cells, top_k, threshs):
    """evaluation/utils.py:31-54: world-xy error of the predicted position in each retrieved cell; retrievals
    from another scene count as infinitely far."""
    assert len(top_cells) == max(top_k) == len(pos_in_cells)
    pred_w = np.array([c.bbox_w[0:2] + pos_in_cells[i, :] * c.cell_size for i, c in enumerate(top_cells)])
    dists = np.linalg.norm(np.asarray(pose.pose_w)[0:2] - pred_w, axis=1)
    scene = pose.cell_id.split("_")[0]
    dists[np.array([c.id.split("_")[0] for c in top_cells]) != scene] = np.inf
    return {k: {t: bool(np.min(dists[0:k]) <= t) for t in threshs} for k in top_k}


def sample_accuracies_batch(pose_xy, pose_scene, cell_bbox_xy, cell_size, cell_scene, pos_in_cells, top_k, threshs):
    """``calc_sample_accuracies`` for all poses at once. pose_xy f64[Q,2], pose_scene / cell_scene str arrays [Q] / [Q,K],
    cell_bbox_xy f64[Q,K,2], cell_size f64[Q,K], pos_in_cells f64[Q,K,2] -> {k: {t: bool[Q]}}."""
    pred_w = cell_bbox_xy + pos_in_cells * cell_size[..., None]
    dists = np.linalg.norm(pose_xy[:, None, :] - pred_w, axis=2)
    dists = np.where(cell_scene != pose_scene[:, None], np.inf, dists)
    return {k: {t: dists[:, :k].min(axis=1) <= t for t in threshs} for k in top_k}


def _pose_cell_tables(poses, cells, retrievals):
    """The arrays ``sample_accuracies_batch`` wants, gathered once per run from the duck-typed pose / cell objects."""
    row = {str(c.id): i for i, c in enumerate(cells)}
    bbox = np.array([np.asarray(c.bbox_w, dtype=np.float64)[0:2] for c in cells])
    size = np.array([float(c.cell_size) for c in cells])
    scene = np.array([str(c.id).split("_")[0] for c in cells])
    ridx = np.array([[row[str(cid)] for cid in r] for r in retrievals], dtype=np.int64)
    pose_xy = np.array([np.asarray(p.pose_w, dtype=np.float64)[0:2] for p in poses])
    pose_scene = np.array([str(p.cell_id).split("_")[0] for p in poses])
    return pose_xy, pose_scene, bbox[ridx], size[ridx], scene[ridx]


@torch.no_grad()
def run_coarse(model, dataloader, args, retrieve: Optional[Callable] = None):
    """Returns (retrievals: List[ndarray of cell ids], accuracies{k:{t: float}}) — the result contract of
    evaluation/pipeline.py:41-87, computed for all poses at once on [Q,K] arrays instead of per pose."""
    extras: dict = {}
    acc, acc_close, top = eval_epoch(model, dataloader, args, retrieve=retrieve, _extras=extras)
    print("Retrieval Accs:")
    print(acc)
    print("Retrieval Accs Close:")
    print(acc_close)
    ds = dataloader.dataset
    retrievals = list(extras["retrieved_ids"])
    assert len(retrievals) == len(ds.all_poses)
    if extras["same_db"]:  # the retrieved ROWS are at hand: no id -> row dictionary walk over [Q,K] strings
        t, ridx = extras["tables"], extras["top_idx"]
        pose_xy, pose_scene, bbox_xy, size, scene = t.pose_xy, t.pose_scene, t.bbox_xy[ridx], t.cell_size[ridx], t.cell_scene[ridx]
    else:
        pose_xy, pose_scene, bbox_xy, size, scene = _pose_cell_tables(ds.all_poses, ds.all_cells, retrievals)
    centre = np.full(bbox_xy.shape, 0.5)  # the coarse-only estimate is the cell centre (pipeline.py:72)
    ok = sample_accuracies_batch(pose_xy, pose_scene, bbox_xy, size, scene, centre, args.top_k, args.threshs)
    return retrievals, {k: {t: float(np.mean(ok[k][t])) for t in args.threshs} for k in args.top_k}
