"""Persistent cell database: what the reference rebuilds from pickled ``Cell`` objects on every
evaluation run — encode every cell of the dataset (training/coarse.py:99-113), keep the encodings as a host array — as a
build-once artefact the engine loads straight into HBM.

    db = CellDatabase.build(coarse_model, cell_dataset, fine_model=None, batch_size=2048)
    db.save("k360_val.t2ldb.npz");  db = CellDatabase.load("k360_val.t2ldb.npz")
    idx, score = db.search(engine, text_embeddings, k=10)        # row ids -> db.cell_ids[idx]

File format (``numpy.savez``; little-endian, row-major): ``cell_ids <U[N]``, ``embeddings f32[N,256]`` (unit rows, the
coarse encodings), optional ``fine_desc f32[N,16,128]`` (the query-independent half of the fine stage), ``bbox_w f64[N,6]``,
``cell_size f64[N]``, ``meta`` (format version, feature mode). 1 KiB per cell without, 9 KiB with the fine descriptors.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

FORMAT_VERSION = 1


class CellDatabase:
    def __init__(self, cell_ids, embeddings, bbox_w=None, cell_size=None, fine_desc=None, meta: Optional[dict] = None):
        self.cell_ids = np.asarray(cell_ids)
        self.embeddings = np.ascontiguousarray(embeddings, dtype=np.float32)
        self.bbox_w = None if bbox_w is None else np.asarray(bbox_w, dtype=np.float64)
        self.cell_size = None if cell_size is None else np.asarray(cell_size, dtype=np.float64)
        self.fine_desc = None if fine_desc is None else np.ascontiguousarray(fine_desc, dtype=np.float32)
        self.meta = dict(meta or {})
        if self.embeddings.ndim != 2 or self.embeddings.shape[0] != len(self.cell_ids):
            raise ValueError("embeddings must be [n_cells, D], one row per cell id")
        if len(set(self.cell_ids.tolist())) != len(self.cell_ids):
            raise ValueError("cell ids are not unique")  # dataloading/kitti360pose/eval.py:139

    def __len__(self):
        return len(self.cell_ids)

    # ---- build ------------------------------------------------------------------------------------------------
    @classmethod
    @torch.no_grad()
    def build(cls, model, cell_dataset, fine_model=None, batch_size: int = 2048, object_points_fn=None):
        """``cell_dataset``: ``dataset.get_cell_dataset()`` of the reference (items with cells / cell_ids / objects /
        object_points). ``object_points_fn(objects) -> object_points`` overrides the items' own point batches."""
        from .cross_matcher import pad_objects

        model.eval()
        ids, embs, fines, bboxes, sizes = [], [], [], [], []
        n = len(cell_dataset)
        for lo in range(0, n, batch_size):
            items = [cell_dataset[i] for i in range(lo, min(n, lo + batch_size))]
            objects = [it["objects"] for it in items]
            pts = object_points_fn(objects) if object_points_fn is not None else [it.get("object_points") for it in items]
            embs.append(model.encode_objects(objects, pts).cpu().numpy())
            ids.extend(str(it["cell_ids"]) for it in items)
            for it in items:
                c = it.get("cells")
                bboxes.append(np.asarray(getattr(c, "bbox_w", np.zeros(6)), dtype=np.float64))
                sizes.append(float(getattr(c, "cell_size", 0.0)))
            if fine_model is not None:
                padded = [pad_objects(o) for o in objects]
                fpts = object_points_fn(padded) if object_points_fn is not None else None
                fines.append(fine_model.encode_cells(padded, fpts).cpu().numpy())
        a = model.args
        meta = {"format": FORMAT_VERSION, "class_embed": bool(getattr(a, "class_embed", False)),
                "color_embed": bool(getattr(a, "color_embed", False)), "use_features": list(a.use_features)}
        return cls(np.array(ids), np.concatenate(embs) if embs else np.zeros((0, 256), np.float32), np.array(bboxes),
                   np.array(sizes), np.concatenate(fines) if fines else None, meta)

    # ---- persistence ------------------------------------------------------------------------------------------
    def save(self, path: str):
        arrays = {"cell_ids": self.cell_ids.astype(str), "embeddings": self.embeddings,
                  "meta": np.array(repr(dict(self.meta, format=FORMAT_VERSION)))}
        if self.bbox_w is not None:
            arrays["bbox_w"] = self.bbox_w
        if self.cell_size is not None:
            arrays["cell_size"] = self.cell_size
        if self.fine_desc is not None:
            arrays["fine_desc"] = self.fine_desc
        np.savez(path, **arrays)

    @classmethod
    def load(cls, path: str):
        import ast

        with np.load(path, allow_pickle=False) as f:
            meta = ast.literal_eval(str(f["meta"]))
            if meta.get("format") != FORMAT_VERSION:
                raise ValueError(f"{path}: unsupported database format {meta.get('format')!r}")
            return cls(f["cell_ids"], f["embeddings"], f["bbox_w"] if "bbox_w" in f.files else None,
                       f["cell_size"] if "cell_size" in f.files else None, f["fine_desc"] if "fine_desc" in f.files else None, meta)

    # ---- use --------------------------------------------------------------------------------------------------
    def to_engine(self, engine, device="cuda", lo: int = 0, hi: Optional[int] = None):
        """Upload rows [lo, hi) as the engine's database shard (global row ids start at lo)."""
        hi = len(self) if hi is None else hi
        engine.db_set(torch.from_numpy(self.embeddings[lo:hi]).to(device), row_offset=lo, owner=(self, lo, hi))

    def search(self, engine, text_embeddings: torch.Tensor, k: int):
        """Search the WHOLE database on ``engine``. The engine is shared (eval_epoch, other databases): residency is
        decided by identity — the engine's ``db_owner`` token, replaced by every ``db_set`` — never by row count."""
        own = engine.db_owner
        if not (isinstance(own, tuple) and len(own) == 3 and own[0] is self and own[1] == 0 and own[2] == len(self)):
            self.to_engine(engine, text_embeddings.device)
        return engine.search(text_embeddings.contiguous().float(), k)
