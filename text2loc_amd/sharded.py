"""Row-sharded database search across the GPUs of one node (new design — the reference is
single-process, single-device: training/coarse.py:235, no torch.distributed call sites).

One process per GPU. Rank r owns DB rows [lo_r, hi_r) (contiguous, read-only, resident in its HBM). Every
rank holds all queries. A search is: local fused top-k on the shard (global row ids via ``row_offset``)
-> ONE collective, ``all_gather`` of the per-rank [Q,K] {float64 score, row id} records over RCCL/xGMI
(Q*K*16 bytes per rank: latency bound, link bandwidth is irrelevant) -> merge on every rank by
(score desc, id asc). Because every shard's list is already exact in float64, the merged top-k equals the
unsharded result bit for bit. Cell encoding is embarrassingly parallel over cells: no collective.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np


def shard_bounds(n_rows: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous row shard of rank ``rank``: ceil(n/world) rows each, the tail shards may be short or empty."""
    per = -(-n_rows // world)
    lo = min(n_rows, rank * per)
    return lo, min(n_rows, lo + per)


def merge_topk_host(idx: np.ndarray, score: np.ndarray, k: int):
    """Host restatement of the merge (numpy) — used by the CPU (gloo) tests of the N>1 logic and as the
    checker of the HIP merge kernel. idx i[P,Q,K] (-1 = empty), score f64[P,Q,K]."""
    P, Q, K = idx.shape
    ids = idx.transpose(1, 0, 2).reshape(Q, P * K).astype(np.int64)
    sc = score.transpose(1, 0, 2).reshape(Q, P * K).astype(np.float64).copy()
    sc[ids < 0] = -np.inf
    big = np.where(ids < 0, np.iinfo(np.int64).max, ids)
    out_i = np.empty((Q, k), dtype=np.int64)
    out_s = np.empty((Q, k), dtype=np.float64)
    for q in range(Q):
        order = np.lexsort((big[q], -sc[q]))[:k]
        out_i[q], out_s[q] = ids[q, order], sc[q, order]
    out_i[np.isneginf(out_s)] = -1
    return out_i, out_s


merge_topk = merge_topk_host  # name used by the tests


def _all_gather(dist, out, inp, group):
    """``all_gather_into_tensor`` on the tensors' own device under RCCL ("nccl"); under gloo (CPU tests, and the
    several-processes-on-one-GPU dry run of the engine path, where RCCL refuses duplicate devices) device tensors are
    staged through the host."""
    if inp.is_cuda and dist.get_backend(group) != "nccl":
        h_out = out.new_empty(out.shape, device="cpu")
        dist.all_gather_into_tensor(h_out, inp.cpu().contiguous(), group=group)
        out.copy_(h_out)
    else:
        dist.all_gather_into_tensor(out, inp.contiguous(), group=group)


def world_rank(group=None) -> Tuple[int, int]:
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def gather_rows(local, n_total: int, group=None):
    """Embarrassingly parallel stages (cell encoding, the fine stage's (pose, cell) pairs — BASELINE config 5): rank r
    computed rows ``shard_bounds(n_total, world, r)`` of a row-wise result; every rank gets all ``n_total`` rows back
    through ONE all_gather of fixed-size blocks (the ragged tail is padded, then trimmed). No-op for a single process."""
    import torch
    import torch.distributed as dist

    world, rank = world_rank(group)
    if world == 1:
        return local
    lo, hi = shard_bounds(n_total, world, rank)
    assert local.shape[0] == hi - lo, (local.shape, lo, hi)
    per = -(-n_total // world)
    block = local.new_zeros((per,) + tuple(local.shape[1:]))
    block[: hi - lo] = local
    out = local.new_empty((world * per,) + tuple(local.shape[1:]))
    _all_gather(dist, out, block, group)
    # shard r starts at row min(n, r*per) = its block's offset in the gathered buffer (every shard before the tail one is
    # full), so the first n_total rows ARE the result in order
    return out[:n_total]


def assert_replicated(tensors, group=None, what: str = "tensors"):
    """Every rank must hold the SAME ``tensors`` (AutoSearcher's contract: same database, same queries). ONE small all_reduce
    (MAX over [f, -f] of a fingerprint: shapes, a float64 sum and a position-weighted float64 sum per tensor) proves it, and a
    rank whose data differs — e.g. a DistributedSampler'd dataloader feeding each rank its own queries — raises on EVERY rank
    instead of returning slices of different query sets stitched together. No-op for a single process."""
    import torch
    import torch.distributed as dist

    world, _ = world_rank(group)
    if world == 1:
        return
    fp, n_bad = [], 0
    for t in tensors:
        x = t.detach().reshape(-1).to(torch.float64)
        # non-finite values (an overflowed / untrained encoder) are counted and zeroed: NaN != NaN would otherwise read as "the
        # ranks differ" on identical data; the count is part of the fingerprint and is reported as what it is
        finite = torch.isfinite(x)
        bad = int(x.numel() - int(finite.sum()))
        n_bad += bad
        x = torch.where(finite, x, torch.zeros_like(x))
        w = torch.arange(1, x.numel() + 1, dtype=torch.float64, device=x.device) % 8191.0 + 1.0
        fp += [float(t.dim())] + [float(d) for d in t.shape] + [float(bad), float(x.sum()), float((x * w).sum())]
    f = torch.tensor(fp, dtype=torch.float64)
    both = torch.cat([f, -f])
    if dist.get_backend(group) == "nccl":
        both = both.cuda()
    try:
        dist.all_reduce(both, op=dist.ReduceOp.MAX, group=group)
    except RuntimeError as e:  # fingerprints of different LENGTH (different ranks of the tensors): the collective itself objects
        raise RuntimeError(f"{what} differ across ranks (all_reduce of the fingerprints failed: {e})") from e
    both = both.cpu()
    hi, lo = both[: len(fp)], -both[len(fp):]
    if n_bad and torch.equal(hi, lo):
        import warnings

        warnings.warn(f"{what}: {n_bad} non-finite values (the same on every rank) — the encoder overflowed or is untrained; "
                      "retrieval over them is meaningless", RuntimeWarning, stacklevel=2)
    if not torch.equal(hi, lo):
        raise RuntimeError(f"{what} differ across the {world} ranks (fingerprint max != min in {int((hi != lo).sum())} of {len(fp)} "
                           "fields): sharded retrieval needs every rank to pass the same database and the same queries — evaluate "
                           "the full query set on every rank (no DistributedSampler), or use retrieve= with your own layout")


class ShardedSearcher:
    """search_fn(queries, k) -> (idx[Q,K] int32, score[Q,K] float64) on this rank's shard (global ids);
    merge_fn(idx[P,Q,K], score[P,Q,K]) -> (idx[Q,K], score[Q,K]). With an ``Engine`` both default to the
    HIP kernels; the CPU tests inject host stand-ins to exercise the collective plumbing under gloo."""

    def __init__(self, engine=None, group=None, search_fn: Optional[Callable] = None,
                 merge_fn: Optional[Callable] = None):
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.engine = engine
        self._default_fns = search_fn is None and merge_fn is None
        self.search_fn = search_fn or (lambda q, k: engine.search(q, k))
        self.merge_fn = merge_fn or (lambda i, s: engine.merge_topk(i, s))
        self._scratch = None  # intermediates of the engine path: ONE set, sized to the largest (Q, k) seen and sliced

    def set_db_shard(self, all_rows_or_shard, n_total: Optional[int] = None):
        """Give either the full [N,256] matrix (this rank keeps its slice) or this rank's slice + n_total."""
        if n_total is None:
            n_total = int(all_rows_or_shard.shape[0])
            lo, hi = shard_bounds(n_total, self.world, self.rank)
            shard = all_rows_or_shard[lo:hi].contiguous()
        else:
            lo, hi = shard_bounds(n_total, self.world, self.rank)
            shard = all_rows_or_shard
            assert shard.shape[0] == hi - lo
        self.lo, self.hi, self.n_total = lo, hi, n_total
        if self.engine is not None:
            self.engine.db_set(shard, row_offset=lo)
        return lo, hi

    def _scratch_for(self, Q: int, k: int, dev):
        """(own block u8[block_bytes] with idx / score views into it, all blocks u8[world * block_bytes], block_bytes,
        score_offset): ONE scratch set per (Q, k) geometry last used — it is re-made when the geometry changes, so variable
        query-batch sizes do not accumulate buffers. Reuse is safe because every consumer is ordered on the stream that owns
        the set; a caller on ANOTHER stream gets fresh tensors for that call."""
        import torch

        stream = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0
        sc = self._scratch
        if sc is not None and sc["key"] == (Q, k, dev) and sc["stream"] != stream:
            own, idx, score, bb, so = self.engine.result_block(Q, k, dev)
            return own.view(-1), idx, score, torch.empty(self.world * bb, dtype=torch.uint8, device=dev), bb, so
        if sc is None or sc["key"] != (Q, k, dev):
            own, idx, score, bb, so = self.engine.result_block(Q, k, dev)
            sc = self._scratch = {"key": (Q, k, dev), "stream": stream, "own": own.view(-1), "idx": idx, "score": score,
                                  "all": torch.empty(self.world * bb, dtype=torch.uint8, device=dev), "bb": bb, "so": so}
        return sc["own"], sc["idx"], sc["score"], sc["all"], sc["bb"], sc["so"]

    def search(self, queries, k: int, out=None):
        """``out`` = (idx i32[Q,k], score f64[Q,k]) to write the merged result into (engine path; a serving loop that
        rotates a few output sets avoids two allocations per call)."""
        import torch

        if self.world > 1 and self.engine is not None and self._default_fns:
            # ONE collective and THREE launches per rank: scan + re-rank write ids and scores into one block, the blocks of all
            # ranks are all-gathered back to back (12 B per candidate), t2l_merge_gathered ranks them. The exchange buffers are
            # scratch, reused call after call (a step is tens of microseconds of GPU time: per-call allocations would make the
            # host the bottleneck)
            Q = int(queries.shape[0])
            own, idx, sc, allb, bb, so = self._scratch_for(Q, int(k), queries.device)
            self.engine.search(queries, k, out=(idx, sc))
            _all_gather(self.dist, allb, own, self.group)
            return self.engine.merge_gathered(allb, bb, so, self.world, Q, int(k), out=out)
        if self.world == 1 and self.engine is not None and self._default_fns:
            return self.engine.search(queries, k, out=out)
        idx, sc = self.search_fn(queries, k)
        if self.world == 1:
            return idx, sc
        Q = idx.shape[0]
        # rank-major concatenation along dim 0 (the layout both RCCL and gloo accept) == [world][Q][k]
        all_i = torch.empty((self.world * Q, k), dtype=idx.dtype, device=idx.device)
        all_s = torch.empty((self.world * Q, k), dtype=sc.dtype, device=sc.device)
        _all_gather(self.dist, all_i, idx, self.group)
        _all_gather(self.dist, all_s, sc, self.group)
        return self.merge_fn(all_i.view(self.world, Q, k), all_s.view(self.world, Q, k))


class QueryShardedSearcher:
    """The other way to use N GPUs when the whole DB fits every GPU (11,259 x 256 x 4 B = 11.5 MB does): the database
    is REPLICATED, the QUERIES are split (rank r answers queries [r*ceil(Q/P), ...)), no exchange on the data path; one
    optional ``all_gather`` hands every rank the complete [Q,K] result (what ``eval_epoch`` wants). Throughput scales with
    the number of GPUs because every rank scans the full DB for 1/P of the queries, whereas row-sharding
    (``ShardedSearcher``, the north-star layout, needed once the DB outgrows one GPU) makes every rank touch every query."""

    def __init__(self, engine=None, group=None, search_fn: Optional[Callable] = None):
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.engine = engine
        self.search_fn = search_fn or (lambda q, k: engine.search(q, k))

    def set_db(self, all_rows):
        if self.engine is not None:
            self.engine.db_set(all_rows.contiguous(), row_offset=0)
        return 0, int(all_rows.shape[0])

    def search(self, queries, k: int, gather: bool = True):
        import torch

        Q = int(queries.shape[0])
        lo, hi = shard_bounds(Q, self.world, self.rank)
        per = -(-Q // self.world)
        idx, sc = self.search_fn(queries[lo:hi].contiguous(), k) if hi > lo else (
            torch.empty((0, k), dtype=torch.int32, device=queries.device), torch.empty((0, k), dtype=torch.float64, device=queries.device))
        if self.world == 1 or not gather:
            return idx, sc
        pi = torch.full((per, k), -1, dtype=idx.dtype, device=idx.device)   # ragged tail -> fixed-size records
        ps = torch.full((per, k), float("-inf"), dtype=sc.dtype, device=sc.device)
        pi[: hi - lo], ps[: hi - lo] = idx, sc
        all_i = torch.empty((self.world * per, k), dtype=idx.dtype, device=idx.device)
        all_s = torch.empty((self.world * per, k), dtype=sc.dtype, device=sc.device)
        _all_gather(self.dist, all_i, pi, self.group)
        _all_gather(self.dist, all_s, ps, self.group)
        return all_i[:Q], all_s[:Q]


# ---- which layout for which database ------------------------------------------------------------------------------------------
# A resident row costs 2.5 KiB of HBM in the engine (f32 row 1 KiB + f16 plane 0.5 KiB + split-bf16 plane 1 KiB). Row-sharding
# is what a database needs that does NOT fit one GPU; for one that does, it makes every rank convert, scan against and re-rank
# EVERY query and merge P lists per query (measured per-rank floor of an 8-GPU step at N = 11,259 x Q = 4,096: 39 us against a
# 45 us single-GPU step — no scaling), whereas replicating it and splitting the QUERIES has no data-path exchange at all and
# scales with the number of GPUs. `layout="auto"`: query-sharded while the whole database takes at most `replicate_fraction`
# (default a quarter) of one GPU's HBM — at 288 GB that is ~28 M rows; KITTI360Pose has 11 k —, row-sharded beyond.
ROW_BYTES_RESIDENT = 2560


def choose_layout(n_rows: int, hbm_bytes: Optional[int] = None, replicate_fraction: float = 0.25) -> str:
    if hbm_bytes is None:
        try:
            import torch

            hbm_bytes = torch.cuda.get_device_properties(torch.cuda.current_device()).total_memory if torch.cuda.is_available() else 288 << 30
        except Exception:
            hbm_bytes = 288 << 30
    return "query" if n_rows * ROW_BYTES_RESIDENT <= replicate_fraction * hbm_bytes else "row"


class AutoSearcher:
    """``set_db(all_rows)`` + ``search(queries, k)`` over N ranks with the layout ``choose_layout`` picks (or a forced one): every
    rank passes the SAME full [N,256] matrix and the SAME queries and gets the complete [Q,k] result back — what
    ``coarse.eval_epoch`` calls when the caller asked for it (``args.shard_layout``); ``check_replicated`` (default on) proves the
    "same" with one small all_reduce per set_db / search call."""

    def __init__(self, engine=None, group=None, layout: str = "auto", search_fn: Optional[Callable] = None, merge_fn: Optional[Callable] = None,
                 check_replicated: bool = True):
        if layout not in ("auto", "query", "row"):
            raise ValueError("layout must be 'auto', 'query' or 'row'")
        self._args = (engine, group, search_fn, merge_fn)
        self.check_replicated = bool(check_replicated)
        self.layout_request, self.layout, self.impl = layout, None, None

    def set_db(self, all_rows):
        engine, group, search_fn, merge_fn = self._args
        if self.check_replicated:
            assert_replicated([all_rows], group, "databases")
        self.layout = self.layout_request if self.layout_request != "auto" else choose_layout(int(all_rows.shape[0]))
        if self.layout == "query":
            self.impl = QueryShardedSearcher(engine, group, search_fn=search_fn)
            return self.impl.set_db(all_rows)
        self.impl = ShardedSearcher(engine, group, search_fn=search_fn, merge_fn=merge_fn)
        return self.impl.set_db_shard(all_rows)

    def search(self, queries, k: int):
        if self.check_replicated:
            assert_replicated([queries], self._args[1], "query sets")
        return self.impl.search(queries, k)
