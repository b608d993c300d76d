"""CPU tier: the N>1 path — shard bounds, host merge, and the all_gather plumbing of ShardedSearcher under
gloo with world_size 2 (and 3, ragged/empty shards). The per-shard search is stood in by the oracle; on a GPU
box the same class drives the HIP kernels (tests/test_gpu_search.py covers those)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import t2l_oracle as O
from text2loc_amd import synth
from text2loc_amd.sharded import merge_topk_host, shard_bounds


@pytest.mark.parametrize("n,world", [(11259, 8), (10, 3), (3, 8), (0, 2), (64, 1)])
def test_shard_bounds_partition_rows(n, world):
    spans = [shard_bounds(n, world, r) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == n
    for (a, b), (c, d) in zip(spans, spans[1:]):
        assert b == c and a <= b and c <= d
    assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= -(-n // world)


def test_merge_topk_host_equals_unsharded():
    db, qs, _ = synth.make_retrieval_problem(777, 40, seed=8, noise=2.0)
    ridx, rsc = O.retrieve_topk(db, qs, 10)
    parts_i, parts_s = [], []
    for r in range(5):
        lo, hi = shard_bounds(len(db), 5, r)
        i, s = O.retrieve_topk(db[lo:hi], qs, 10)
        parts_i.append(i + lo)
        parts_s.append(s)
    idx, sc = merge_topk_host(np.stack(parts_i), np.stack(parts_s), 10)
    assert np.array_equal(idx, ridx) and np.abs(sc - rsc).max() < 1e-12


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_rows, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    from text2loc_amd.sharded import ShardedSearcher

    db, qs, _ = synth.make_retrieval_problem(n_rows, 33, seed=6, noise=2.0)
    K = 10
    state = {}

    def search_fn(q, k):  # oracle stand-in for the HIP shard search: global ids, -1 / -inf padding
        lo, hi = state["lo"], state["hi"]
        i, s = O.retrieve_topk(db[lo:hi], q.numpy(), k) if hi > lo else (np.zeros((len(q), 0), np.int64),
                                                                         np.zeros((len(q), 0)))
        idx = np.full((len(q), k), -1, dtype=np.int32)
        sc = np.full((len(q), k), -np.inf)
        idx[:, : i.shape[1]] = i + lo
        sc[:, : s.shape[1]] = s
        return torch.from_numpy(idx), torch.from_numpy(sc)

    def merge_fn(i, s):
        a, b = merge_topk_host(i.numpy(), s.numpy(), K)
        return torch.from_numpy(a), torch.from_numpy(b)

    ss = ShardedSearcher(engine=None, search_fn=search_fn, merge_fn=merge_fn)
    state["lo"], state["hi"] = ss.set_db_shard(torch.from_numpy(db))
    idx, sc = ss.search(torch.from_numpy(qs), K)
    ridx, rsc = O.retrieve_topk(db, qs, K)
    kk = ridx.shape[1]
    ok = np.array_equal(idx.numpy()[:, :kk], ridx) and np.abs(sc.numpy()[:, :kk] - rsc).max() < 1e-12  # BLAS order
    out_q.put((rank, bool(ok), state["lo"], state["hi"]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_rows", [(2, 501), (3, 7)])
def test_sharded_search_under_gloo(world, n_rows):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_rows, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res), res
    spans = sorted((lo, hi) for _, _, lo, hi in res)
    assert spans[0][0] == 0 and spans[-1][1] == n_rows


def _qworker(rank, world, port, n_q, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    from text2loc_amd.sharded import QueryShardedSearcher

    db, qs, _ = synth.make_retrieval_problem(300, n_q, seed=4, noise=2.0)
    K = 10

    def search_fn(q, k):  # oracle stand-in for the HIP search over the replicated DB
        i, s = O.retrieve_topk(db, q.numpy(), k)
        return torch.from_numpy(i.astype(np.int32)), torch.from_numpy(s)

    qs_ = QueryShardedSearcher(engine=None, search_fn=search_fn)
    idx, sc = qs_.search(torch.from_numpy(qs), K)
    ridx, rsc = O.retrieve_topk(db, qs, K)
    ok = idx.shape == (n_q, K) and np.array_equal(idx.numpy(), ridx) and np.abs(sc.numpy() - rsc).max() < 1e-12
    li, _ = qs_.search(torch.from_numpy(qs), K, gather=False)
    lo, hi = shard_bounds(n_q, world, rank)
    ok = ok and np.array_equal(li.numpy(), ridx[lo:hi])
    out_q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_q", [(2, 33), (3, 2)])
def test_query_sharded_search_under_gloo(world, n_q):
    """replicated DB, split queries (ragged / empty query shards), results gathered to every rank"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_qworker, args=(r, world, port, n_q, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res


def _gworker(rank, world, port, n_rows, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    from text2loc_amd.sharded import gather_rows

    full = torch.arange(n_rows * 6, dtype=torch.float32).reshape(n_rows, 3, 2)  # stands for [pairs, ...] match results
    lo, hi = shard_bounds(n_rows, world, rank)
    got = gather_rows(full[lo:hi].clone(), n_rows)
    out_q.put((rank, bool(torch.equal(got, full))))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_rows", [(2, 41), (3, 7), (3, 2), (2, 0)])
def test_gather_rows_under_gloo(world, n_rows):
    """the config-5 split of the fine stage: contiguous row blocks per rank (ragged / empty tails), one all_gather"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gworker, args=(r, world, port, n_rows, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res


# ---- data-parallel training plumbing: gathered global loss + summed gradients ------------------------
def _torch_contrastive(im, s, t=0.1):
    im = im / torch.norm(im, dim=1, keepdim=True)
    s = s / torch.norm(s, dim=1, keepdim=True)
    sim = im @ s.t()
    num = torch.exp(torch.diag(sim) / t)
    den = torch.exp(sim / t)
    return torch.mean(-torch.log(num / den.sum(0)) - torch.log(num / den.sum(1)))


def _dp_problem(world, b_local):
    g = torch.Generator().manual_seed(3)
    n = world * b_local
    x_txt, x_cell = torch.randn(n, 24, generator=g), torch.randn(n, 40, generator=g)
    w_txt, w_cell = torch.randn(24, 256, generator=g) * 0.2, torch.randn(40, 256, generator=g) * 0.2
    return x_txt.double(), x_cell.double(), w_txt.double(), w_cell.double()


def _dp_worker(rank, world, port, b_local, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    from text2loc_amd.losses import gather_rows_with_grad
    from text2loc_amd.optim import all_reduce_flat

    x_txt, x_cell, w_txt, w_cell = _dp_problem(world, b_local)
    w_txt.requires_grad_(True)
    w_cell.requires_grad_(True)
    lo = rank * b_local
    im = gather_rows_with_grad(x_txt[lo:lo + b_local] @ w_txt)     # local rows -> the global [W*B,256] matrix, on every rank
    s = gather_rows_with_grad(x_cell[lo:lo + b_local] @ w_cell)
    loss = _torch_contrastive(im, s)
    loss.backward()                                               # d(global loss)/d(local rows) only
    flat = torch.cat([w_txt.grad.reshape(-1), w_cell.grad.reshape(-1)])
    local = flat.clone()
    all_reduce_flat(flat, None)                                   # ONE collective, SUM
    out_q.put((rank, float(loss.detach()), local.numpy(), flat.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,b_local", [(2, 8), (3, 5)])
def test_gathered_loss_and_summed_gradients_equal_the_single_process_step(world, b_local):
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, b_local, out_q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([out_q.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    x_txt, x_cell, w_txt, w_cell = _dp_problem(world, b_local)
    w_txt.requires_grad_(True)
    w_cell.requires_grad_(True)
    loss = _torch_contrastive(x_txt @ w_txt, x_cell @ w_cell)
    loss.backward()
    ref = torch.cat([w_txt.grad.reshape(-1), w_cell.grad.reshape(-1)]).numpy()
    for rank, l, local, summed in res:
        assert abs(l - float(loss.detach())) < 1e-12           # every rank evaluates the GLOBAL loss
        assert np.abs(summed - ref).max() < 1e-12                # sum of the per-rank gradients == the full-batch gradient
        assert np.abs(local - ref).max() > 1e-6                  # ... which no rank has by itself
    assert np.abs(sum(r[2] for r in res) - ref).max() < 1e-12


# ---- the layout policy (sharded.choose_layout / AutoSearcher) ---------------------------------------------------------------------
def test_choose_layout_replicates_what_fits_and_row_shards_beyond():
    from text2loc_amd.sharded import ROW_BYTES_RESIDENT, choose_layout

    hbm = 288 << 30
    assert choose_layout(11259, hbm) == "query"              # KITTI360Pose: 29 MB resident
    assert choose_layout(20_000_000, hbm) == "query"         # 51 GB: still a sixth of one GPU
    assert choose_layout(40_000_000, hbm) == "row"           # 102 GB: shard it
    edge = int(0.25 * hbm) // ROW_BYTES_RESIDENT
    assert choose_layout(edge, hbm) == "query" and choose_layout(edge + 1, hbm) == "row"


def _aworker(rank, world, port, layout, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    from text2loc_amd.sharded import AutoSearcher

    db, qs, _ = synth.make_retrieval_problem(257, 21, seed=8, noise=2.0)
    K = 5
    state = {}

    def search_fn(q, k):  # stands in for the HIP search over whatever rows this rank holds
        lo, hi = state.get("lo", 0), state.get("hi", len(db))
        i, s = O.retrieve_topk(db[lo:hi], q.numpy(), k)
        return torch.from_numpy((i + lo).astype(np.int32)), torch.from_numpy(s)

    def merge_fn(i, s):
        a, b = merge_topk_host(i.numpy(), s.numpy(), K)
        return torch.from_numpy(a), torch.from_numpy(b)

    srch = AutoSearcher(engine=None, layout=layout, search_fn=search_fn, merge_fn=merge_fn)
    lo, hi = srch.set_db(torch.from_numpy(db))
    if srch.layout == "row":
        state["lo"], state["hi"] = lo, hi
    idx, sc = srch.search(torch.from_numpy(qs), K)
    ridx, rsc = O.retrieve_topk(db, qs, K)
    ok = np.array_equal(np.asarray(idx), ridx) and np.abs(np.asarray(sc) - rsc).max() < 1e-12
    out_q.put((rank, bool(ok), srch.layout))
    dist.destroy_process_group()


@pytest.mark.parametrize("layout,expect", [("auto", "query"), ("row", "row"), ("query", "query")])
def test_auto_searcher_under_gloo(layout, expect):
    """every rank hands over the same full database and queries and gets the complete result, in either layout"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_aworker, args=(r, world, port, layout, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok and lay == expect for _, ok, lay in res), res


# ---- eval_epoch's retrieval under an initialised process group: local unless asked, and loud when the ranks' data differ -------------
class _FakeEngine:
    """db_set / search with the oracle's arithmetic (the collective plumbing is what is under test)"""

    def db_set(self, emb, row_offset=0):
        self.db, self.off = emb.numpy(), row_offset

    def search(self, q, k):
        i, s = O.retrieve_topk(self.db, q.numpy(), k)
        return torch.from_numpy((i + self.off).astype(np.int32)), torch.from_numpy(s)


class _FakeModel:
    def __init__(self, layout):
        import argparse

        self.args = argparse.Namespace(shard_layout=layout) if layout != "absent" else argparse.Namespace()
        self._eng = _FakeEngine()

    def engine(self):
        return self._eng


def _rworker(rank, world, port, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    from text2loc_amd.coarse import _engine_retrieve
    from text2loc_amd.sharded import assert_replicated

    db, qs, _ = synth.make_retrieval_problem(129, 12, seed=4, noise=2.0)
    own_qs = qs if rank == 0 else np.ascontiguousarray(qs[::-1])  # what a DistributedSampler does: every rank its own queries
    ridx, _ = O.retrieve_topk(db, own_qs, 5)
    res = {}
    # (1) no shard_layout (absent or None): an initialised group changes nothing — no collective, each rank's own answer
    for layout in ("absent", None):
        idx, _ = _engine_retrieve(_FakeModel(layout))(torch.from_numpy(db), torch.from_numpy(own_qs), 5)
        res[f"local_{layout}"] = bool(np.array_equal(idx, ridx))
    # (2) opted in with the SAME data on every rank: complete result on every rank, both layouts
    full, _ = O.retrieve_topk(db, qs, 5)
    for layout in ("query", "auto"):
        idx, _ = _engine_retrieve(_FakeModel(layout))(torch.from_numpy(db), torch.from_numpy(qs), 5)
        res[f"sharded_{layout}"] = bool(np.array_equal(idx, full))
    # (3) opted in with rank-different queries: RuntimeError on EVERY rank, before any slice is stitched
    try:
        _engine_retrieve(_FakeModel("query"))(torch.from_numpy(db), torch.from_numpy(own_qs), 5)
        res["differs_raises"] = False
    except RuntimeError as e:
        res["differs_raises"] = "differ across" in str(e)
    # (4) differing SHAPES are caught too (fingerprints of equal length, different fields)
    try:
        assert_replicated([torch.zeros(3 + rank, 2)], None, "things")
        res["shape_raises"] = False
    except RuntimeError:
        res["shape_raises"] = True
    # (5) identical data holding NaN / inf is NOT rank divergence (NaN != NaN): a warning that names the non-finite values
    import warnings

    bad = torch.from_numpy(db.copy())
    bad[3, 7], bad[9, 0] = float("nan"), float("inf")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        try:
            assert_replicated([bad], None, "databases")
            res["nan_same_data_passes"] = any("non-finite" in str(x.message) for x in w)
        except RuntimeError:
            res["nan_same_data_passes"] = False
    # ... and a NaN on ONE rank only still raises
    one = torch.from_numpy(db.copy())
    if rank == 1:
        one[0, 0] = float("nan")
    try:
        assert_replicated([one], None, "databases")
        res["nan_one_rank_raises"] = False
    except RuntimeError:
        res["nan_one_rank_raises"] = True
    out_q.put((rank, res))
    dist.destroy_process_group()


def test_eval_retrieval_is_local_unless_asked_and_checks_replication():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rworker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, r in res:
        assert all(r.values()), (rank, r)
