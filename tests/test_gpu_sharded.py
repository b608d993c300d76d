"""GPU tier: the N>1 path with the REAL engine — several processes share GPU 0 (one Engine context each),
rendezvous over gloo (RCCL refuses duplicate devices; `sharded._all_gather` stages the one exchange through the host),
and drive exactly the branch `bench.py --gpus N` runs: HIP shard search -> t2l_pack_pairs -> ONE all_gather ->
t2l_merge_pairs. The merged ids must equal the unsharded float64 ranking (C oracle) integer for integer."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import c_oracle
from text2loc_amd import synth

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_rows, n_q, k, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from text2loc_amd.engine import Engine
    from text2loc_amd.sharded import QueryShardedSearcher, ShardedSearcher

    db, qs, _ = synth.make_retrieval_problem(n_rows, n_q, seed=21, noise=1.0)
    d_db, d_q = torch.from_numpy(db).cuda(), torch.from_numpy(qs).cuda()
    eng = Engine(0)
    ss = ShardedSearcher(eng)
    assert ss.engine is eng and ss._default_fns  # the packed-pairs branch
    lo, hi = ss.set_db_shard(d_db)
    idx, sc = ss.search(d_q, k)
    # the other layout: DB replicated, queries split
    eng_q = Engine(0)
    qsr = QueryShardedSearcher(eng_q)
    qsr.set_db(d_db)
    qi, qsc = qsr.search(d_q, k)
    torch.cuda.synchronize()
    out_q.put((rank, lo, hi, idx.cpu().numpy(), sc.cpu().numpy(), qi.cpu().numpy(), qsc.cpu().numpy()))
    dist.barrier()
    eng.close()
    eng_q.close()
    dist.destroy_process_group()


def _run(world, n_rows, n_q, k):
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_rows, n_q, k, out_q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [out_q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return sorted(res, key=lambda r: r[0])


@pytest.mark.parametrize("world,n_rows,n_q", [(4, 11259, 1024), (3, 11259, 4096), (2, 777, 130)])
def test_engine_row_sharded_equals_unsharded(world, n_rows, n_q):
    k = 10
    res = _run(world, n_rows, n_q, k)
    db, qs, _ = synth.make_retrieval_problem(n_rows, n_q, seed=21, noise=1.0)
    ridx, rsc = c_oracle.retrieve_topk(db, qs, k)
    spans = [(r[1], r[2]) for r in res]
    assert spans[0][0] == 0 and spans[-1][1] == n_rows and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    for rank, lo, hi, idx, sc, qi, qsc in res:
        assert np.array_equal(idx.astype(np.int64), ridx), f"rank {rank}: row-sharded ids differ from the float64 ranking"
        assert np.abs(sc - rsc).max() < 1e-12
        assert np.array_equal(qi.astype(np.int64), ridx), f"rank {rank}: query-sharded ids differ"
        assert np.abs(qsc - rsc).max() < 1e-12


def test_engine_empty_shard_inside_sharded_search():
    """N=5 over 4 ranks: shards of 2, 2, 1 and 0 rows — the empty shard answers -1 / -inf (no scan launch, no gather from a
    null database) and never wins the merge; k > rows of every shard."""
    k, n_rows = 4, 5
    res = _run(4, n_rows, 9, k)
    db, qs, _ = synth.make_retrieval_problem(n_rows, 9, seed=21, noise=1.0)
    ridx, rsc = c_oracle.retrieve_topk(db, qs, k)
    assert res[3][1] == res[3][2] == 5  # rank 3 holds nothing
    for rank, lo, hi, idx, sc, qi, qsc in res:
        assert np.array_equal(idx.astype(np.int64)[:, : ridx.shape[1]], ridx)
        assert np.abs(sc[:, : ridx.shape[1]] - rsc).max() < 1e-12


def test_empty_database_answers_minus_one():
    from text2loc_amd.engine import Engine

    eng = Engine(0)
    eng.db_set(torch.zeros((0, 256), device="cuda"))
    idx, sc = eng.search(torch.randn(7, 256, device="cuda"), 3)
    assert (idx.cpu().numpy() == -1).all() and np.isneginf(sc.cpu().numpy()).all()
    eng.close()
