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


@pytest.mark.parametrize("world,n_rows,n_q", [(8, 11259, 4096), (4, 11259, 1024), (3, 11259, 1500), (2, 777, 130)])
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


def test_world_8_with_empty_and_short_shards():
    """BASELINE config 3's process count with real engines: N=50 over 8 ranks = shards of 7,7,7,7,7,7,7,1 rows, k=10 > every
    shard; N=9 = 2,2,2,2,1,0,0,0 (three EMPTY shards inside the exchange)."""
    for n_rows in (50, 9):
        k = 10
        res = _run(8, n_rows, 33, k)
        db, qs, _ = synth.make_retrieval_problem(n_rows, 33, seed=21, noise=1.0)
        ridx, rsc = c_oracle.retrieve_topk(db, qs, k)
        kk = ridx.shape[1]
        assert [r[2] - r[1] for r in res] == [max(0, min(n_rows, (r + 1) * -(-n_rows // 8)) - min(n_rows, r * -(-n_rows // 8))) for r in range(8)]
        for rank, lo, hi, idx, sc, qi, qsc in res:
            assert np.array_equal(idx.astype(np.int64)[:, :kk], ridx), f"N={n_rows} rank {rank}"
            assert np.abs(sc[:, :kk] - rsc).max() < 1e-12
            assert (idx[:, kk:] == -1).all()


# ---- BASELINE config 5 on N ranks: cross_matcher.run_fine's world > 1 branch (consumer: evaluation/pipeline.py:90-204) ----------
def _fine_problem():
    from tests.test_gpu_fine import HintTable, _fine_args
    from tests.test_host_logic import StubCell, make_objects
    from text2loc_amd.cross_matcher import CrossMatch

    args = _fine_args(True)
    args.top_k, n_cells, Q = [1, 3, 5, 10], 37, 53
    cells_np = synth.make_cells(n_cells, seed=77, min_obj=4, max_obj=22)
    objects = make_objects(cells_np, 77)
    rng = np.random.default_rng(5)
    table = rng.standard_normal((Q, 6, 128)).astype(np.float32)
    model = CrossMatch(synth.KNOWN_CLASS, synth.COLOR_NAMES, args, language_encoder=HintTable(table))
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_fine_weights(5).items()}, strict=False)

    class D:
        def __init__(self, i):
            self.direction, self.object_color_text, self.object_label = "north", "red", f"q{i}x"

    class Pose:
        def __init__(self, i, cell):
            self.descriptions = [D(i)]
            self.cell_id = cell.id
            self.pose_w = np.array([*(cell.bbox_w[0:2] + rng.uniform(0, 1, 2) * cell.cell_size), 0.0])

    stub_cells = []
    for i in range(n_cells):
        lo = np.array([15.0 * (i % 6), 15.0 * (i // 6), 0.0])
        c = StubCell(f"sceneA_{i:03d}", np.concatenate([lo, lo + 30.0]), 30.0)
        c.objects = objects[i]
        stub_cells.append(c)
    retr = [np.array([stub_cells[j].id for j in rng.choice(n_cells, size=10, replace=False)]) for _ in range(Q)]
    poses = [Pose(i, stub_cells[int(np.where([c.id == retr[i][0] for c in stub_cells])[0][0])]) for i in range(Q)]

    class Ds:
        all_poses, all_cells = poses, stub_cells

    class Dl:
        dataset = Ds()

    return model.to("cuda").eval(), retr, Dl(), args


def _fine_worker(rank, world, port, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from text2loc_amd.cross_matcher import run_fine

    try:
        model, retr, dl, args = _fine_problem()
        acc_local, off_local = run_fine(model, retr, dl, args, return_offsets=True)  # no shard_layout: an initialised group changes nothing
        args.shard_layout = "auto"                                                   # opt-in: cells and pairs split over the ranks
        acc, offsets = run_fine(model, retr, dl, args, return_offsets=True)
        torch.cuda.synchronize()
        same = acc_local == acc and bool(np.array_equal(off_local, offsets))
        out_q.put((rank, acc, offsets, "" if same else "the un-sharded call inside an initialised group differs from the sharded one"))
    except Exception as e:  # (a worker that dies silently leaves the parent waiting for its timeout: report instead)
        out_q.put((rank, None, None, repr(e)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])  # (BASELINE config 5 names 8 GPUs; 2 is the smallest uneven split)
def test_run_fine_sharded_equals_single_process(world):
    """The distinct retrieved cells (37 -> blocks of 19/18 or 5/5/5/5/5/4/4/4 — BASELINE config 5 names 8 GPUs) and the
    530 (pose, cell) pairs are split across the ranks and all-gathered once each: accuracies AND every offset equal the single-process run bit for bit."""
    from text2loc_amd.cross_matcher import run_fine

    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fine_worker, args=(r, world, port, out_q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = sorted([out_q.get(timeout=240) for _ in range(world)], key=lambda r: r[0])
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
    assert all(p.exitcode == 0 for p in procs) and not [r[3] for r in res if r[3]], [r[3] for r in res]
    model, retr, dl, args = _fine_problem()
    acc1, off1 = run_fine(model, retr, dl, args, return_offsets=True)
    assert off1.shape == (53, 10, 2) and np.isfinite(off1).all()
    for rank, acc, off, _ in res:
        assert acc == acc1, (rank, acc, acc1)
        assert np.array_equal(off, off1), f"rank {rank}: offsets differ from the single-process run"


def test_empty_database_answers_minus_one():
    from text2loc_amd.engine import Engine

    eng = Engine(0)
    eng.db_set(torch.zeros((0, 256), device="cuda"))
    idx, sc = eng.search(torch.randn(7, 256, device="cuda"), 3)
    assert (idx.cpu().numpy() == -1).all() and np.isneginf(sc.cpu().numpy()).all()
    eng.close()


# ---- RCCL itself (backend "nccl"), as far as one GPU allows: a single-rank process group on cuda:0 --------------------------------
def _rccl_worker(port, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from text2loc_amd.engine import Engine
    from text2loc_amd.losses import _GatherRowsFn
    from text2loc_amd.optim import all_reduce_flat
    from text2loc_amd.sharded import ShardedSearcher, _all_gather

    n_rows, n_q, k = 11259, 512, 10
    db, qs, _ = synth.make_retrieval_problem(n_rows, n_q, seed=21, noise=1.0)
    d_db, d_q = torch.from_numpy(db).cuda(), torch.from_numpy(qs).cuda()
    eng = Engine(0)
    ss = ShardedSearcher(eng)
    ss.set_db_shard(d_db)
    ref_i, ref_s = eng.search(d_q, k)
    # the three lines of ShardedSearcher.search's world > 1 branch, on the real exchange buffers, with the collective on the device:
    # scan + re-rank fill the rank's own block, RCCL gathers the blocks (u8), the merge kernel ranks them — ordered by streams only
    ok = True
    for _ in range(20):
        own, idx, sc, allb, bb, so = ss._scratch_for(n_q, k, d_q.device)
        allb.zero_()
        eng.search(d_q, k, out=(idx, sc))
        _all_gather(dist, allb, own, None)
        mi, ms = eng.merge_gathered(allb, bb, so, 1, n_q, k)
        ok = ok and bool(torch.equal(mi, ref_i)) and bool(torch.equal(ms, ref_s))
    # every dtype the N > 1 paths send through all_gather / all_reduce
    dtypes_ok = {}
    for dt in (torch.uint8, torch.int32, torch.float64, torch.float32):
        x = (torch.arange(4096, device="cuda") % 251).to(dt)
        y = torch.empty_like(x)
        _all_gather(dist, y, x, None)
        dtypes_ok[str(dt)] = bool(torch.equal(x, y))
    g = torch.full((1 << 20,), 0.5, device="cuda")
    all_reduce_flat(g, None, mean=True)
    e = torch.randn(16, 256, device="cuda", requires_grad=True)
    ge = _GatherRowsFn.apply(e, None)  # (gather_rows_with_grad short-cuts a world of 1)
    ge.square().sum().backward()
    torch.cuda.synchronize()
    out_q.put((ok, dtypes_ok, float(g.min()), float(g.max()), tuple(ge.shape), float((e.grad - 2 * e.detach()).abs().max())))
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


def test_rccl_single_rank_group_drives_the_exchange_on_the_device():
    """RCCL refuses two ranks on one GPU, so a box with one GPU can only run a world of 1 over it — which still proves what the
    gloo runs above cannot: the library initialises in this image, the exchange block (u8), ids (i32), scores (f64) and gradients
    (f32) are dtypes it accepts, and scan -> re-rank -> all_gather -> merge is correctly ordered when the collective runs on the
    device (RCCL's own stream against the engine's launches on the current stream)."""
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), out_q))
    p.start()
    ok, dtypes_ok, gmin, gmax, gshape, gerr = out_q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert ok
    assert all(dtypes_ok.values()), dtypes_ok
    assert gmin == gmax == 0.5
    assert gshape == (16, 256) and gerr < 1e-6
