"""GPU tier, SURVEY.md 8e / f-4b: data-parallel training of the object branch. Two processes share GPU 0 (one engine context
each, gloo rendezvous: RCCL refuses duplicate devices, the collectives are staged through the host) and run
``ContrastiveLoss(gather=True)`` (all_gather of the [B,256] text and cell embeddings -> the GLOBAL contrastive matrix) +
``optim.Adam(data_parallel=True)`` (ONE all_reduce over the engine-owned flat gradient buffer). The reference is single
process (training/coarse.py:31-58), so the checker is the single-process engine itself: the same 2 x 32 cells as two
accumulated backward passes of the global loss (BatchNorm statistics are per rank = per block in both). Round 4: also 8 ranks
x 8 cells — the published batch of 64 in the 8-GPU node's geometry — as eight accumulated passes."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from text2loc_amd import synth

pytestmark = pytest.mark.gpu

LR = 1e-3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(world, b_local):
    from tests.test_gpu_train_loop import TableText, _args
    from tests.test_host_logic import make_objects
    from text2loc_amd.cell_retrieval import CellRetrievalNetwork

    n = world * b_local
    cells = synth.make_cells(n, seed=61)
    objects = make_objects(cells, 61)
    model = CellRetrievalNetwork(synth.KNOWN_CLASS, synth.COLOR_NAMES, _args(), language_encoder=TableText(n, 9))
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_object_branch_weights(8).items()}, strict=False)
    for layer in model.obj_inter_module:  # dropout off: the two runs draw different mask seeds
        layer.dropout.p = layer.dropout1.p = layer.dropout2.p = 0.0
        layer.self_attn.dropout = 0.0
    return model.to("cuda").train(), objects


def _params(model):
    return {n: p.detach().cpu().numpy().copy() for n, p in model.named_parameters()
            if n.startswith(("obj_inter_module.", "object_encoder.mlp_merge", "object_encoder.pos_encoder", "language_encoder."))}


def _worker(rank, world, b_local, port, out_q, sync_bn=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from text2loc_amd.losses import ContrastiveLoss
    from text2loc_amd.optim import Adam

    model, objects = _build(world, b_local)
    opt = Adam(model, lr=LR, data_parallel=True, sync_bn=sync_bn)
    crit = ContrastiveLoss(0.1, gather=True)
    lo = rank * b_local
    ids = list(range(lo, lo + b_local))
    opt.zero_grad()
    loss = crit(model.encode_text(ids), model.encode_objects(objects[lo:lo + b_local]))
    loss.backward()
    local = model.train_flat_grad().detach().cpu().numpy().copy()
    opt.all_reduce_grads()
    summed = model.train_flat_grad().detach().cpu().numpy().copy()
    table_grad = model.language_encoder.table.grad.detach().cpu().numpy().copy()
    opt._dp = False  # the gradients are reduced already
    opt.step()
    torch.cuda.synchronize()
    out_q.put((rank, float(loss.detach()), local, summed, table_grad, _params(model)))
    dist.barrier()
    dist.destroy_process_group()


# (2, 32): the round-3 case; (8, 8): BASELINE config 4's batch of 64 as 8 ranks x 8 cells — the 8-GPU node's geometry
@pytest.mark.parametrize("world,b_local", [(2, 32), (8, 8)])
def test_ranks_equal_the_accumulated_single_process_step(world, b_local):
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, b_local, port, out_q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([out_q.get(timeout=900) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0

    # the checker: one process, the global loss differentiated block by block into the same gradient buffers (pass r: rank r's
    # cells and sentences live, every other block a constant — exactly what rank r differentiates)
    from text2loc_amd.losses import ContrastiveLoss
    from text2loc_amd.optim import Adam

    model, objects = _build(world, b_local)
    opt = Adam(model, lr=LR)
    crit = ContrastiveLoss(0.1)
    blocks = [list(range(r * b_local, (r + 1) * b_local)) for r in range(world)]
    opt.zero_grad()
    p_det = [model.encode_objects(objects[b[0]:b[-1] + 1]).detach().clone() for b in blocks]
    losses = []
    for r, b in enumerate(blocks):
        p_live = model.encode_objects(objects[b[0]:b[-1] + 1])
        t = [model.encode_text(bb) if j == r else model.encode_text(bb).detach() for j, bb in enumerate(blocks)]
        p = [p_live if j == r else p_det[j] for j in range(world)]
        loss_r = crit(torch.cat(t), torch.cat(p))
        loss_r.backward()
        losses.append(float(loss_r.detach()))
    flat_ref = model.train_flat_grad().detach().cpu().numpy().copy()
    table_ref = model.language_encoder.table.grad.detach().cpu().numpy().copy()
    opt.step()
    torch.cuda.synchronize()
    ref_params = _params(model)

    l0, loc0, sum0, tg0, p0 = res[0][1:]
    assert max(losses) - min(losses) < 1e-5
    assert abs(l0 - losses[0]) < 2e-5 * abs(l0)
    local_sum = np.zeros_like(sum0)
    for _, lr_, loc, summ, tg, pr in res:
        assert abs(lr_ - l0) < 1e-6
        assert np.array_equal(summ, sum0)  # every rank holds the same reduced gradient ...
        assert np.array_equal(tg, tg0)
        local_sum = local_sum + loc
    # ... which is the SUM of the local ones (one flat collective; float addition order of the ring is the only freedom beyond 2 ranks)
    if world == 2:
        assert np.allclose(sum0, local_sum, rtol=0, atol=0)
    else:
        assert np.abs(sum0 - local_sum).max() <= 4e-6 * max(1.0, float(np.abs(sum0).max()))
    assert not np.array_equal(res[0][2], res[1][2])
    rms = float(np.sqrt((flat_ref ** 2).mean()))
    assert np.abs(sum0 - flat_ref).max() < 2e-3 * rms, (np.abs(sum0 - flat_ref).max(), rms)  # float atomics order only
    assert np.median(np.abs(sum0 - flat_ref)) < 1e-5 * rms
    assert np.abs(tg0 - table_ref).max() < 1e-6 + 1e-4 * np.abs(table_ref).max()
    for n, v in ref_params.items():
        for _, _, _, _, _, pr in res[1:]:
            assert np.array_equal(p0[n], pr[n]), n  # replicas stay in lock-step
        if (n.startswith("object_encoder.") and n.endswith(".0.bias")) or n.endswith("in_proj_bias"):
            continue  # zero true gradient (a Linear bias in front of a BatchNorm, the attention's key bias): Adam steps on rounding noise
        err = np.abs(p0[n] - v)
        # Adam's first step is lr * g / (|g| + eps): elements whose gradient is rounding noise may move by up to 2 lr
        assert float((err < 1e-5 * (1 + np.abs(v))).mean()) > 0.97 and float(err.max()) <= 2.1 * LR, (n, float(err.max()))


@pytest.mark.parametrize("world,b_local", [(2, 32), (8, 8)])
def test_ranks_with_sync_batchnorm_equal_the_one_process_step_on_the_global_batch(world, b_local):
    """``optim.Adam(data_parallel=True, sync_bn=True)`` + ``ContrastiveLoss(gather=True)``: BatchNorm statistics over every rank's
    objects (t2l_train_sync_bn) — the checker is now ONE process taking ONE step on all world x b_local cells, i.e. the reference's own
    geometry (training/coarse.py:31-58), not an accumulation of per-block passes."""
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, b_local, port, out_q, True)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([out_q.get(timeout=900) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0

    from text2loc_amd.losses import ContrastiveLoss
    from text2loc_amd.optim import Adam

    model, objects = _build(world, b_local)
    opt = Adam(model, lr=LR)
    opt.zero_grad()
    n = world * b_local
    loss = ContrastiveLoss(0.1)(model.encode_text(list(range(n))), model.encode_objects(objects))
    loss.backward()
    flat_ref = model.train_flat_grad().detach().cpu().numpy().copy()
    table_ref = model.language_encoder.table.grad.detach().cpu().numpy().copy()
    opt.step()
    torch.cuda.synchronize()
    ref_params = _params(model)

    l0, _, sum0, tg0, p0 = res[0][1:]
    assert abs(l0 - float(loss.detach())) < 2e-5 * abs(l0)
    for _, lr_, _, summ, tg, pr in res:
        assert abs(lr_ - l0) < 1e-6 and np.array_equal(summ, sum0) and np.array_equal(tg, tg0)
    rms = float(np.sqrt((flat_ref ** 2).mean()))
    # float32 sums over differently partitioned rows (each rank's statistics blocks vs one process's): the tolerances of the
    # reference-golden train-step test (tests/test_gpu_train.py), tensor by tensor
    base = model.train_flat_grad().data_ptr()
    for name, gv in model._train_grads.items():
        o = (gv.data_ptr() - base) // 4
        got, exp = sum0[o:o + gv.numel()].astype(np.float64), flat_ref[o:o + gv.numel()].astype(np.float64)
        err, rms_t = np.abs(got - exp), float(np.sqrt((exp ** 2).mean()))
        if name.startswith("object_encoder.") and name.endswith(".0.bias"):
            assert err.max() < 1e-4, name  # true gradient 0 (a BatchNorm follows); the ranks' local sums cancel to rounding noise
            continue
        if ".num_encoder.0." in name:
            # [64,1] Linear in front of a BatchNorm: scale-invariant (its gradient is the eps / (var + eps) residual of cancelling terms),
            # and with these weights var(w x) is of eps' order, so 1 / sqrt(var + eps) amplifies the float32 rounding of (y - mean) by
            # ~300 into this BatchNorm's own gradients
            assert err.max() < 0.1 * np.abs(exp).max() + 2e-4, (name, err.max())
            continue
        assert (err < 1e-2 * rms_t + 1e-6).mean() >= 0.95 and err.max() < 0.2 * rms_t + 1e-5, (name, float(err.max()), rms_t)
    assert np.median(np.abs(sum0 - flat_ref)) < 1e-4 * rms
    assert np.abs(tg0 - table_ref).max() < 1e-6 + 1e-4 * np.abs(table_ref).max()
    for name, v in ref_params.items():
        for _, _, _, _, _, pr in res[1:]:
            assert np.array_equal(p0[name], pr[name]), name
        if (name.startswith("object_encoder.") and name.endswith(".0.bias")) or name.endswith("in_proj_bias"):
            continue
        err = np.abs(p0[name] - v)
        assert float((err < 1e-5 * (1 + np.abs(v))).mean()) > 0.97 and float(err.max()) <= 2.1 * LR, (name, float(err.max()))
