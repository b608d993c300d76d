"""GPU tier, SURVEY.md 8e / a9: data-parallel training that takes THE REFERENCE'S step. The reference trains one process on the
whole batch (training/coarse.py:31-58): BatchNorm1d normalises over all objects of the 64 cells. With cross-rank BatchNorm
statistics (t2l_train_sync_bn: the per-channel sums of every BatchNorm stage are added up over the ranks between the statistics
and the apply launch) W ranks x 64/W cells reproduce the goldens of the reference's own B = 64 train step
(tests/golden/train_step_{embed,pn}.npz): forward rows, loss, the summed parameter gradients, the running statistics.
Processes share GPU 0 under gloo (RCCL refuses duplicate devices; the sums are staged through the host)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _slice_cells(cells, lo, hi):
    o = np.asarray(cells["offsets"])
    a, b = int(o[lo]), int(o[hi])
    out = {"offsets": (o[lo:hi + 1] - a).astype(np.int32)}
    for k, v in cells.items():
        if k not in ("offsets", "counts"):
            out[k] = np.ascontiguousarray(np.asarray(v)[a:b])
    return out


def _worker(rank, world, port, mode, sync, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.test_gpu_train import bind, to_dev
    from tests.test_oracle_train import load_case
    from text2loc_amd.engine import Engine

    g = np.load(os.path.join(GOLDEN, f"train_step_{mode}.npz"), allow_pickle=True)
    cells, sd, embed = load_case(g, mode)
    B = int(g["n_cells"])
    lo, hi = rank * B // world, (rank + 1) * B // world
    eng = Engine(0)
    tensors = bind(eng, sd, embed)
    if sync:
        eng.train_sync_bn()
    mine = eng.encode_cells_train(to_dev(_slice_cells(cells, lo, hi), embed), dropout_p=0.0, seed=0)
    rows = torch.zeros(B, 256, device="cuda")
    rows[lo:hi] = mine
    h = rows.cpu()
    dist.all_reduce(h)  # (disjoint rows: the sum is the gather)
    positive = h.cuda()
    loss, _, gp = eng.contrastive_loss(torch.from_numpy(g["anchor"]).cuda(), positive, float(g["temperature"]))
    eng.encode_cells_backward(gp[lo:hi].contiguous())
    torch.cuda.synchronize()
    grads = {}
    for n in [str(x) for x in g["used_params"]]:
        t = tensors[n][1].cpu()
        dist.all_reduce(t)  # what optim.Adam(data_parallel=True) does on the flat buffer
        grads[n] = t.numpy()
    bufs = {k: v[0].cpu().numpy() for k, v in tensors.items() if "running_" in k}
    out_q.put((rank, h.numpy(), float(loss), grads, bufs, eng.sync_bn_calls()))
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


def _run(world, mode, sync):
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, sync, out_q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([out_q.get(timeout=900) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("world,mode", [(2, "embed"), (2, "pn"), (8, "embed")])
def test_ranks_with_cross_rank_batchnorm_take_the_reference_step(world, mode):
    from tests.test_oracle_train import golden_view

    g = np.load(os.path.join(GOLDEN, f"train_step_{mode}.npz"), allow_pickle=True)
    res = _run(world, mode, True)
    for rank, positive, loss, grads, bufs, calls in res:
        # BatchNorm stages per direction: the small branches' two layers (one launch each for all branches), mlp_merge, and in "pn"
        # mode mlp_pointnet
        assert calls == (6 if mode == "embed" else 8)
        assert np.abs(positive - g["positive"]).max() < 2e-5  # the same bound as the one-process test (test_gpu_train.py)
        assert abs(loss - float(g["loss"])) < 2e-5
        for n in [str(x) for x in g["used_params"]]:
            exp, got = golden_view(g, "grad", n, grads[n])
            rms = float(g[f"grad_norm/{n}"]) / np.sqrt(grads[n].size)
            if n.startswith("object_encoder.") and n.endswith(".0.bias"):
                assert np.abs(got).max() < 1e-4 and np.abs(exp).max() < 1e-4, n  # true gradient 0 (BatchNorm follows)
                continue
            err = np.abs(got - exp)
            assert (err < 1e-2 * rms + 1e-5).mean() >= 0.95 and err.max() < 0.2 * rms + 1e-5, (n, err.max(), rms)
            nrm = float(np.sqrt((grads[n].astype(np.float64) ** 2).sum()))
            assert abs(nrm - float(g[f"grad_norm/{n}"])) < 2e-3 * float(g[f"grad_norm/{n}"]) + 2e-4, n
        for k in g.files:
            if k.startswith("buf/") and "running_" in k and k[4:] in bufs:
                assert np.allclose(bufs[k[4:]], g[k], rtol=2e-4, atol=2e-5), k
    for r in res[1:]:  # every rank holds the same summed gradients and running statistics
        for n, v in res[0][3].items():
            assert np.array_equal(v, r[3][n]), n
        for n, v in res[0][4].items():
            assert np.array_equal(v, r[4][n]), n


def test_per_rank_statistics_do_not_take_the_reference_step():
    """The control: without the exchange two ranks x 32 cells normalise over their own objects — a different (legitimate) model
    of the step, measurably not the reference's."""
    g = np.load(os.path.join(GOLDEN, "train_step_embed.npz"), allow_pickle=True)
    res = _run(2, "embed", False)
    assert res[0][5] == 0
    assert np.abs(res[0][1] - g["positive"]).max() > 1e-4


def _text_worker(rank, world, port, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.test_gpu_text_train import _encoder, _no_dropout
    from text2loc_amd import synth
    from text2loc_amd.cell_retrieval import LanguageEncoder
    from text2loc_amd.losses import ContrastiveLoss

    g = np.load(os.path.join(GOLDEN, "train_step_text.npz"), allow_pickle=True)
    B, S, L = int(g["batch"]), int(g["n_hints"]), int(g["n_tokens"])
    lo, hi = rank * B // world, (rank + 1) * B // world
    enc = _encoder(int(g["weight_seed"]))
    _no_dropout(enc)
    enc.train()
    enc._sync_bn_cfg = (None,)  # what CellRetrievalNetwork.sync_batchnorm() sets on its language encoder
    hidden = torch.from_numpy(synth.make_t5_hidden(B * S, L, seed=int(g["hidden_seed"]))).cuda()
    n0 = LanguageEncoder.train_engine_calls
    mine = enc.head(hidden[lo * S:hi * S].contiguous(), hi - lo)
    assert LanguageEncoder.train_engine_calls == n0 + 1
    rows = torch.zeros(B, 256)
    rows[lo:hi] = mine.detach().cpu()
    dist.all_reduce(rows)
    full = rows.cuda()
    full[lo:hi] = mine  # this rank differentiates its own rows of the global loss
    loss = ContrastiveLoss(float(g["temperature"]))(torch.nn.functional.normalize(full), torch.from_numpy(g["cells"]).cuda())
    loss.backward()
    torch.cuda.synchronize()
    grads = {}
    for n, p in enc.named_parameters():
        if p.grad is not None:
            t = p.grad.cpu()
            dist.all_reduce(t)
            grads["language_encoder." + n] = t.numpy()
    bufs = {"language_encoder." + n: b.cpu().numpy() for n, b in enc.named_buffers()}
    out_q.put((rank, rows.numpy(), float(loss.detach()), grads, bufs, enc._th_train_engine.sync_bn_calls()))
    dist.barrier()
    dist.destroy_process_group()


def test_text_head_ranks_with_cross_rank_batchnorm_take_the_reference_step():
    """inter_mlp's BatchNorm1d (language_encoder.py:99) over the sentences of BOTH ranks: 2 x 4 descriptions against the reference's own
    B = 8 step (tests/golden/train_step_text.npz) — head output, loss, summed gradients, running statistics."""
    from tests.test_oracle_train import golden_view

    g = np.load(os.path.join(GOLDEN, "train_step_text.npz"), allow_pickle=True)
    world = 2
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_text_worker, args=(r, world, port, out_q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([out_q.get(timeout=900) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    P = "language_encoder."
    for rank, out, loss, grads, bufs, calls in res:
        assert calls == 2
        assert np.abs(out - g["head_out"]).max() < 5e-5 * max(1.0, np.abs(g["head_out"]).max())
        assert abs(loss - float(g["loss"])) < 2e-5 * abs(float(g["loss"]))
        for n in [str(x) for x in g["used_params"]]:
            exp, got = golden_view(g, "grad", n, grads[n])
            rms = float(g[f"grad_norm/{n}"]) / np.sqrt(max(grads[n].size, 1))
            if n.endswith(("inter_mlp.0.0.bias", "intra_module.0.norm2.bias")):
                assert np.abs(got).max() < 1e-4
                continue
            if n.endswith("in_proj_bias") and len(got) <= 1024:
                D = len(got) // 3
                sel = np.r_[0:D, 2 * D:3 * D]
                exp, got = exp[sel], got[sel]
            err = np.abs(got - exp)
            assert (err < 1e-2 * rms + 1e-6).mean() >= 0.95 and err.max() < 0.2 * rms + 1e-5, (n, float(err.max()), rms)
        for k in g.files:
            if k.startswith("buf/") and "num_batches_tracked" not in k:
                assert np.allclose(bufs[k[4:]], g[k], rtol=2e-4, atol=2e-5), k
