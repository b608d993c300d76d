"""GPU tier: the drop-in surface at evaluation scale — ``run_coarse(model, dataloader, args)`` (evaluation/pipeline.py:41-87 ->
training/coarse.py:63-157) over a KITTI360Pose-shaped dataset, with the engine-sized batching of ``coarse.eval_epoch``:

* the DATABASE side is re-batched freely (a cell's eval-mode embedding depends on that cell alone): bit-identical rows whatever
  ``args.batch_size`` / ``engine_batching`` say;
* the QUERY side keeps the reference's batching (padding length L of the batch's longest sentence, no padding mask): the one-pass
  ``encode_text_batches`` equals the per-batch loop (same memo rows; <= 2e-6 from the inter-sentence kernel's tile placement),
  and a different ``batch_size`` really gives different vectors;
* at BASELINE config 2's size (11,259 cells, 4,096 poses) in the PUBLISHED feature mode (PointNet++ in the engine): every
  retrieved id equals the float64 ranking of the embeddings, a sample of cells equals the oracle chain, accuracies equal the
  oracle's bookkeeping.
"""
import numpy as np
import pytest
import torch

from text2loc_amd import synth

pytestmark = pytest.mark.gpu


def _dataset(n_cells, n_poses, seed, points):
    from text2loc_amd.kitti360pose import Kitti360PoseDataset

    cells, poses = synth.make_k360_records(n_cells, n_poses, seed=seed, pts_per_obj=(25, 40))
    return Kitti360PoseDataset.from_records(cells, poses, object_points="sample" if points else None, seed=seed)


def _loader(ds, bs):
    return torch.utils.data.DataLoader(ds, batch_size=bs, collate_fn=ds.collate_fn, shuffle=False)


@pytest.mark.parametrize("published", [False, True])
def test_eval_epoch_is_independent_of_batching(published):
    from text2loc_amd.coarse import eval_epoch
    from text2loc_amd.text_cache import TextCache

    ds = _dataset(150, 96, seed=3, points=published)
    runs = {}
    for bs, engine_batching in ((1, True), (7, True), (64, True), (7, False), (1, False)):
        args = synth.coarse_args(class_embed=not published, color_embed=not published, batch_size=bs, top_k=[1, 3, 5],
                                 engine_batching=engine_batching, engine_batch_cells=64)
        model = synth.make_coarse_model(args, sentences=TextCache.sentences_of(ds), seed=1)
        acc, close, retr, ce, te = eval_epoch(model, _loader(ds, bs), args, return_encodings=True)
        runs[(bs, engine_batching)] = (acc, close, np.array([retr[q] for q in range(len(retr))]), ce, te)
    base = runs[(7, False)]  # the reference's own loop shape: one encode call per args.batch_size items
    # text: the one-pass form == the per-batch loop at the same batch size, up to the inter-sentence kernel's round-off (a
    # description's place in a 32-row tile changes the order of its softmax / LayerNorm sums: <= 1e-6 on unit vectors; the
    # per-sentence vectors — where the batch's padding length L enters — are the same memo rows either way)
    assert np.abs(runs[(7, True)][4] - base[4]).max() < 2e-6
    assert np.abs(runs[(1, True)][4] - runs[(1, False)][4]).max() < 2e-6
    # ... and the batch size IS part of the text result (padding to the batch's longest sentence, language_encoder.py:113-131)
    assert np.abs(runs[(1, True)][4] - base[4]).max() > 1e-3
    if not published:
        # database side: identical rows under every batching (host-packed per call vs flattened once + GPU reductions differ only in
        # the a1 reductions' arithmetic: numpy float64 means vs the kernel's — compare the engine-batched runs among themselves
        # bit for bit and against the per-call runs to float32 round-off)
        for key in ((1, True), (64, True)):
            assert np.array_equal(runs[key][3], runs[(7, True)][3]), key
        assert np.abs(runs[(7, True)][3] - base[3]).max() < 2e-6
        assert np.array_equal(runs[(1, False)][3], base[3])
        assert np.array_equal(runs[(7, True)][2], base[2]) or np.abs(runs[(7, True)][3] - base[3]).max() > 0
    else:
        # published mode: the point batches are DRAWN (FixedPoints), host draws per item vs the engine's counter-based draws per
        # chunk: the engine-batched runs share their draws (same chunking) and must agree bit for bit
        for key in ((1, True), (64, True)):
            assert np.array_equal(runs[key][3], runs[(7, True)][3]), key
        assert np.array_equal(runs[(1, False)][3], base[3])  # host draws are keyed on (seed, cell): batch-size independent too
    for key, (acc, close, ids, ce, te) in runs.items():
        assert ids.shape == (96, 5) and set(acc) == {1, 3, 5}


def test_forward_batches_equals_the_per_batch_loop_and_learns_new_descriptions():
    from text2loc_amd.text_cache import TextCache

    ds = _dataset(40, 50, seed=5, points=False)
    args = synth.coarse_args(class_embed=True, color_embed=True)
    model = synth.make_coarse_model(args, sentences=TextCache.sentences_of(ds), seed=2)
    texts = ds.eval_texts()
    le = model.language_encoder
    with torch.no_grad():
        for bs in (1, 6, 50, 64):
            loop = torch.cat([model.encode_text(texts[i:i + bs]) for i in range(0, len(texts), bs)])
            le.text_cache._desc.clear()  # first sight: forward_batches splits and remembers them itself
            one = model.encode_text_batches(texts, bs)
            again = model.encode_text_batches(texts, bs)
            assert torch.equal(one, again) and float((one - loop).abs().max()) < 2e-6, bs
            if bs >= len(texts):
                assert torch.equal(one, loop)  # one batch: the same launch
        # a sentence the cache cannot serve (no T5 behind this cache): the loop path raises like encode_text does
        with pytest.raises(Exception):
            model.encode_text_batches(texts[:3] + ["The pose is north of a purple spaceship."], 2)


def test_run_coarse_at_config2_size_in_the_published_mode():
    """11,259 cells (231,674 objects x 256 sampled points through PointNet++) x 4,096 poses, batch_size = 1 (evaluation/args.py:11)."""
    from oracle import c_oracle
    from oracle import t2l_oracle as O
    from oracle import t2l_oracle_pointnet as OP
    from text2loc_amd.coarse import eval_epoch, run_coarse
    from text2loc_amd.text_cache import TextCache

    n_cells, n_poses = 11259, 4096
    ds = _dataset(n_cells, n_poses, seed=0, points=True)
    args = synth.coarse_args(batch_size=1)
    model = synth.make_coarse_model(args, sentences=TextCache.sentences_of(ds), seed=0)
    dl = _loader(ds, 1)
    acc, close, retr, ce, te, dists, scores = eval_epoch(model, dl, args, return_distance=True)
    assert ce.shape == (n_cells, 256) and te.shape == (n_poses, 256) and np.isfinite(ce).all() and np.isfinite(te).all()
    assert np.abs(np.linalg.norm(ce, axis=1) - 1).max() < 1e-5
    # (1) every retrieved id = the float64 ranking of the engine's embeddings (integer-exact), scores to 1e-12
    ids = np.array([c.id for c in ds.all_cells])
    k = max(args.top_k)
    ridx, rsc = c_oracle.retrieve_topk(ce.astype(np.float32), te.astype(np.float32), k)
    got = np.array([retr[q] for q in range(n_poses)])
    assert np.array_equal(got, ids[ridx])
    assert np.abs(scores - rsc).max() < 1e-12
    # (2) bookkeeping == the oracle's restatement of training/coarse.py:127-150
    centers = np.array([c.get_center()[0:2] for c in ds.all_cells])
    poses_xy = np.array([p.pose_w[0:2] for p in ds.all_poses])
    qids = np.array([p.cell_id for p in ds.all_poses])
    racc, rclose = O.eval_accuracies(ridx, ids, qids, poses_xy, centers, ds.all_cells[0].cell_size, args.top_k)
    assert acc == racc and close == rclose
    # (3) a sample of cells through the oracle chain: the chunk's sampled points (the engine's draws, restated in the oracle:
    # t2l_oracle_pointnet.sample_object_points) -> PointNet++ -> encoder, at the tolerance of tests/test_gpu_pointnet.py
    cs = ds.get_cell_dataset().packed()
    sd = {k2: v.detach().cpu().numpy() for k2, v in model.state_dict().items()}
    chunk = int(getattr(args, "engine_batch_cells", 4096))
    rng = np.random.default_rng(0)
    for cell in rng.choice(n_cells, size=3, replace=False):
        ci = int(cell) // chunk
        lo = ci * chunk
        o_lo = int(cs.offsets[lo])
        a, b = int(cs.offsets[cell]) - o_lo, int(cs.offsets[cell + 1]) - o_lo
        po = cs.point_offsets[o_lo:int(cs.offsets[min(n_cells, lo + chunk)]) + 1]
        seed = (0 + 0x9E3779B1 * ci) & 0xFFFFFFFF
        # the oracle's sampler keys its draws on the object's index inside the call: restate the chunk call for objects [a, b) only
        pos = np.zeros((b - a, 256, 3), np.float32)
        col = np.zeros((b - a, 256, 3), np.float32)
        j = np.arange(256, dtype=np.uint64)
        for o in range(a, b):
            p0, n = int(po[o]), int(po[o + 1] - po[o])
            okey = np.uint64((seed ^ ((o * 0x85EBCA77) & 0xFFFFFFFF)) & 0xFFFFFFFF)
            x = OP._lowbias32(j * np.uint64(0x9E3779B1) + okey)
            idx = (((x >> np.uint64(8)) * np.uint64(n)) >> np.uint64(24)).astype(np.int64)
            pos[o - a], col[o - a] = cs.xyz[p0 + idx], cs.rgb[p0 + idx]
        f2 = OP.pointnet_features(pos, col, np.array([0, b - a]), sd)
        objs = ds.all_cells[cell].objects
        one = {"counts": np.array([b - a], np.int32), "offsets": np.array([0, b - a], np.int32),
               "class_idx": np.zeros(b - a, np.int32), "color_idx": np.zeros(b - a, np.int32),
               "rgb": np.array([np.mean(o.rgb, axis=0) for o in objs], np.float32),
               "center": np.array([np.mean(o.xyz, axis=0) for o in objs], np.float32),
               "n_pts": np.array([len(o.xyz) for o in objs], np.float32), "pn_feat": f2.astype(np.float32)}
        ref = O.encode_cells(one, sd, False, False)
        assert np.abs(ce[cell] - ref[0]).max() < 2e-4, (cell, np.abs(ce[cell] - ref[0]).max())
    # (4) run_coarse's result contract on top of it (second pass over the same dataset: the flattened set is reused)
    retrievals, at = run_coarse(model, dl, args)
    assert len(retrievals) == n_poses and all(np.array_equal(retrievals[q], got[q]) for q in range(0, n_poses, 97))
    pose_scene = np.array([p.cell_id.split("_")[0] for p in ds.all_poses])
    cell_scene = np.array([c.id.split("_")[0] for c in ds.all_cells])
    bbox = np.array([c.bbox_w[0:2] for c in ds.all_cells])
    rat = O.coarse_pose_accuracies(ridx, poses_xy, pose_scene, bbox, cell_scene, ds.all_cells[0].cell_size, args.top_k, args.threshs)
    assert at == rat


def test_arithmetic_ab_tool_on_a_small_dataset():
    """tools/arith_ab.py (bench.py's `arithmetic_ab` secondary) end to end at a size that takes seconds: the coarse
    model trains on a synthetic dataset, the validation poses are searched with the database built in split-f16 and in plain f16 —
    the exact run repeats bit for bit, the plain-f16 embeddings stay within the north star's 1e-3, and the report has its fields."""
    import importlib.util
    import os.path as osp

    spec = importlib.util.spec_from_file_location("arith_ab", osp.join(osp.dirname(osp.dirname(osp.abspath(__file__))), "tools", "arith_ab.py"))
    ab = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ab)
    r = ab.measure(published=False, n_cells=512, n_train=512, n_eval=256, epochs=2)
    assert r["exact_run_repeats_bit_for_bit"]
    assert len(r["train_losses"]) == 2 and r["train_losses"][1] < r["train_losses"][0]
    for name in ("db_f16_vs_exact", "all_f16_vs_exact"):
        d = r[name]
        assert 0.0 < d["max_abs_cell_embedding_diff"] < 1e-3, d
        assert d["same_list_top1"] >= 0.95 and 0.0 <= d["same_list_top10"] <= 1.0
    assert set(r["recall"]) == {"exact", "db_f16", "all_f16"}
