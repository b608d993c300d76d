"""GPU tier: the PointNet++ object backbone through the C ABI against the build's own CPU restatement
(oracle/t2l_oracle_pointnet.py). PARITY UNPINNED w.r.t. the reference: its arithmetic for this stage lives in absent
third-party packages (see the oracle's header); these tests pin the kernels to the documented deterministic semantics."""
import numpy as np
import pytest
import torch

from oracle import t2l_oracle as O
from oracle import t2l_oracle_pointnet as OP
from text2loc_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[0, 1], ids=["split-f16", "f32"])
def eng(request):
    """Both kernel sets (split-f16 MFMAs with the magnitude watch / everything on the f32 MFMA) against the same bars."""
    from text2loc_amd.engine import Engine

    e = Engine(0)
    e.set_option("encoder_f32", request.param)
    e.all_f32 = request.param
    sd = synth.make_object_branch_weights(0)
    sd.update(synth.make_pointnet_weights(0))
    e.load_weights(sd, class_embed=False, color_embed=False)
    e._sd = sd
    yield e
    e.close()


def run(eng, cells, pos, rgb):
    out = eng.pointnet_features(torch.from_numpy(pos).cuda(), torch.from_numpy(rgb).cuda(), cells["offsets"])
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("self_loops", [True, False])
def test_features2_match_the_restatement(eng, self_loops):
    cells = synth.make_cells(3, seed=1, min_obj=1, max_obj=5)
    pos, rgb = synth.make_sampled_points(cells, 1)
    eng.set_option("pointnet_pyg_self_loops", 1 if self_loops else 0)
    got = run(eng, cells, pos, rgb)
    eng.set_option("pointnet_pyg_self_loops", 1)
    ref = OP.pointnet_features(pos, rgb, cells["offsets"], eng._sd, pyg_self_loops=self_loops)
    assert got.shape == ref.shape == (int(cells["offsets"][-1]), 256)
    assert np.abs(got - ref).max() < 2e-4 * max(1.0, np.abs(ref).max()), np.abs(got - ref).max()
    assert (ref > 0).mean() > 0.2  # not a dead network


def test_magnitude_watch_hands_large_objects_to_the_f32_kernels(eng):
    """Colour values of ~1e5 in SOME objects leave the f16 range inside the split products: those objects are flagged and
    recomputed by the f32 launches (flags are sticky down the levels), the others stay on the split kernels; all match."""
    cells = synth.make_cells(2, seed=9, min_obj=3, max_obj=4)
    pos, rgb = synth.make_sampled_points(cells, 9)
    rgb = rgb.copy()
    rgb[1] *= np.float32(2.0e5)   # beyond 65504 from the first layer on
    rgb[4] *= np.float32(3.0e3)   # fits f16 as an input, overflows deeper in the MLP chain (or not at all): either way exact enough
    got = run(eng, cells, pos, rgb)
    ref = OP.pointnet_features(pos, rgb, cells["offsets"], eng._sd)
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() < 2e-4 * max(1.0, np.abs(ref).max()), np.abs(got - ref).max()


def test_a_batch_of_thousands_of_objects_matches_the_restatement_cell_by_cell(eng):
    """The self-loop tiles and the global MLP run as PERSISTENT workgroups (eight / four tiles per round, the packed-weight stream
    wrapping round after round): with ~3,000 objects every one of them goes through several rounds. Cells are independent (the
    self-loop edges stay inside a cell), so the restatement is run on a few cells taken from the start, the middle and the end of
    the batch — objects that sit in different rounds and workgroups — and on flagged objects in between."""
    cells = synth.make_cells(150, seed=21)
    off = cells["offsets"]
    pos, rgb = synth.make_sampled_points(cells, 21)
    assert int(off[-1]) > 2500
    rgb = rgb.copy()
    big = int(off[75])           # one object in the middle leaves the f16 range: the f32 kernels recompute it (and only it)
    rgb[big] *= np.float32(2.0e5)
    for self_loops in (1, 0):
        eng.set_option("pointnet_pyg_self_loops", self_loops)
        got = run(eng, cells, pos, rgb)
        assert np.isfinite(got).all()
        for c0, c1 in ((0, 2), (74, 77), (148, 150)):
            lo, hi = int(off[c0]), int(off[c1])
            ref = OP.pointnet_features(pos[lo:hi], rgb[lo:hi], off[c0:c1 + 1] - off[c0], eng._sd, pyg_self_loops=bool(self_loops))
            err = np.abs(got[lo:hi] - ref).max()
            assert err < 2e-4 * max(1.0, np.abs(ref).max()), (self_loops, c0, err)
    eng.set_option("pointnet_pyg_self_loops", 1)


def test_degenerate_objects(eng):
    """All points identical (every ball query saturates at 32, FPS ties everywhere) and a tiny cluster (1-neighbour balls)."""
    cells = synth.make_cells(1, seed=3, min_obj=3, max_obj=3)
    pos, rgb = synth.make_sampled_points(cells, 3)
    pos[0] = 0.25
    pos[1] = np.linspace(-1, 1, 256)[:, None].astype(np.float32) * np.array([1, 0, 0], np.float32)  # a line: sparse balls
    got = run(eng, cells, pos, rgb)
    ref = OP.pointnet_features(pos, rgb, cells["offsets"], eng._sd)
    assert np.abs(got - ref).max() < 2e-4 * max(1.0, np.abs(ref).max())


def test_published_mode_cell_embeddings_from_raw_points(eng):
    """a3 -> a2 -> a4 chained on the GPU: points -> features2 -> mlp_pointnet ... -> cell embedding, vs the oracle chain."""
    cells = synth.make_cells(4, seed=5, min_obj=2, max_obj=6)
    pos, rgb = synth.make_sampled_points(cells, 5)
    f2 = eng.pointnet_features(torch.from_numpy(pos).cuda(), torch.from_numpy(rgb).cuda(), cells["offsets"])
    packed = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in cells.items() if k != "counts"}
    packed["pn_feat"] = f2
    emb = eng.encode_cells(packed).cpu().numpy()
    rc = dict(cells)
    rc["pn_feat"] = OP.pointnet_features(pos, rgb, cells["offsets"], eng._sd)
    ref = O.encode_cells(rc, eng._sd, False, False)
    assert np.abs(emb - ref).max() < 1e-4


def test_needs_pointnet_weights():
    from text2loc_amd.engine import Engine, T2LError

    e = Engine(0)
    e.load_weights(synth.make_object_branch_weights(0), class_embed=True, color_embed=True)
    with pytest.raises(T2LError, match="pointnet"):
        e.pointnet_features(torch.zeros(1, 256, 3, device="cuda"), torch.zeros(1, 256, 3, device="cuda"), np.array([0, 1]))
    e.close()


def test_drop_in_published_mode_from_point_batches():
    """CellRetrievalNetwork(class_embed off).encode_objects(objects, point batches): Object3d lists + the dataloader's
    per-cell point batches in, cell embeddings out — PointNet++, object encoder and set transformer all in the engine."""
    import argparse

    from tests.test_host_logic import make_objects
    from text2loc_amd import packing
    from text2loc_amd.cell_retrieval import CellRetrievalNetwork

    class Txt(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))

        @property
        def device(self):
            return self.p.device

    args = argparse.Namespace(coarse_embed_dim=256, object_size=28, object_inter_module_num_heads=4,
                              object_inter_module_num_layers=2, class_embed=False, color_embed=False,
                              use_features=["class", "color", "position", "num"], pointnet_freeze=True)
    cells = synth.make_cells(3, seed=9, min_obj=2, max_obj=5)
    objects = make_objects(cells, 9)
    model = CellRetrievalNetwork(synth.KNOWN_CLASS, synth.COLOR_NAMES, args, language_encoder=Txt())
    sd = synth.make_object_branch_weights(2)
    sd.update(synth.make_pointnet_weights(2, n_classes=len(synth.KNOWN_CLASS), n_colors=len(synth.COLOR_NAMES)))
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=False)
    model = model.to("cuda").eval()
    batches = packing.sample_object_points(objects, 256, np.random.default_rng(0), transform="normalize")
    emb = model.encode_objects(objects, batches).cpu().numpy()
    # the same chain through the restatements
    pos = np.concatenate([b["pos"].reshape(-1, 256, 3) for b in batches])
    rgb = np.concatenate([b["x"].reshape(-1, 256, 3) for b in batches])
    for p in pos:  # NormalizeScale: centred, inside the unit cube
        assert np.abs(p.mean(0)).max() < 1e-5 and 0.99 < np.abs(p).max() <= 1.0
    packed = packing.pack_cells(objects, model.object_encoder.known_classes, model.object_encoder.known_colors)
    packed["pn_feat"] = OP.pointnet_features(pos, rgb, packed["offsets"], sd)
    ref = O.encode_cells(packed, sd, False, False)
    assert np.abs(emb - ref).max() < 1e-4
    # point batches sampled on the GPU feed the same path (different draw, so compare with the restatement on those batches)
    gb = packing.sample_object_points_gpu(model.engine(), objects, "cuda", seed=5)
    emb_g = model.encode_objects(objects, gb).cpu().numpy()
    pos_g = np.concatenate([b["pos"].cpu().numpy().reshape(-1, 256, 3) for b in gb])
    rgb_g = np.concatenate([b["x"].cpu().numpy().reshape(-1, 256, 3) for b in gb])
    packed_g = dict(packed)
    packed_g["pn_feat"] = OP.pointnet_features(pos_g, rgb_g, packed["offsets"], sd)
    assert np.abs(emb_g - O.encode_cells(packed_g, sd, False, False)).max() < 1e-4
    # precomputed features2 remain accepted
    feats = [torch.from_numpy(packed["pn_feat"][packed["offsets"][i]:packed["offsets"][i + 1]]) for i in range(3)]
    emb2 = model.encode_objects(objects, feats).cpu().numpy()
    assert np.abs(emb2 - ref).max() < 1e-4


def test_argument_errors(eng):
    from text2loc_amd.engine import T2LError

    z = torch.zeros(2, 256, 3, device="cuda")
    with pytest.raises(T2LError, match="offsets ending at n"):
        eng.pointnet_features(z, z, np.array([0, 3]))
    with pytest.raises(T2LError, match=r"\[n,256,3\]"):
        eng.pointnet_features(torch.zeros(2, 128, 3, device="cuda"), z, np.array([0, 2]))
    with pytest.raises(T2LError, match="non-decreasing"):
        eng.pointnet_features(z, z, np.array([0, 3, 2]))
    with pytest.raises(T2LError, match="CUDA"):
        eng.pointnet_features(z.cpu(), z, np.array([0, 2]))


@pytest.mark.parametrize("transform", ["fixed", "normalize", "rotate_normalize"])
def test_point_batches_sampled_on_the_gpu(eng, transform):
    """FixedPoints(256) [+ RandomRotate(120, z)] [+ NormalizeScale] on the GPU vs the numpy restatement (same counter-based
    draws), then straight into PointNet++. "fixed" = `--no_pc_augment`, the published configuration."""
    rs = np.random.default_rng(4)
    n_pts = np.array([8, 25, 300, 4000, 61], dtype=np.int64)
    poff = np.concatenate([[0], np.cumsum(n_pts)]).astype(np.int64)
    centre = rs.uniform(0.2, 0.8, size=(5, 3)).repeat(n_pts, axis=0)
    xyz = (centre + rs.standard_normal((int(poff[-1]), 3)) * 0.08).astype(np.float32)  # cell-frame objects
    rgb = rs.uniform(0, 1, size=(int(poff[-1]), 3)).astype(np.float32)
    args = (torch.from_numpy(xyz).cuda(), torch.from_numpy(rgb).cuda(), torch.from_numpy(poff).cuda())
    pos, col = eng.sample_object_points(*args, seed=77, transform=transform)
    rpos, rcol = OP.sample_object_points(xyz, rgb, poff, 77, transform=transform)
    assert np.array_equal(col.cpu().numpy(), rcol)  # same indices
    p = pos.cpu().numpy()
    if transform == "fixed":
        assert np.array_equal(p, rpos)  # raw points, bit for bit
        for o in range(5):
            raw = xyz[poff[o]:poff[o + 1]]
            assert all((raw == q).all(axis=1).any() for q in p[o, :16])
    else:
        assert np.abs(p - rpos).max() < 4e-6
        assert np.abs(p.mean(axis=1)).max() < 1e-5 and np.all(np.abs(p).max(axis=(1, 2)) <= 1.0) and np.all(np.abs(p).max(axis=(1, 2)) > 0.9999)
    if transform == "rotate_normalize":
        pn, _ = eng.sample_object_points(*args, seed=77, transform="normalize")
        assert not torch.equal(pos, pn)
        # a rotation about z: z differs from the un-rotated batch by the normalisation scale only
        zr, zn = p[..., 2], pn.cpu().numpy()[..., 2]
        ratio = (zr / np.where(np.abs(zn) > 1e-3, zn, np.nan))
        assert np.nanstd(ratio, axis=1).max() < 1e-3
    pos2, _ = eng.sample_object_points(*args, seed=78, transform=transform)
    assert not torch.equal(pos, pos2)
    f2 = eng.pointnet_features(pos, col, np.array([0, 2, 5]))
    ref = OP.pointnet_features(rpos, rcol, np.array([0, 2, 5]), eng._sd)
    assert np.abs(f2.cpu().numpy() - ref).max() < 2e-4 * max(1.0, np.abs(ref).max())
    with pytest.raises(Exception, match="transform"):
        eng.sample_object_points(*args, seed=1, transform="bogus")


def test_raw_cell_frame_coordinates_through_the_backbone(eng):
    """`--no_pc_augment`: the backbone eats cell-normalised coordinates as they are (objects a few percent of the cell wide,
    anywhere in [0,1]^3), so most ball queries hold MORE than 32 in-radius points (first 32 in index order) and the
    relative positions are small against the absolute ones. Also a cell-sized object (a building facade) in the same batch."""
    cells = synth.make_cells(3, seed=21, min_obj=2, max_obj=5)
    total = int(cells["offsets"][-1])
    rs = np.random.default_rng(21)
    centre = rs.uniform(0.05, 0.95, size=(total, 1, 3))
    extent = rs.uniform(0.01, 0.12, size=(total, 1, 3))
    pos = (centre + rs.standard_normal((total, 256, 3)) * extent).astype(np.float32)
    pos[0] = rs.uniform(0, 1, size=(256, 3)).astype(np.float32) * np.array([1, 0.05, 0.6], np.float32)  # facade across the cell
    rgb = np.clip(rs.uniform(0, 1, size=(total, 1, 3)) + 0.1 * rs.standard_normal((total, 256, 3)), 0, 1).astype(np.float32)
    d = np.linalg.norm(pos[1][:, None] - pos[1][None], axis=-1)
    assert (d < 0.2).sum(axis=1).max() > 32  # the saturating regime really occurs
    got = run(eng, cells, pos, rgb)
    ref = OP.pointnet_features(pos, rgb, cells["offsets"], eng._sd)
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() < 2e-4 * max(1.0, np.abs(ref).max()), np.abs(got - ref).max()


def test_plain_f16_option_of_the_backbone(eng):
    """Option encoder_f16 on t2l_pointnet_features: one f16 product per operand pair in the edge MLPs and the global MLP
    (11.0 -> 5.9 ms for 10.7 k objects in round 5); features2 within 1e-3 of the restatement's scale, the magnitude watch unchanged."""
    cells = synth.make_cells(4, seed=12, min_obj=2, max_obj=6)
    pos, rgb = synth.make_sampled_points(cells, 12)
    ref = OP.pointnet_features(pos, rgb, cells["offsets"], eng._sd)
    base = run(eng, cells, pos, rgb)
    eng.set_option("encoder_f16", 1)
    try:
        got = run(eng, cells, pos, rgb)
        rgb2 = rgb.copy()
        rgb2[0] *= np.float32(2.0e5)  # leaves the f16 range: that object goes to the f32 kernels as before
        got2 = run(eng, cells, pos, rgb2)
    finally:
        eng.set_option("encoder_f16", 0)
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.abs(base - ref).max() < 2e-4 * scale
    assert np.abs(got - ref).max() < 1e-3 * scale
    assert eng.all_f32 or not np.array_equal(got, base)  # really the other kernel (the all-f32 set ignores the option)
    assert np.isfinite(got2).all()
