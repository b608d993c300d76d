"""GPU tier: a1 — per-object reductions over raw points (t2l_reduce_objects) vs the oracle / reference goldens.
The reference sums rgb in float32 and xyz in float64 with numpy; the kernel accumulates in float64 and rounds once.
Tolerance: 2e-5 absolute on means of values in [0,1] (the float32 running sums of the REFERENCE over up to 60,000 points
carry ~sqrt(n)*2^-24 of error; the kernel's float64 sums are the more accurate side); colour indices
must agree except where the mean sits within that tolerance of a decision boundary (none in the fixtures)."""
import numpy as np
import pytest

from oracle import t2l_oracle as O
from text2loc_amd import packing, synth
from tests.test_host_logic import make_objects

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from text2loc_amd.engine import Engine

    e = Engine(0)
    yield e
    e.close()


def test_reductions_vs_reference_golden(eng, golden):
    g = golden("objects_reduce")
    cells = synth.make_cells(int(g["n_cells"]), seed=int(g["cell_seed"]))
    objs = make_objects(cells, int(g["cell_seed"]))[:3]
    p = packing.pack_cells_gpu(eng, objs, packing.class_table(synth.KNOWN_CLASS))
    n = int(g["n_objects"])
    assert np.abs(p["rgb"].cpu().numpy() - g["color_rgb"][:n]).max() < 2e-5
    assert np.abs(p["center"].cpu().numpy() - g["center"][:n]).max() < 2e-5
    assert np.array_equal(p["n_pts"].cpu().numpy(), g["n_pts"][:n].astype(np.float32))
    assert np.array_equal(p["color_idx"].cpu().numpy(), g["color_embed_index"][:n])
    assert np.array_equal(p["class_idx"].cpu().numpy(), g["class_index"][:n])


def test_gpu_packer_equals_host_packer_and_feeds_the_encoder(eng):
    cells = synth.make_cells(40, seed=21)
    objs = make_objects(cells, 21)
    host = packing.pack_cells(objs, packing.class_table(synth.KNOWN_CLASS))
    dev = packing.pack_cells_gpu(eng, objs, packing.class_table(synth.KNOWN_CLASS))
    for k in ("offsets", "class_idx", "color_idx", "n_pts"):
        assert np.array_equal(dev[k].cpu().numpy(), host[k]), k
    for k in ("rgb", "center"):
        assert np.abs(dev[k].cpu().numpy() - host[k]).max() < 2e-5, k
    sd = synth.make_object_branch_weights(0)
    eng.load_weights(sd, class_embed=True, color_embed=False)
    a = eng.encode_cells(dev).cpu().numpy()
    ref = O.encode_cells(host, sd, True, False)
    assert np.abs(a - ref).max() < 1e-4  # well inside the 1e-3 bar; inputs differ by float32 round-off


def test_edge_objects(eng):
    class Obj:
        def __init__(self, n, seed):
            r = np.random.default_rng(seed)
            self.label, self.xyz, self.rgb = "pole", r.uniform(0, 1, (n, 3)), r.uniform(0, 1, (n, 3)).astype(np.float32)

    objs = [[Obj(1, 0), Obj(63, 1), Obj(64, 2), Obj(65, 3), Obj(60000, 4)]]
    p = packing.pack_cells_gpu(eng, objs, packing.class_table(synth.KNOWN_CLASS))
    for i, o in enumerate(objs[0]):
        crgb, cidx, center, n = O.object_reductions(o.xyz.astype(np.float32), o.rgb, synth.COLORS)
        assert np.abs(p["rgb"][i].cpu().numpy() - crgb).max() < 3e-5  # numpy's float32 running sum over 60k points
        assert np.abs(p["center"][i].cpu().numpy() - center).max() < 3e-5
        assert p["n_pts"][i].item() == n
        assert p["color_idx"][i].item() == synth.color_name_to_embed_index(cidx)
