"""TEST INFRASTRUCTURE — build-container only. Never imported by the product, tests, bench or smoke.

Makes the upstream Text2Loc reference (read-only at /root/reference) importable in THIS container so
that ``oracle/gen_golden.py`` can run it and dump golden vectors into ``tests/golden/``.
The reference itself never travels (no source, no bytecode): only the vectors do.

What is shimmed (import surface only — nothing on the pinned arithmetic path):
easydict, cv2, nltk.sent_tokenize, torch_geometric.{nn,transforms,data}; two private numpy modules
and two numpy aliases that numpy 2.x removed. PointNet++ arithmetic (torch_geometric / torch_cluster /
torch_scatter wheels, requirements.txt:15-18) is NOT available => parity unpinned for that stage; the
published-mode goldens are taken downstream of it by substituting fixed ``features2`` tables.
"""
from __future__ import annotations

import argparse
import os
import os.path as osp
import pickle
import sys
import types

import numpy as np

REFERENCE = "/root/reference"
HERE = osp.dirname(osp.abspath(__file__))
REPO = osp.dirname(HERE)


def setup_reference_imports():
    if not osp.isdir(REFERENCE):
        raise RuntimeError("reference checkout not present: goldens can only be generated in the build container")
    for p in (REFERENCE, osp.join(HERE, "refshim"), REPO):
        if p in sys.path:
            sys.path.remove(p)
    sys.path.insert(0, REPO)
    sys.path.insert(0, REFERENCE)
    sys.path.insert(0, osp.join(HERE, "refshim"))
    # numpy 2.x removed these private modules / aliases the reference imports
    if "numpy.lib.arraysetops" not in sys.modules:
        m = types.ModuleType("numpy.lib.arraysetops")
        m.isin = np.isin
        sys.modules["numpy.lib.arraysetops"] = m
    if "numpy.lib.function_base" not in sys.modules:
        m = types.ModuleType("numpy.lib.function_base")
        m.flip = np.flip
        sys.modules["numpy.lib.function_base"] = m
    if not hasattr(np, "int0"):
        np.int0 = np.intp
    if not hasattr(np, "float"):
        np.float = float


TEMPLATE_WORDS = ["the", "pose", "is", "of", "a", "on-top", "north", "south", "east", "west", "The", "."]


def make_tiny_t5(dirname: str, seed: int = 0):
    """A local HF dir with a random 1-layer T5 encoder (d_model=1024 so the head shapes match
    t5-large) and a word-level tokenizer; AutoTokenizer/T5EncoderModel.from_pretrained work offline."""
    import torch
    from tokenizers import Tokenizer, models, pre_tokenizers, processors
    from transformers import PreTrainedTokenizerFast, T5Config, T5EncoderModel
    from text2loc_amd.synth import COLOR_NAMES, KNOWN_CLASS

    os.makedirs(dirname, exist_ok=True)
    vocab = {"<pad>": 0, "</s>": 1, "<unk>": 2}
    words = list(TEMPLATE_WORDS)
    for c in KNOWN_CLASS + COLOR_NAMES:
        words.extend(c.replace("-", " - ").split())
    for w in words:
        if w not in vocab:
            vocab[w] = len(vocab)
    tok = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.Whitespace()
    tok.post_processor = processors.TemplateProcessing(single="$A </s>", special_tokens=[("</s>", 1)])
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, pad_token="<pad>", eos_token="</s>", unk_token="<unk>")
    fast.save_pretrained(dirname)
    torch.manual_seed(seed)
    cfg = T5Config(vocab_size=len(vocab), d_model=1024, d_kv=64, d_ff=256, num_layers=1, num_heads=4,
                   is_encoder_decoder=False, use_cache=False)
    T5EncoderModel(cfg).save_pretrained(dirname)
    return dirname


def make_args(hf_dir: str, pointnet_path: str, class_embed: bool, color_embed: bool, **over):
    a = argparse.Namespace(
        coarse_embed_dim=256, object_size=28, object_inter_module_num_heads=4,
        object_inter_module_num_layers=2, hungging_model=hf_dir, fixed_embedding=True,
        intra_module_num_layers=1, intra_module_num_heads=4, inter_module_num_layers=1,
        inter_module_num_heads=4, class_embed=class_embed, color_embed=color_embed,
        use_features=["class", "color", "position", "num"], pointnet_layers=3, pointnet_variation=0,
        pointnet_numpoints=256, pointnet_path=pointnet_path, pointnet_freeze=True, pointnet_features=2,
        ranking_loss="contrastive", top_k=[1, 3, 5], threshs=[5, 10, 15], batch_size=16,
        temperature=0.1,
    )
    for k, v in over.items():
        setattr(a, k, v)
    return a


def make_pointnet_ckpt(path: str, seed: int = 0):
    import torch
    from models.pointcloud.pointnet2 import PointNet2

    torch.manual_seed(seed)
    ns = argparse.Namespace(pointnet_layers=3, pointnet_variation=0)
    pn = PointNet2(22, 8, ns)
    torch.save(pn.state_dict(), path)
    return path


def build_objects(cells: dict, seed: int = 0):
    """Reference ``Object3d`` instances over the point sets of text2loc_amd.synth.make_object_points.
    Returns List[List[Object3d]]."""
    from datapreparation.kitti360pose.imports import Object3d
    from text2loc_amd.synth import make_object_points

    out = [[] for _ in range(len(cells["counts"]))]
    gid = 0
    for b, o, label, xyz, rgb in make_object_points(cells, seed):
        out[b].append(Object3d(o, gid, xyz, rgb, label))
        gid += 1
    return out


def write_dataset(base: str, objects, seed: int = 0, scene="2013_05_28_drive_0010_sync", n_poses=None,
                  cell_size=30.0, grid=8):
    """Synthetic cells/poses pickles built with the reference's own Cell/Pose/Description classes."""
    from datapreparation.kitti360pose.imports import Cell, DescriptionBestCell, DescriptionPoseCell, Pose

    rng = np.random.default_rng([seed, 0xDA7A])
    scene_short = scene.split("_")[-2]
    cells = []
    for i, objs in enumerate(objects):
        gx, gy = i % grid, i // grid
        lo = np.array([gx * cell_size / 2.0, gy * cell_size / 2.0, 0.0])  # stride = cell_size/2 (overlapping cells)
        bbox = np.hstack((lo, lo + cell_size))
        cells.append(Cell(i, scene_short, objs, cell_size, bbox))
    n_poses = n_poses or len(cells)
    dirs = ["north", "south", "east", "west", "on-top"]
    poses = []
    for p in range(n_poses):
        ci = int(rng.integers(0, len(cells)))
        cell = cells[ci]
        pin = rng.uniform(0.3, 0.7, size=3)
        pw = cell.bbox_w[0:3] + pin * cell_size
        descs = []
        picks = rng.choice(len(cell.objects), size=6, replace=len(cell.objects) < 6)
        for oi in picks:
            obj = cell.objects[int(oi)]
            d = DescriptionPoseCell(obj, dirs[int(rng.integers(0, 5))], np.zeros(2), np.zeros(2), obj.get_center())
            descs.append(DescriptionBestCell.from_unmatched(d))
        poses.append(Pose(pin, pw, cell.id, scene_short, descs))
    os.makedirs(osp.join(base, "cells"), exist_ok=True)
    os.makedirs(osp.join(base, "poses"), exist_ok=True)
    pickle.dump(cells, open(osp.join(base, "cells", scene + ".pkl"), "wb"))
    pickle.dump(poses, open(osp.join(base, "poses", scene + ".pkl"), "wb"))
    return cells, poses
