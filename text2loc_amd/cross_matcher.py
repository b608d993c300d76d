"""Drop-in for the reference's fine stage: ``models.cross_matcher.CrossMatch`` (models/cross_matcher.py:39-141) and
``evaluation.pipeline.run_fine`` (evaluation/pipeline.py:88-204).

    CrossMatch(known_classes, known_colors, args)
        .forward(objects, hints, object_points) -> Tensor[B,2]   offsets = pose estimate inside each cell, in [0,1]^2
        .state_dict() / .load_state_dict()    same key names as the reference's fine checkpoint
        .encode_cells(objects, object_points) -> Tensor[B,16,128]   (new: the query-independent half, cacheable per cell)
        .match(cell_desc, hint_desc, cell_index, hint_index) -> Tensor[P,2]

The text branch (``LanguageEncoder(is_fine=True)``: T5 + one Transformer layer over tokens + Linear/BN) stays on
PyTorch-ROCm; the 3D-submap branch (ObjectEncoder at fine_embed_dim incl. PointNet++ in the published mode), the cascaded
cross-attention decoder layers and the offset head run in the engine (t2l_fine_*). The nn.Modules below are PARAMETER
CONTAINERS for the engine-side tensors. Eval only (the fine model's training step is not built).
"""
from __future__ import annotations

from copy import copy
from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn

from . import packing
from .cell_retrieval import LanguageEncoder, ObjectEncoderParams
from .engine import Engine, T2LError

FINE_DIM, PAD_SIZE = 128, 16


def get_mlp_offset(dims: List[int]) -> nn.Sequential:
    """Linear/ReLU stack without trailing activation; key layout ``{0,2,...}`` (models/cross_matcher.py:17-36)."""
    mods: list = []
    for i in range(len(dims) - 1):
        mods.append(nn.Linear(dims[i], dims[i + 1]))
        if i < len(dims) - 2:
            mods.append(nn.ReLU())
    return nn.Sequential(*mods)


class PadObject:
    """``Object3d.create_padding()`` (datapreparation/kitti360pose/imports.py:75-83): label 'pad', 8 points within 1 mm of
    the origin, black. The reference draws the 8 points with the global numpy RNG on every call; here they are fixed."""

    label = "pad"
    _xyz = np.random.default_rng(0xBAD).random((8, 3)) * 0.001
    _rgb = np.zeros((8, 3), dtype=np.float32)

    def __init__(self):
        self.xyz, self.rgb = PadObject._xyz, PadObject._rgb


def pad_objects(objects: list, pad_size: int = PAD_SIZE) -> list:
    """Cut to / pad to ``pad_size`` objects (dataloading/kitti360pose/eval.py:147-156)."""
    objs = list(objects)[:pad_size]
    while len(objs) < pad_size:
        objs.append(PadObject())
    return objs


def create_hint_description(pose) -> List[str]:
    """dataloading/kitti360pose/base.py:60-68."""
    return [f"The pose is {d.direction} of a {d.object_color_text} {d.object_label}." for d in pose.descriptions]


class CrossMatch(nn.Module):
    def __init__(self, known_classes: List[str], known_colors: List[str], args, language_encoder: Optional[nn.Module] = None):
        super().__init__()
        self.args = args
        self.embed_dim = args.fine_embed_dim
        if self.embed_dim != FINE_DIM:
            raise T2LError(f"the engine's fine stage is built for fine_embed_dim={FINE_DIM}, got {self.embed_dim}")
        if getattr(args, "pad_size", PAD_SIZE) != PAD_SIZE:
            raise T2LError(f"the engine's fine stage is built for pad_size={PAD_SIZE}")
        n_layers = int(args.fine_num_decoder_layers)
        if not 0 <= n_layers <= 4:
            raise T2LError(f"the engine's fine stage holds 0..4 decoder layers, got fine_num_decoder_layers={n_layers}")
        self.object_encoder = ObjectEncoderParams(FINE_DIM, known_classes, args, known_colors)
        self.language_encoder = language_encoder if language_encoder is not None else LanguageEncoder(
            FINE_DIM, hungging_model=args.hungging_model, fixed_embedding=args.fixed_embedding,
            intra_module_num_layers=args.fine_intra_module_num_layers, intra_module_num_heads=args.fine_intra_module_num_heads,
            is_fine=True)
        self.mlp_offsets = get_mlp_offset([FINE_DIM, FINE_DIM // 2, 2])
        mk = lambda: nn.TransformerDecoderLayer(d_model=FINE_DIM, nhead=args.fine_num_decoder_heads, dim_feedforward=4 * FINE_DIM)
        if n_layers > 0:
            self.cross_hints = nn.ModuleList([mk() for _ in range(n_layers)])
            self.cross_objects = nn.ModuleList([mk() for _ in range(n_layers)])
        else:  # cross_matcher.py:75-79: ONE layer (state_dict keys "cross_hints.*", no index), the hints attend the raw objects once
            self.cross_hints = mk()
            self.cross_objects = None
        self._engine: Optional[Engine] = None
        self._weights_version = None

    @property
    def device(self):
        return next(self.mlp_offsets.parameters()).device

    def get_device(self):
        return self.device

    # ---- engine plumbing ----------------------------------------------------------------------------------
    def engine(self) -> Engine:
        dev = self.device
        if dev.type != "cuda":
            raise T2LError("the fine stage runs on the MI355X only (model.to('cuda')); there is no CPU fallback")
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        if self._engine is None or self._engine.device != idx:
            self._engine = Engine(idx)
            self._weights_version = None
        params = [p for n, p in self.state_dict(keep_vars=True).items() if not n.startswith("language_encoder.")]
        version = tuple((p.data_ptr(), p._version) for p in params)
        if version != self._weights_version:
            a = self.args
            sd = {k: v for k, v in self.state_dict().items() if not k.startswith("language_encoder.")}
            self._engine.fine_load_weights(sd, class_embed=bool(getattr(a, "class_embed", False)),
                                           color_embed=bool(getattr(a, "color_embed", False)), use_features=tuple(a.use_features),
                                           num_layers=a.fine_num_decoder_layers, num_heads=a.fine_num_decoder_heads)
            self._weights_version = version
        return self._engine

    # ---- the two halves -----------------------------------------------------------------------------------
    @torch.no_grad()
    def encode_cells(self, objects, object_points=None) -> torch.Tensor:
        """[B,16,128] unit-row object descriptors of B padded cells (cross_matcher.py:97-104)."""
        if self.training:
            raise T2LError("the fine stage is eval-only here (call model.eval()); its training step is not built")
        if any(len(o) != PAD_SIZE for o in objects):
            raise T2LError(f"every cell must hold exactly pad_size={PAD_SIZE} objects (cross_matcher.pad_objects pads / cuts)")
        eng = self.engine()
        dev = self.device
        oe = self.object_encoder
        if any(getattr(o, "_t2l_feat", None) is None for objs in objects for o in objs):
            packed = packing.pack_cells_gpu(eng, objects, oe.known_classes, oe.known_colors, dev)
        else:
            packed = packing.to_device(packing.pack_cells(objects, oe.known_classes, oe.known_colors), dev)
        a = self.args
        if "class" in a.use_features and not bool(getattr(a, "class_embed", False)):
            if object_points is None or any(p is None for p in object_points):
                raise T2LError("class_embed is off: object_points must hold, per cell, features2 [16,256] or the point batch")
            first = object_points[0]
            if isinstance(first, (torch.Tensor, np.ndarray)):
                pn = torch.cat([torch.as_tensor(np.asarray(p) if not isinstance(p, torch.Tensor) else p) for p in object_points]).to(dev, torch.float32)
            else:
                get = lambda p, n: p[n] if isinstance(p, dict) else getattr(p, n)
                pos = torch.cat([torch.as_tensor(get(p, "pos")).reshape(-1, 256, 3) for p in object_points]).to(dev, torch.float32)
                rgb = torch.cat([torch.as_tensor(get(p, "x")).reshape(-1, 256, 3) for p in object_points]).to(dev, torch.float32)
                pn = eng.pointnet_features(pos.contiguous(), rgb.contiguous(), np.arange(0, len(objects) * PAD_SIZE + 1, PAD_SIZE, dtype=np.int32))
            packed["pn_feat"] = pn.reshape(-1, 256).contiguous()
        return eng.fine_encode_objects(packed)

    @torch.no_grad()
    def match(self, cell_desc: torch.Tensor, hint_desc: torch.Tensor, cell_index=None, hint_index=None) -> torch.Tensor:
        ci = None if cell_index is None else torch.as_tensor(cell_index, dtype=torch.int32, device=cell_desc.device).contiguous()
        hi = None if hint_index is None else torch.as_tensor(hint_index, dtype=torch.int32, device=cell_desc.device).contiguous()
        return self.engine().fine_match(cell_desc.contiguous(), hint_desc.detach().float().contiguous(), ci, hi)

    @torch.no_grad()
    def forward(self, objects, hints, object_points=None) -> torch.Tensor:
        """One (pose, cell) pair per batch entry, as ``run_fine`` calls it (evaluation/pipeline.py:113-116)."""
        hint_enc = self.language_encoder(hints)  # [B, n_hints, 128]  (cross_matcher.py:95)
        return self.match(self.encode_cells(objects, object_points), hint_enc)


@torch.no_grad()
def encode_pose_hints(language_encoder, texts: List[str], max_batch: int = 256) -> torch.Tensor:
    """Hint encodings [n_poses, n_hints, 128] of one text per pose, equal to what the reference's per-pose call produces.

    The reference runs the model once per pose on ``max(top_k)`` copies of that pose's text (evaluation/pipeline.py:113-116,
    dataloading/kitti360pose/eval.py:181-186), so ``padding='longest'`` pads to THAT pose's longest hint — and the
    intra-module TransformerEncoderLayers have no padding mask and max-pool over every token position
    (language_encoder.py:127-134), i.e. the pad length is part of the result. Poses are therefore batched only with
    poses of the same (hint count, padded token length): inside such a group a joint call pads exactly like the
    per-pose call. One tokenizer pass per pose decides the groups; the model runs once per group chunk."""
    tok = getattr(language_encoder, "tokenizer", None)
    split = getattr(language_encoder, "split_sentences", None)
    if tok is None or split is None:  # injected encoders without a tokenizer (tests with a table): one call per pose
        return torch.cat([language_encoder([t]) for t in texts]) if texts else torch.zeros((0, 0, FINE_DIM))
    groups = {}
    for i, t in enumerate(texts):
        ss = split(t)
        ids = tok(ss, return_tensors="pt", padding="longest")["input_ids"]
        groups.setdefault((len(ss), int(ids.shape[1])), []).append(i)
    if len({k[0] for k in groups}) > 1:
        raise T2LError(f"poses carry different numbers of hints: {sorted({k[0] for k in groups})}")
    out = None
    for _, members in sorted(groups.items()):
        for lo in range(0, len(members), max_batch):
            sel = members[lo:lo + max_batch]
            enc = language_encoder([texts[i] for i in sel])
            if out is None:
                out = torch.empty((len(texts),) + tuple(enc.shape[1:]), dtype=enc.dtype, device=enc.device)
            out[torch.as_tensor(sel, device=enc.device)] = enc
    return out if out is not None else torch.zeros((0, 0, FINE_DIM))


@torch.no_grad()
def run_fine(model: CrossMatch, retrievals, dataloader, args, transform_fine=None, object_points_fn=None,
             return_offsets: bool = False):
    """evaluation/pipeline.py:88-204: offsets of every pose against its max(top_k) retrieved cells -> {k: {t: accuracy}}.
    The reference builds a ``Kitti360TopKDataset`` item per pose and runs one forward per pose, re-encoding a cell for
    every pose that retrieved it; here every distinct retrieved cell is padded and encoded ONCE, every pose's hints are
    encoded once, and all poses x max(top_k) pairs are matched in one launch.
    ``object_points_fn(list_of_padded_object_lists) -> object_points`` supplies the PointNet++ inputs in the published
    feature mode. Without one they are sampled here (``packing.sample_object_points``) under ``transform_fine`` — a transform
    name ("fixed" / "normalize"), or, as in the reference's script, chosen from ``args.no_pc_augment_fine``
    (evaluation/pipeline.py:220-223: FixedPoints only under the flag, which the published commands pass)."""
    model.eval()
    a = model.args
    if object_points_fn is None and "class" in a.use_features and not bool(getattr(a, "class_embed", False)):
        name = transform_fine if isinstance(transform_fine, str) else packing.point_transform_from_args(args, fine=True)
        pts_rng = np.random.default_rng(int(getattr(args, "seed", 0) or 0))

        def object_points_fn(chunk):
            return packing.sample_object_points(chunk, 256, pts_rng, transform=name)
    ds = dataloader.dataset
    poses, cells = ds.all_poses, ds.all_cells
    K = max(args.top_k)
    assert len(poses) == len(retrievals) and all(len(r) == K for r in retrievals), "retrievals must be trimmed to max(top_k)"
    cells_dict = {c.id: c for c in cells}
    uniq = sorted({str(cid) for r in retrievals for cid in r})
    row = {cid: i for i, cid in enumerate(uniq)}
    # OPT-IN, like coarse.eval_epoch (args.shard_layout set to anything but None / "none"): with more than one process (one per GPU)
    # the distinct cells and the (pose, cell) pairs are split across the ranks in contiguous blocks — no exchange inside either
    # stage — and all-gathered once each. Every rank must then call run_fine with the same retrievals (proven by one small
    # all_reduce); without the option an initialised process group changes nothing (rank-0-only validation stays local).
    from .sharded import assert_replicated, gather_rows, shard_bounds, world_rank

    layout = getattr(args, "shard_layout", None) or getattr(a, "shard_layout", None)
    world, rank = world_rank() if layout not in (None, "", "none", "off") else (1, 0)
    if world > 1:
        assert_replicated([torch.tensor([row[str(cid)] for r in retrievals for cid in r], dtype=torch.float64)], None, "retrievals")
    c_lo, c_hi = shard_bounds(len(uniq), world, rank)
    padded = [pad_objects(cells_dict[cid].objects) for cid in uniq[c_lo:c_hi]]
    descs = []
    for lo in range(0, len(padded), 2048):
        chunk = padded[lo:lo + 2048]
        descs.append(model.encode_cells(chunk, object_points_fn(chunk) if object_points_fn is not None else None))
    cell_desc = torch.cat(descs) if descs else torch.zeros((0, PAD_SIZE, FINE_DIM), device=model.device)
    cell_desc = gather_rows(cell_desc, len(uniq)) if world > 1 else cell_desc
    hint_desc = encode_pose_hints(model.language_encoder, [" ".join(create_hint_description(p)) for p in poses])
    ci = np.array([row[str(cid)] for r in retrievals for cid in r], dtype=np.int32)
    hi = np.repeat(np.arange(len(poses), dtype=np.int32), K)
    p_lo, p_hi = shard_bounds(len(ci), world, rank)
    local = model.match(cell_desc, hint_desc, ci[p_lo:p_hi], hi[p_lo:p_hi]) if p_hi > p_lo else \
        torch.zeros((0, 2), device=model.device)
    offsets = (gather_rows(local, len(ci)) if world > 1 else local).cpu().numpy().reshape(len(poses), K, 2)
    from .coarse import _pose_cell_tables, sample_accuracies_batch

    pose_xy, pose_scene, bbox_xy, size, scene = _pose_cell_tables(poses, cells, retrievals)
    ok = sample_accuracies_batch(pose_xy, pose_scene, bbox_xy, size, scene, offsets.astype(np.float64), args.top_k, args.threshs)
    acc = {k: {t: float(np.mean(ok[k][t])) for t in args.threshs} for k in args.top_k}
    return (acc, offsets) if return_offsets else acc  # offsets f32[n_poses, max(top_k), 2]: the per-pair estimates
