"""Drop-in for the reference's ``models.cell_retrieval.CellRetrievalNetwork`` (models/cell_retrieval.py:13-120)
on the boundary that ``training.coarse.eval_epoch`` / ``evaluation.pipeline.run_coarse`` consume:

    CellRetrievalNetwork(known_classes, known_colors, args)
        .embed_dim  .eval()  .train()  .to(device)  .device / .get_device()
        .state_dict() / .load_state_dict(strict=False)     (same parameter names as the reference checkpoint)
        .encode_objects(objects, object_points) -> Tensor[B,256]   # fused HIP kernel (libt2l.so)
        .encode_text(descriptions)              -> Tensor[B,256]   # frozen T5 + head, PyTorch-ROCm (unchanged path)
        .forward() raises, as in the reference (cell_retrieval.py:112-113)

The object branch's modules below are PARAMETER CONTAINERS only (so that checkpoints load and save with the
reference's key names); their arithmetic runs in the HIP engine. The text branch stays on PyTorch, as the
north star prescribes. Not built yet (DESIGN.md "out of scope / next"): training-mode forward+backward of the
object branch (BatchNorm batch statistics) and PointNet++ itself — in the published feature mode
(class_embed off) ``object_points`` must carry precomputed ``features2`` per cell.
"""
from __future__ import annotations

import re
from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import packing
from .engine import EMBED_DIM, OBJECT_SIZE, Engine, T2LError


def get_mlp(channels: Sequence[int], add_batchnorm: bool = True, last_relu: bool = True) -> nn.Sequential:
    """[Linear, BatchNorm1d, ReLU] blocks; key layout ``{i}.0`` Linear, ``{i}.1`` BatchNorm (the reference's
    get_mlp has a trailing ReLU, get_mlp2 = last_relu False; models/language_encoder.py:16-74)."""
    blocks = []
    for i in range(1, len(channels)):
        mods: list = [nn.Linear(channels[i - 1], channels[i])]
        if add_batchnorm:
            mods.append(nn.BatchNorm1d(channels[i]))
        if last_relu or i < len(channels) - 1:
            mods.append(nn.ReLU())
        blocks.append(nn.Sequential(*mods))
    return nn.Sequential(*blocks)


class ObjectEncoderParams(nn.Module):
    """Parameters of models/object_encoder.py:28-64 under the same names (PointNet++ sub-module excluded)."""

    def __init__(self, embed_dim: int, known_classes: List[str], args):
        super().__init__()
        self.known_classes = packing.class_table(known_classes)
        self.known_colors = packing.color_table()
        self.class_embedding = nn.Embedding(len(self.known_classes), embed_dim, padding_idx=0)
        self.color_embedding = nn.Embedding(len(self.known_colors), embed_dim, padding_idx=0)
        self.pos_encoder = get_mlp([3, 64, embed_dim])
        self.color_encoder = get_mlp([3, 64, embed_dim])
        self.num_encoder = get_mlp([1, 64, embed_dim])
        self.mlp_pointnet = get_mlp([256, embed_dim])  # pointnet_features == 2 -> features2 (object_encoder.py:60-61)
        self.mlp_merge = get_mlp([len(args.use_features) * embed_dim, embed_dim])


class LanguageEncoder(nn.Module):
    """Text branch (models/language_encoder.py:76-152): frozen T5 encoder -> 1 Transformer layer over tokens (no
    padding mask) -> max over tokens -> Linear+BN -> residual Transformer layer over the hint sentences -> max.
    Stays on PyTorch-ROCm. ``llm_model``/``tokenizer`` may be injected (tests, precomputed-embedding runs)."""

    def __init__(self, embedding_dim: int, hungging_model: Optional[str] = None, fixed_embedding: bool = False,
                 intra_module_num_layers: int = 2, intra_module_num_heads: int = 4, is_fine: bool = False,
                 inter_module_num_layers: int = 2, inter_module_num_heads: int = 4, llm_model=None, tokenizer=None,
                 input_dim: Optional[int] = None):
        super().__init__()
        self.is_fine = is_fine
        self.fixed_embedding = bool(fixed_embedding)
        if llm_model is None and hungging_model is not None:
            from transformers import AutoTokenizer, T5EncoderModel

            tokenizer = AutoTokenizer.from_pretrained(hungging_model)
            llm_model = T5EncoderModel.from_pretrained(hungging_model)
        self.tokenizer = tokenizer
        self.llm_model = llm_model
        if self.fixed_embedding and isinstance(llm_model, nn.Module):
            for p in llm_model.parameters():
                p.requires_grad = False
        if input_dim is None:
            input_dim = llm_model.encoder.embed_tokens.weight.shape[-1]
        self.intra_module = nn.ModuleList([
            nn.TransformerEncoderLayer(input_dim, intra_module_num_heads, dim_feedforward=input_dim * 4)
            for _ in range(intra_module_num_layers)])
        self.inter_mlp = get_mlp([input_dim, embedding_dim], last_relu=False)
        if not is_fine:
            self.inter_module = nn.ModuleList([
                nn.TransformerEncoderLayer(embedding_dim, inter_module_num_heads, dim_feedforward=embedding_dim * 4)
                for _ in range(inter_module_num_layers)])

    @staticmethod
    def split_sentences(text: str) -> List[str]:
        return [s for s in re.split(r"(?<=[.!?])\s+", text.strip()) if s]

    def head(self, hidden: torch.Tensor, batch_size: int) -> torch.Tensor:
        """hidden: last_hidden_state [n_sentences_total, L, C] -> [B, D] (language_encoder.py:127-148)."""
        x = hidden.permute(1, 0, 2)
        for layer in self.intra_module:
            x = layer(x)
        x = x.permute(1, 0, 2).contiguous().max(dim=1)[0]
        x = self.inter_mlp(x)
        x = x.view(batch_size, x.shape[0] // batch_size, -1)
        if self.is_fine:
            return x
        x = x.permute(1, 0, 2)
        for layer in self.inter_module:
            x = x + layer(x)
        return x.max(dim=0)[0]

    def forward(self, descriptions: List[str]) -> torch.Tensor:
        sentences: List[str] = []
        for d in descriptions:
            sentences.extend(self.split_sentences(d))
        inputs = self.tokenizer(sentences, return_tensors="pt", padding="longest")
        dev = self.device
        out = self.llm_model(input_ids=inputs["input_ids"].to(dev), attention_mask=inputs["attention_mask"].to(dev),
                             output_attentions=False)
        hidden = out.last_hidden_state
        if self.fixed_embedding:
            hidden = hidden.detach()
        return self.head(hidden, len(descriptions))

    @property
    def device(self):
        return next(self.inter_mlp.parameters()).device


class CellRetrievalNetwork(nn.Module):
    def __init__(self, known_classes: List[str], known_colors: List[str], args, language_encoder: Optional[nn.Module] = None):
        super().__init__()
        self.args = args
        self.embed_dim = args.coarse_embed_dim
        if self.embed_dim != EMBED_DIM:
            raise T2LError(f"the engine is built for coarse_embed_dim={EMBED_DIM}, got {self.embed_dim}")
        if args.object_size != OBJECT_SIZE:
            raise T2LError(f"the engine is built for object_size={OBJECT_SIZE}, got {args.object_size}")
        self.object_size = args.object_size
        self.object_encoder = ObjectEncoderParams(self.embed_dim, known_classes, args)
        self.obj_inter_module = nn.ModuleList([
            nn.TransformerEncoderLayer(self.embed_dim, args.object_inter_module_num_heads,
                                       dim_feedforward=2 * self.embed_dim)
            for _ in range(args.object_inter_module_num_layers)])
        self.language_encoder = language_encoder if language_encoder is not None else LanguageEncoder(
            self.embed_dim, hungging_model=args.hungging_model, fixed_embedding=args.fixed_embedding,
            intra_module_num_layers=args.intra_module_num_layers, intra_module_num_heads=args.intra_module_num_heads,
            is_fine=False, inter_module_num_layers=args.inter_module_num_layers,
            inter_module_num_heads=args.inter_module_num_heads)
        self._engine: Optional[Engine] = None
        self._weights_version = None

    # ---- reference surface ------------------------------------------------------------------------------
    def forward(self):
        raise Exception("Not implemented.")

    @property
    def device(self):
        return self.language_encoder.device

    def get_device(self):
        return self.language_encoder.device

    def encode_text(self, descriptions):
        return F.normalize(self.language_encoder(descriptions))

    @torch.no_grad()
    def encode_objects(self, objects, object_points=None):
        if self.training:
            raise T2LError("training-mode encode_objects (BatchNorm batch statistics + backward) is not built yet; "
                           "call model.eval() — the engine implements the reference's eval path")
        dev = self.device
        if dev.type != "cuda":
            raise T2LError("encode_objects runs on the MI355X only (model.to('cuda')); there is no CPU fallback")
        a = self.args
        class_embed = bool(getattr(a, "class_embed", False))
        pn = None
        if "class" in a.use_features and not class_embed:
            if object_points is None or any(p is None for p in object_points):
                raise T2LError("class_embed is off: object_points must hold precomputed PointNet++ features2 "
                               "[n_i,256] per cell (PointNet++ kernels are not built yet)")
            pn = [p.detach().cpu().numpy() if isinstance(p, torch.Tensor) else np.asarray(p) for p in object_points]
        eng = self.engine()
        if any(getattr(o, "_t2l_feat", None) is None for objs in objects for o in objs):
            # raw points not reduced yet: one HBM pass on the GPU (t2l_reduce_objects) instead of three numpy
            # reductions per object on the host (what the reference redoes on every call)
            packed = packing.pack_cells_gpu(eng, objects, self.object_encoder.known_classes,
                                            self.object_encoder.known_colors, dev)
            if pn is not None:
                packed["pn_feat"] = torch.from_numpy(np.concatenate(
                    [np.asarray(f, dtype=np.float32).reshape(-1, 256) for f in pn], axis=0)).to(dev)
            return eng.encode_cells(packed)
        packed = packing.pack_cells(objects, self.object_encoder.known_classes, self.object_encoder.known_colors, pn)
        return eng.encode_cells(packing.to_device(packed, dev))

    # ---- engine plumbing --------------------------------------------------------------------------------
    def engine(self) -> Engine:
        dev = self.device
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        if self._engine is None or self._engine.device != idx:
            self._engine = Engine(idx)
            self._weights_version = None
        version = tuple((p.data_ptr(), p._version) for p in self._object_params())
        if version != self._weights_version:
            self.sync_weights()
            self._weights_version = version
        return self._engine

    def _object_params(self):
        for n, p in self.state_dict(keep_vars=True).items():
            if n.startswith(("object_encoder.", "obj_inter_module.")):
                yield p

    def sync_weights(self):
        """Fold BatchNorm, re-lay out and upload the object-branch weights (t2l_load_weights)."""
        a = self.args
        sd = {k: v for k, v in self.state_dict().items() if k.startswith(("object_encoder.", "obj_inter_module."))}
        self._engine.load_weights(sd, class_embed=bool(getattr(a, "class_embed", False)),
                                  color_embed=bool(getattr(a, "color_embed", False)),
                                  use_features=tuple(a.use_features), num_layers=a.object_inter_module_num_layers,
                                  num_heads=a.object_inter_module_num_heads)
