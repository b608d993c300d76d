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
north star prescribes. PointNet++ (published feature mode) runs in the engine too when ``object_points`` carries the
cells' point batches (parity unpinned, see oracle/t2l_oracle_pointnet.py). Under ``model.train()`` ``encode_objects`` runs the engine's training-mode forward
(batch-statistics BatchNorm, the TransformerEncoderLayers' dropout) and returns a tensor whose ``backward``
runs the engine's backward kernels, which ADD into the ``.grad`` of these same nn.Parameters; step them with
``text2loc_amd.optim.Adam`` (or any torch optimizer). PointNet++ trains too (the published configuration: no
``--pointnet_freeze``): with point batches in ``object_points`` the backbone's training-mode forward
(t2l_pointnet_features_train: per-cell batch statistics, as the reference's one-call-per-cell loop) and backward
(t2l_pointnet_backward) run in the engine; with ``--pointnet_freeze`` only the forward does (the reference freezes the
weights, not the BatchNorm mode). Precomputed ``features2`` passed as tensors that require grad receive their gradient.
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


class _PointConvParams(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.local_nn = get_mlp(channels)


class _SetAbstractionParams(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.point_conv = _PointConvParams(channels)


class _GlobalAbstractionParams(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.mlp = get_mlp(channels)


class PointNet2Params(nn.Module):
    """Parameters of models/pointcloud/pointnet2.py:52-64 under the same names (so that the published
    ``pointnet_acc0.86_lr1_p256.pth`` / ``coarse.pth`` key layout loads); the arithmetic is t2l_pointnet_features."""

    def __init__(self, num_classes: int, num_colors: int):
        super().__init__()
        self.sa1 = _SetAbstractionParams([3 + 3, 32, 64])
        self.sa2 = _SetAbstractionParams([64 + 3, 128, 128])
        self.sa3 = _SetAbstractionParams([128 + 3, 256, 256])
        self.ga = _GlobalAbstractionParams([256 + 3, 512, 1024])
        self.lin1 = nn.Linear(1024, 512)
        self.lin2 = nn.Linear(512, 256)
        self.class_classifier = nn.Linear(256, num_classes)
        self.color_classifier = nn.Linear(256, num_colors)


class ObjectEncoderParams(nn.Module):
    """Parameters of models/object_encoder.py:28-64 under the same names (PointNet++ sub-module excluded)."""

    def __init__(self, embed_dim: int, known_classes: List[str], args, known_colors: Optional[List[str]] = None):
        super().__init__()
        self.pointnet = PointNet2Params(len(known_classes), len(known_colors) if known_colors is not None else 8)
        if bool(getattr(args, "pointnet_freeze", False)):  # store_true flag, default off (training/args.py:54)
            self.pointnet.requires_grad_(False)
        self.known_classes = packing.class_table(known_classes)
        self.known_colors = packing.color_table()
        self.class_embedding = nn.Embedding(len(self.known_classes), embed_dim, padding_idx=0)
        self.color_embedding = nn.Embedding(len(self.known_colors), embed_dim, padding_idx=0)
        self.pos_encoder = get_mlp([3, 64, embed_dim])
        self.color_encoder = get_mlp([3, 64, embed_dim])
        self.num_encoder = get_mlp([1, 64, embed_dim])
        self.mlp_pointnet = get_mlp([256, embed_dim])  # pointnet_features == 2 -> features2 (object_encoder.py:60-61)
        self.mlp_merge = get_mlp([len(args.use_features) * embed_dim, embed_dim])


class _TextHeadTrainFn(torch.autograd.Function):
    """The head after T5 under model.train(): forward and backward are HIP (t2l_text_head_train / t2l_text_head_backward). Parameter
    gradients do not flow through autograd: the engine accumulates them straight into the parameters' ``.grad`` buffers (bound by
    pointer), which torch's Adam steps; the hidden states are constants (T5 is frozen). ``hook`` is a dummy leaf that makes
    autograd call ``backward``."""

    @staticmethod
    def forward(ctx, hook, hidden, enc, batch_size, p_drop, seed):
        out = enc._th_train_engine.text_head_train(hidden, batch_size, dropout_p=p_drop, seed=seed)
        ctx.enc = enc
        ctx.token = enc._th_train_token = object()
        return out

    @staticmethod
    def backward(ctx, grad_out):
        enc = ctx.enc
        if enc._th_train_token is not ctx.token:
            raise T2LError("backward of a stale text-head call: the engine keeps the activations of the LAST training-mode forward only")
        enc._th_train_engine.text_head_backward(grad_out.contiguous().float())
        return None, None, None, None, None, None


def _apply_sync_bn(eng: Engine, cfg):
    """cfg: None (per-rank BatchNorm statistics) or (group,) — see CellRetrievalNetwork.sync_batchnorm."""
    if getattr(eng, "_sync_bn_cfg", None) != cfg:
        if cfg is None:
            eng.train_sync_bn(enable=False)
        else:
            eng.train_sync_bn(group=cfg[0])
        eng._sync_bn_cfg = cfg


class LanguageEncoder(nn.Module):
    """Text branch (models/language_encoder.py:76-152): frozen T5 encoder -> 1 Transformer layer over tokens (no
    padding mask) -> max over tokens -> Linear+BN -> residual Transformer layer over the hint sentences -> max.
    T5 stays on PyTorch-ROCm; the d=1024 layer + max + inter_mlp behind it run in the HIP engine in eval mode on the GPU
    (``_head_first_half``). ``llm_model``/``tokenizer`` may be injected (tests, precomputed-embedding runs)."""

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

    # ---- the head after T5 --------------------------------------------------------------------------------------------
    use_engine_head = True   # eval mode on the GPU: the whole head after T5 runs in the HIP engine (t2l_text_head + t2l_text_inter)
    head_engine_calls = 0    # calls served by the engine / by the PyTorch path (f16-range overflow, training mode, CPU, ...)
    head_torch_calls = 0
    inter_engine_calls = 0   # ... and for the inter-sentence half (t2l_text_inter)
    inter_torch_calls = 0

    @staticmethod
    def _layer_is_stock(layer, d, ff, heads) -> bool:
        """What text_head.hip hard-codes about a TransformerEncoderLayer: post-norm, ReLU, LayerNorm eps 1e-5, (d, ff, heads)."""
        act = getattr(layer, "activation", None)
        relu = act is F.relu or isinstance(act, nn.ReLU) or getattr(act, "__name__", "") == "relu"
        return (not getattr(layer, "norm_first", False) and relu and layer.linear1.in_features == d and layer.linear1.out_features == ff
                and layer.self_attn.num_heads == heads and abs(layer.norm1.eps - 1e-5) < 1e-12 and abs(layer.norm2.eps - 1e-5) < 1e-12)

    def _engine_gate(self, hidden: torch.Tensor) -> bool:
        """Everything t2l_text_head_load_weights / t2l_text_head would reject is refused HERE, so eval mode never raises where the
        PyTorch path works (shapes, D % 4, layer flavour, BatchNorm eps)."""
        if not (self.use_engine_head and hidden.is_cuda and not self.training and not torch.is_grad_enabled()):
            return False
        D = self.inter_mlp[0][0].out_features
        bn = self.inter_mlp[0][1]
        return (len(self.intra_module) == 1 and hidden.shape[-1] == 1024 and 1 <= hidden.shape[1] <= 32 and D <= 256 and D % 4 == 0
                and self._layer_is_stock(self.intra_module[0], 1024, 4096, 4) and isinstance(bn, nn.BatchNorm1d) and abs(bn.eps - 1e-5) < 1e-12)

    def _inter_gate(self, n_sent_per: int) -> bool:
        return (not self.is_fine and len(self.inter_module) == 1 and self.inter_mlp[0][0].out_features == 256 and 1 <= n_sent_per <= 32
                and self._layer_is_stock(self.inter_module[0], 256, 1024, 4))

    def _head_params(self):
        ps = getattr(self, "_th_params", None)
        if ps is None:  # (cached: walking state_dict() on every forward showed up in the profile of small batches)
            ps = self._th_params = [t for n, t in self.state_dict(keep_vars=True).items()
                                    if n.startswith(("intra_module.", "inter_mlp.", "inter_module."))]
        return ps

    def _head_engine(self, device) -> Optional[Engine]:
        """The engine context holding this head's packed weights on ``device`` (re-packed when a tensor changes)."""
        idx = device.index if device.index is not None else torch.cuda.current_device()
        version = (idx, self._head_generation) + tuple((t.data_ptr(), t._version) for t in self._head_params())
        if getattr(self, "_th_version", None) != version:
            if getattr(self, "_th_engine", None) is None or self._th_engine.device != idx:
                self._th_engine = Engine(idx)
            sd = {"language_encoder." + n: t for n, t in self.state_dict().items()
                  if n.startswith(("intra_module.", "inter_mlp.", "inter_module."))}
            self._th_engine.text_head_load_weights(sd)
            self._th_params = None  # (load_state_dict may have swapped tensors: re-collect)
            version = (idx, self._head_generation) + tuple((t.data_ptr(), t._version) for t in self._head_params())
            self._th_version = version
        return self._th_engine

    # Overflow handling. An engine call raises a device flag when a value entering an f16 product left the f16 range; the batch
    # must then be re-run on the PyTorch path. Default: read the flag after every call (one 4-byte copy + stream sync, as in
    # round 3). Between begin_deferred() and end_deferred() no call synchronises: the flags are collected on the device, an
    # overflowed batch is poisoned with NaN on the device (never silently wrong), and end_deferred() returns — after ONE sync —
    # the ordinals of the calls to re-run (coarse.eval_epoch does exactly that around its text loop).
    def begin_deferred(self):
        self._deferred = []

    def end_deferred(self) -> List[int]:
        flags, self._deferred = getattr(self, "_deferred", None), None
        if not flags:
            return []
        host = torch.stack([f.reshape(()) for f in flags]).cpu().numpy()
        return [i for i, v in enumerate(host) if v != 0]

    def _settle(self, out: torch.Tensor, flag: torch.Tensor):
        """-> (out, overflowed): synchronous check, or (deferred mode) NaN-poison on the device and report False."""
        if getattr(self, "_deferred", None) is None:
            return out, bool(flag.item())
        if not self._deferred:  # (a stage called outside head(): its own entry)
            self._deferred.append(torch.zeros((1,), dtype=torch.int32, device=out.device))
        self._deferred[-1] = torch.maximum(self._deferred[-1], flag)
        poison = torch.where(flag == 0, torch.ones((), device=out.device), torch.full((), float("nan"), device=out.device))
        return out * poison, False

    def _head_first_half(self, hidden: torch.Tensor) -> torch.Tensor:
        """[n_sentences, L, C] -> [n_sentences, D]: intra_module over the tokens, max over the tokens, inter_mlp
        (language_encoder.py:127-135). On the GPU in eval mode this is t2l_text_head (split-f16 MFMA GEMMs); the PyTorch
        modules below it are the training path and the path of a batch whose activations leave the f16 range."""
        if self._engine_gate(hidden):
            out, flag = self._head_engine(hidden.device).text_head(hidden.contiguous().float(), check=False)
            out, overflowed = self._settle(out, flag)
            if not overflowed:
                LanguageEncoder.head_engine_calls += 1
                return out
        LanguageEncoder.head_torch_calls += 1
        x = hidden.permute(1, 0, 2)
        for layer in self.intra_module:
            x = layer(x)
        x = x.permute(1, 0, 2).contiguous().max(dim=1)[0]
        return self.inter_mlp(x)

    def _open_call(self, device):
        if getattr(self, "_deferred", None) is not None:
            self._deferred.append(torch.zeros((1,), dtype=torch.int32, device=device))  # one entry per head() / forward() call

    # ---- training mode: the whole head after T5 in the engine (t2l_text_head_train / _backward) ------------------------------------
    use_engine_train_head = True
    _head_generation = 0
    train_engine_calls = 0
    _th_train_engine = None
    _th_train_token = None
    _th_train_key = None
    _th_drop_calls = 0

    def _train_gate(self, hidden: torch.Tensor, batch_size: int):
        """-> dropout p when the engine's training head can serve this call, else None (the PyTorch modules do)."""
        if not (self.use_engine_head and self.use_engine_train_head and self.training and torch.is_grad_enabled() and hidden.is_cuda
                and not hidden.requires_grad and not self.is_fine and hidden.dim() == 3 and hidden.shape[-1] == 1024
                and 1 <= hidden.shape[1] <= 32 and hidden.shape[0] % batch_size == 0 and hidden.shape[0] // batch_size <= 32):
            return None
        return self._train_structure_gate()

    def _train_structure_gate(self):
        """The call-independent half of _train_gate: is this the published head (one stock layer each side of inter_mlp, one dropout
        probability, nothing frozen)? -> that probability, else None."""
        if not (self.use_engine_head and self.use_engine_train_head and not self.is_fine):
            return None
        if not (len(self.intra_module) == 1 and len(self.inter_module) == 1 and self.inter_mlp[0][0].out_features == 256
                and self._layer_is_stock(self.intra_module[0], 1024, 4096, 4) and self._layer_is_stock(self.inter_module[0], 256, 1024, 4)
                and isinstance(self.inter_mlp[0][1], nn.BatchNorm1d) and abs(self.inter_mlp[0][1].eps - 1e-5) < 1e-12
                and self.inter_mlp[0][1].momentum == 0.1 and self.inter_mlp[0][1].track_running_stats):
            return None
        ps = set()
        for layer in (self.intra_module[0], self.inter_module[0]):
            ps.update((float(layer.dropout.p), float(layer.dropout1.p), float(layer.dropout2.p), float(layer.self_attn.dropout)))
        if len(ps) != 1 or not all(p.requires_grad for p in self._head_params() if p.dtype == torch.float32 and p.dim() > 0 and p.is_leaf
                                   and isinstance(p, nn.Parameter)):
            return None  # (site-specific probabilities / partly frozen heads: the PyTorch path)
        return ps.pop()

    # ---- the head's optimizer on the engine (t2l_text_adam_step; text2loc_amd.optim.Adam routes the head's parameters here)
    def engine_optimizer_params(self):
        """[(name, parameter)] of the head parameters the engine's training path binds — what ``optim.Adam`` hands to
        ``engine_adam_step`` instead of a torch optimizer — or [] when this head stays on the PyTorch modules in training."""
        if self._train_structure_gate() is None:
            return []
        return [(n, p) for n, p in self.named_parameters()
                if n.startswith(("intra_module.0.", "inter_mlp.0.", "inter_module.0.")) and p.requires_grad]

    def engine_adam_step(self, lr, betas=(0.9, 0.999), eps=1e-8):
        dev = next(self.intra_module.parameters()).device
        self._bind_text_train(dev)  # (a head that only ran on its PyTorch modules so far: the same .grad tensors get bound now)
        self._th_train_engine.text_adam_step(lr, betas[0], betas[1], eps)
        self._head_generation += 1  # parameters changed behind torch's back (no ._version bump): eval-mode weights / memos are stale

    def engine_zero_grad(self):
        dev = next(self.intra_module.parameters()).device
        self._bind_text_train(dev)
        self._th_train_engine.text_zero_grad()

    def _bind_text_train(self, device):
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if self._th_train_engine is None or self._th_train_engine.device != idx:
            self._th_train_engine, self._th_train_key = Engine(idx), None
        tensors = {}
        for n, t in self.state_dict(keep_vars=True).items():
            if not n.startswith(("intra_module.0.", "inter_mlp.0.", "inter_module.0.")) or n.endswith("num_batches_tracked"):
                continue
            if isinstance(t, nn.Parameter):
                if t.grad is None:
                    t.grad = torch.zeros_like(t)  # the engine accumulates into it (optimizer.zero_grad(set_to_none=True) drops it again)
                tensors["language_encoder." + n] = (t.data, t.grad)
            else:
                tensors["language_encoder." + n] = (t, None)
        key = tuple((d.data_ptr(), 0 if g is None else g.data_ptr()) for d, g in tensors.values())
        if key != self._th_train_key:
            # same head, moved storage (model.to(), a re-assigned .grad): the engine-side Adam moments must survive the re-bind
            self._th_train_engine.set_option("train_keep_adam_state", 1 if self._th_train_key is not None else 0)
            self._th_train_engine.text_train_bind(tensors)
            self._th_train_key = key
        _apply_sync_bn(self._th_train_engine, getattr(self, "_sync_bn_cfg", None))

    def head(self, hidden: torch.Tensor, batch_size: int) -> torch.Tensor:
        """hidden: last_hidden_state [n_sentences_total, L, C] -> [B, D] (language_encoder.py:127-148)."""
        p_drop = self._train_gate(hidden, batch_size)
        if p_drop is not None:
            self._bind_text_train(hidden.device)
            LanguageEncoder.train_engine_calls += 1
            self._th_drop_calls += 1
            seed = (torch.initial_seed() * 0x9E3779B1 + self._th_drop_calls * 0x85EBCA6B) & 0xFFFFFFFF
            hook = torch.zeros((), device=hidden.device, requires_grad=True)
            bn = self.inter_mlp[0][1]
            if bn.num_batches_tracked is not None:
                bn.num_batches_tracked += 1
            return _TextHeadTrainFn.apply(hook, hidden.contiguous().float(), self, batch_size, p_drop, seed)
        self._open_call(hidden.device)
        n_eng = LanguageEncoder.head_engine_calls
        x = self._head_first_half(hidden)
        gate = LanguageEncoder.head_engine_calls != n_eng  # (a batch that fell back to the PyTorch modules stays on them to the end)
        return self._head_second_half(x, batch_size, gate)

    def _head_second_half(self, x: torch.Tensor, batch_size: int, engine_ok: bool) -> torch.Tensor:
        """[n_sentences_total, D] -> [B, D]: view per description, x += inter_module(x), max over the sentences
        (language_encoder.py:137-147); the fine model returns the view (:139-140)."""
        if x.shape[0] % batch_size:
            raise T2LError(f"{x.shape[0]} sentences do not split evenly over {batch_size} descriptions")
        n_per = x.shape[0] // batch_size
        if self.is_fine:
            return x.view(batch_size, n_per, -1)
        if engine_ok and self._inter_gate(n_per):  # the 256-wide half in the engine too (t2l_text_inter)
            out, flag = self._head_engine(x.device).text_inter(x.contiguous(), batch_size, check=False)
            out, overflowed = self._settle(out, flag)
            if not overflowed:
                LanguageEncoder.inter_engine_calls += 1
                return out
        LanguageEncoder.inter_torch_calls += 1
        x = x.view(batch_size, n_per, -1).permute(1, 0, 2)
        for layer in self.inter_module:
            x = x + layer(x)
        return x.max(dim=0)[0]

    # ---- T5 behind a per-sentence cache (text2loc_amd.text_cache.TextCache) ----------------------------------------------------
    text_cache = None            # set to a TextCache: sentences it holds skip the tokenizer + T5 (valid while T5 is frozen)
    memoise_sentence_vectors = True  # eval mode: the per-sentence half of the head is memoised too (a function of sentence and L)
    cache_in_training = False    # opt-in: under model.train() the reference leaves the frozen T5 in train mode too, i.e. its dropout
                                 # (p = 0.1) stays ACTIVE (language_encoder.py:118-126 never calls llm_model.eval()); the cache
                                 # holds eval-mode states, so serving a training step from it removes that noise
    cache_calls = 0              # forward() calls served from the cache / through T5
    t5_calls = 0

    def _cache_usable(self) -> bool:
        return self.text_cache is not None and self.fixed_embedding and (not self.training or self.cache_in_training)

    def _from_cache(self, sentences: List[str], batch_size: int, hit=None) -> Optional[torch.Tensor]:
        cache = self.text_cache
        if hit is None:
            if not self._cache_usable():
                return None
            hit = cache.lookup(sentences)
            if hit is None:
                return None
        rows, L = hit
        LanguageEncoder.cache_calls += 1
        inference = not self.training and not torch.is_grad_enabled()
        if inference and self.memoise_sentence_vectors:
            version = (bool(self.use_engine_head), self._head_generation) + tuple((t.data_ptr(), t._version) for t in self._head_params())
            deferred, self._deferred = getattr(self, "_deferred", None), None  # (the memo is checked synchronously: never cache a poisoned batch)
            try:
                vec = cache.sentence_vectors(self, L, version)
            finally:
                self._deferred = deferred
            self._open_call(vec.device)
            engine_ok = self.use_engine_head and vec.is_cuda
            return self._head_second_half(vec.index_select(0, rows), batch_size, engine_ok)
        return self.head(cache.hidden_states(rows, L), batch_size)

    def forward(self, descriptions: List[str]) -> torch.Tensor:
        if self._cache_usable() and len(descriptions) and isinstance(descriptions[0], str):
            hit = self.text_cache.lookup_descriptions(descriptions)  # descriptions seen before: no regex, one dict probe each
            if hit is not None:
                return self._from_cache([], len(descriptions), hit[:2])
        sentences: List[str] = []
        per_desc = []
        for d in descriptions:
            ss = self.split_sentences(d)
            per_desc.append(len(ss))
            sentences.extend(ss)
        if len(set(per_desc)) > 1:
            # the reference reshapes [n_sentences_total] -> [batch, n // batch] (language_encoder.py:113,138): ragged hint
            # counts either crash there or silently hand sentences to the wrong description. Refuse them.
            raise T2LError(f"every description of a batch must hold the same number of sentences, got {sorted(set(per_desc))}")
        cached = self._from_cache(sentences, len(descriptions))
        if cached is not None:
            self.text_cache.remember(descriptions, sentences)
            return cached
        LanguageEncoder.t5_calls += 1
        inputs = self.tokenizer(sentences, return_tensors="pt", padding="longest")
        dev = self.device
        out = self.llm_model(input_ids=inputs["input_ids"].to(dev), attention_mask=inputs["attention_mask"].to(dev),
                             output_attentions=False)
        hidden = out.last_hidden_state
        if self.fixed_embedding:
            hidden = hidden.detach()
        return self.head(hidden, len(descriptions))

    def forward_batches(self, descriptions: List[str], batch_size: int) -> torch.Tensor:
        """``torch.cat([self(descriptions[i:i + batch_size]) for i in range(0, n, batch_size)])`` — the text loop of ``eval_epoch``
        (training/coarse.py:88-97) — without one Python round per batch. The batching is part of the reference's RESULT: the
        tokenizer pads to the longest sentence of the BATCH and the intra layer has no padding mask (language_encoder.py:113-131),
        so a description's vector depends on L = the token count of its batch's longest sentence. Behind the sentence cache with
        the eval-mode memo a per-sentence vector is a function of (sentence, L): the batches' L values come from the cached token
        counts (numpy), every distinct L is one gather from its memo table, and the inter-sentence layer — independent per
        description — runs ONCE over all descriptions. Anything else (no cache, training mode, unknown or over-long sentences)
        takes the loop."""
        n, batch_size = len(descriptions), max(1, int(batch_size))
        fast = (n > 0 and self._cache_usable() and isinstance(descriptions[0], str) and not self.training and not torch.is_grad_enabled()
                and self.memoise_sentence_vectors and not self.is_fine)
        hit = None
        if fast:
            cache = self.text_cache
            hit = cache.description_rows(descriptions)
            if hit is None:  # first sight of (some of) these descriptions: split all of them once, T5 over the unseen sentences
                sentences, per = [], set()
                for d in descriptions:
                    ss = self.split_sentences(d)
                    per.add(len(ss))
                    sentences.extend(ss)
                if len(per) == 1 and cache.lookup(sentences) is not None:
                    cache.remember(descriptions, sentences)
                    hit = cache.description_rows(descriptions)
        if hit is None:
            return torch.cat([self(descriptions[i:i + batch_size]) for i in range(0, n, batch_size)], dim=0)
        ia, n_per = hit
        cache = self.text_cache
        tok = cache.n_tok[ia].reshape(n, n_per).max(axis=1)
        starts = np.arange(0, n, batch_size)
        L_desc = np.repeat(np.maximum.reduceat(tok, starts), np.diff(np.append(starts, n)))
        rows = torch.from_numpy(ia).to(cache.device)
        version = (bool(self.use_engine_head), self._head_generation) + tuple((t.data_ptr(), t._version) for t in self._head_params())
        x = None
        deferred, self._deferred = getattr(self, "_deferred", None), None  # (the memos are checked synchronously, as in _from_cache)
        try:
            for L in np.unique(L_desc):
                vec = cache.sentence_vectors(self, int(L), version)
                if x is None:
                    x = torch.empty((n * n_per, vec.shape[1]), dtype=vec.dtype, device=vec.device)
                sel = torch.from_numpy(np.nonzero(np.repeat(L_desc == L, n_per))[0]).to(vec.device)
                x.index_copy_(0, sel, vec.index_select(0, rows.index_select(0, sel)))
        finally:
            self._deferred = deferred
        LanguageEncoder.cache_calls += len(starts)
        self._open_call(x.device)
        return self._head_second_half(x, n, self.use_engine_head and x.is_cuda)

    @property
    def device(self):
        return next(self.inter_mlp.parameters()).device


class _EncodeObjectsTrainFn(torch.autograd.Function):
    """Training-mode encode_objects: forward and backward are HIP (t2l_encode_cells_train / _backward). Parameter
    gradients do not flow through autograd: the engine adds them straight into the bound ``.grad`` buffers, so the
    only differentiable input is ``pn_feat`` (``hook`` is a dummy leaf that makes autograd call ``backward``)."""

    @staticmethod
    def forward(ctx, hook, pn_feat, model, packed, p_drop, seed):
        eng = model._engine
        out = eng.encode_cells_train(packed, dropout_p=p_drop, seed=seed)
        ctx.model = model
        ctx.token = model._train_token = object()
        ctx.need_pn = pn_feat is not None and pn_feat.requires_grad
        ctx.pn_engine = bool(model._pn_in_engine)  # features2 came from t2l_pointnet_features_train of a trainable backbone
        ctx.pn_shape = None if pn_feat is None else pn_feat.shape
        return out

    @staticmethod
    def backward(ctx, grad_out):
        model = ctx.model
        if model._train_token is not ctx.token:
            raise T2LError("backward of a stale encode_objects call: the engine keeps the activations of the LAST "
                           "training-mode forward only (the reference's loop does one forward per backward too)")
        gpn = torch.empty(ctx.pn_shape, dtype=torch.float32, device=grad_out.device) if (ctx.need_pn or ctx.pn_engine) else None
        model._engine.encode_cells_backward(grad_out.contiguous().float(), gpn)
        if ctx.pn_engine:
            model._engine.pointnet_backward(gpn)
        return None, (gpn if ctx.need_pn else None), None, None, None, None


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
        self.object_encoder = ObjectEncoderParams(self.embed_dim, known_classes, args, known_colors)
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
        self._pn_weights_version = None
        self._train_generation = 0   # bumped whenever the engine changes parameters/buffers behind torch's back
        self._train_bound = None     # pointer set the engine's training path is bound to
        self._train_grads = {}       # name -> persistent gradient buffer (kept when .grad is set to None): views of _train_flat
        self._train_flat = None
        self._train_flat_layout = None
        self._train_token = None
        self._train_hook = None
        self._pn_in_engine = False   # the last training-mode forward ran the backbone in the engine with gradients bound
        self._pn_train_cells = 0

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

    def encode_text_batches(self, descriptions, batch_size: int):
        """``cat(encode_text(batch) for batch in batches of batch_size)`` in one pass where the text branch can
        (``LanguageEncoder.forward_batches``: the reference's batching is kept, its Python loop is not)."""
        le = self.language_encoder
        if hasattr(le, "forward_batches"):
            return F.normalize(le.forward_batches(descriptions, batch_size))
        return torch.cat([self.encode_text(descriptions[i:i + batch_size]) for i in range(0, len(descriptions), batch_size)], dim=0)

    def encode_objects(self, objects, object_points=None):
        if self.training and torch.is_grad_enabled():
            return self._encode_objects_train(objects, object_points)
        with torch.no_grad():
            return self._encode_objects_eval(objects, object_points)

    def _pn_features(self, object_points, eng: Engine, train: bool = False):
        """PointNet++ ``features2`` [n_objects,256] on the GPU for the published feature mode (class_embed off), or None.
        ``object_points``: per cell EITHER a precomputed [n_i,256] feature array/tensor OR the cell's point batch as the
        reference's dataloader builds it (a PyG ``Batch`` or anything with ``.pos`` / ``.x`` of shape [n_i*256,3], or a
        dict with those keys) — then the engine's PointNet++ kernels run: t2l_pointnet_features in eval mode,
        t2l_pointnet_features_train (``train``) under model.train()."""
        self._pn_in_engine = False
        a = self.args
        if not ("class" in a.use_features and not bool(getattr(a, "class_embed", False))):
            return None
        if object_points is None or any(p is None for p in object_points):
            raise T2LError("class_embed is off: object_points must hold, per cell, PointNet++ features2 [n_i,256] or the "
                           "cell's point batch (.pos/.x [n_i*256,3])")
        dev = self.device

        def field(p, name):
            return p[name] if isinstance(p, dict) else getattr(p, name, None)

        if all(field(p, "pos") is not None for p in object_points if not isinstance(p, (torch.Tensor, np.ndarray))) and \
                not isinstance(object_points[0], (torch.Tensor, np.ndarray)):
            if "color" not in a.use_features:  # ablation of the reference: void all colours (object_encoder.py:87-90)
                xs = [torch.zeros_like(torch.as_tensor(field(p, "x"))) for p in object_points]
            else:
                xs = [torch.as_tensor(field(p, "x")) for p in object_points]
            pos = torch.cat([torch.as_tensor(field(p, "pos")).reshape(-1, 256, 3) for p in object_points]).to(dev, torch.float32)
            rgb = torch.cat([x.reshape(-1, 256, 3) for x in xs]).to(dev, torch.float32)
            counts = [int(torch.as_tensor(field(p, "pos")).shape[0]) // 256 for p in object_points]
            offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
            if train:
                self._pn_in_engine = any(p.requires_grad for p in self.object_encoder.pointnet.lin2.parameters())
                self._pn_train_cells = len(counts)
                return eng.pointnet_features_train(pos.contiguous(), rgb.contiguous(), offs)
            return eng.pointnet_features(pos.contiguous(), rgb.contiguous(), offs)
        return torch.cat([p if isinstance(p, torch.Tensor) else torch.as_tensor(np.asarray(p)) for p in object_points],
                         dim=0).to(dev, torch.float32).reshape(-1, 256)

    # ---- training mode ----------------------------------------------------------------
    def _train_tensors(self):
        """state_dict key -> (live tensor, gradient buffer | None) for what the configuration uses."""
        a = self.args
        ce, co = bool(getattr(a, "class_embed", False)), bool(getattr(a, "color_embed", False))
        skip = []
        if "class" not in a.use_features or ce:
            skip.append("object_encoder.mlp_pointnet.")
        if "class" not in a.use_features or not ce:
            skip.append("object_encoder.class_embedding.")
        if "color" not in a.use_features or co:
            skip.append("object_encoder.color_encoder.")
        if "color" not in a.use_features or not co:
            skip.append("object_encoder.color_embedding.")
        if "class" not in a.use_features or ce:
            skip.append("object_encoder.pointnet.")
        skip += ["object_encoder.pointnet.class_classifier.", "object_encoder.pointnet.color_classifier."]  # features2 is consumed
        if "position" not in a.use_features:
            skip.append("object_encoder.pos_encoder.")
        if "num" not in a.use_features:
            skip.append("object_encoder.num_encoder.")
        out = {}
        live = []
        for n, t in self.state_dict(keep_vars=True).items():
            if not n.startswith(("object_encoder.", "obj_inter_module.")) or n.startswith(tuple(skip)):
                continue
            if n.endswith("num_batches_tracked"):
                continue
            live.append((n, t))
        # Gradient buffers of the engine-stepped parameters are views into ONE flat tensor (64-float aligned): data-parallel
        # training all-reduces it as a single RCCL collective (optim.Adam(group=...)), and zero_grad(set_to_none=True) /
        # model.to() only drop or move references, never the buffers the engine is bound to.
        want = [(n, t) for n, t in live if isinstance(t, nn.Parameter) and t.requires_grad]
        layout = tuple((n, int(t.numel())) for n, t in want)
        dev = want[0][1].device if want else None
        if want and (self._train_flat is None or self._train_flat_layout != layout or self._train_flat.device != dev):
            offs, total = [], 0
            for _, t in want:
                offs.append(total)
                total += (int(t.numel()) + 63) // 64 * 64
            self._train_flat = torch.zeros(total, dtype=torch.float32, device=dev)
            self._train_flat_layout = layout
            self._train_grads = {n: self._train_flat[o:o + t.numel()].view(t.shape) for (n, t), o in zip(want, offs)}
        for n, t in live:
            if isinstance(t, nn.Parameter) and not t.requires_grad:
                out[n] = (t.data, None)  # --pointnet_freeze (object_encoder.py:53-55): forward only
            elif isinstance(t, nn.Parameter):
                g = self._train_grads[n]
                if t.grad is None:
                    t.grad = g  # hand the persistent buffer back (zero_grad(set_to_none=True) only drops the reference)
                    g.zero_()
                elif t.grad.data_ptr() != g.data_ptr():
                    g.copy_(t.grad)  # a gradient assigned from outside keeps its VALUE; the storage stays the flat buffer's
                    t.grad = g
                out[n] = (t.data, g)
            else:
                out[n] = (t, None)
        return out

    def train_flat_grad(self) -> Optional[torch.Tensor]:
        """The flat buffer every engine-stepped parameter's ``.grad`` is a view of (None before the first bind)."""
        return self._train_flat

    def train_engine(self) -> Engine:
        """The engine with the training path bound to the CURRENT parameter / gradient / buffer storage."""
        dev = self.device
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        if self._engine is None or self._engine.device != idx:
            self._engine = Engine(idx)
            self._weights_version = None
            self._pn_weights_version = None
            self._train_bound = None
        tensors = self._train_tensors()
        key = tuple((n, d.data_ptr(), None if g is None else g.data_ptr()) for n, (d, g) in tensors.items())
        if key != self._train_bound:
            a = self.args
            # same model, moved storage (model.to(), re-assigned .grad): the optimizer state must survive the re-bind
            self._engine.set_option("train_keep_adam_state", 1 if self._train_bound is not None else 0)
            self._engine.train_bind(tensors, class_embed=bool(getattr(a, "class_embed", False)),
                                    color_embed=bool(getattr(a, "color_embed", False)), use_features=tuple(a.use_features),
                                    num_layers=a.object_inter_module_num_layers, num_heads=a.object_inter_module_num_heads)
            self._train_bound = key
        _apply_sync_bn(self._engine, getattr(self, "_sync_bn_cfg", None))
        return self._engine

    def sync_batchnorm(self, group=None, enable: bool = True):
        """Data-parallel training with the reference's batch statistics: every BatchNorm1d of the object branch
        (object_encoder.py:41-52,121-149) and the text head's inter_mlp (language_encoder.py:99) normalises over the objects /
        sentences of ALL ranks' batches — 8 ranks x 8 cells then take the step the reference takes on its one batch of 64
        (training/coarse.py:31-58); what torch.nn.SyncBatchNorm.convert_sync_batchnorm does for module-level BatchNorms. Needs an
        initialised process group; ``ContrastiveLoss(gather=True)`` and ``optim.Adam(data_parallel=True)`` complete the step."""
        cfg = (group,) if enable else None
        self._sync_bn_cfg = cfg
        if hasattr(self.language_encoder, "_bind_text_train"):
            self.language_encoder._sync_bn_cfg = cfg

    def _encode_objects_train(self, objects, object_points):
        dev = self.device
        if dev.type != "cuda":
            raise T2LError("encode_objects runs on the MI355X only (model.to('cuda')); there is no CPU fallback")
        eng = self.train_engine()
        self._pn_train_cells = 0
        pn = self._pn_features(object_points, eng, train=True)  # reads the LIVE backbone tensors (t2l_train_bind)
        if any(getattr(o, "_t2l_feat", None) is None for objs in objects for o in objs):
            packed = packing.pack_cells_gpu(eng, objects, self.object_encoder.known_classes,
                                            self.object_encoder.known_colors, dev)
        else:
            packed = packing.to_device(packing.pack_cells(objects, self.object_encoder.known_classes,
                                                          self.object_encoder.known_colors, None), dev)
        if pn is not None:
            if int(pn.shape[0]) != int(packed["offsets"][-1]):
                raise T2LError("object_points must describe exactly the objects of each cell")
            packed["pn_feat"] = pn.detach().contiguous()
        layer = self.obj_inter_module[0] if len(self.obj_inter_module) else None
        p_drop = float(layer.dropout.p) if layer is not None else 0.0  # nn.TransformerEncoderLayer default 0.1
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())       # torch.manual_seed governs the masks
        if self._train_hook is None or self._train_hook.device != dev:
            self._train_hook = torch.zeros(1, device=dev, requires_grad=True)
        out = _EncodeObjectsTrainFn.apply(self._train_hook, pn, self, packed, p_drop, seed)
        used = {k for k, _, _ in self._train_bound}
        for name, m in self.named_modules():  # BatchNorm1d.train() side effect the engine does not see (int64)
            if isinstance(m, nn.BatchNorm1d) and m.num_batches_tracked is not None and name + ".running_mean" in used:
                if name.startswith("object_encoder.pointnet."):
                    m.num_batches_tracked += self._pn_train_cells  # one backbone call per cell (object_encoder.py:92-95)
                else:
                    m.num_batches_tracked += 1
        self._train_generation += 1  # running statistics moved: the eval-path weights must be re-folded
        return out

    def _encode_objects_eval(self, objects, object_points=None):
        dev = self.device
        if dev.type != "cuda":
            raise T2LError("encode_objects runs on the MI355X only (model.to('cuda')); there is no CPU fallback")
        eng = self.engine()
        pn = self._pn_features(object_points, eng)
        if any(getattr(o, "_t2l_feat", None) is None for objs in objects for o in objs):
            # raw points not reduced yet: one HBM pass on the GPU (t2l_reduce_objects) instead of three numpy
            # reductions per object on the host (what the reference redoes on every call)
            packed = packing.pack_cells_gpu(eng, objects, self.object_encoder.known_classes,
                                            self.object_encoder.known_colors, dev)
        else:
            packed = packing.to_device(packing.pack_cells(objects, self.object_encoder.known_classes,
                                                          self.object_encoder.known_colors, None), dev)
        if pn is not None:
            if int(pn.shape[0]) != int(packed["offsets"][-1]):
                raise T2LError("object_points must describe exactly the objects of each cell")
            packed["pn_feat"] = pn.detach().contiguous()
        return eng.encode_cells(packed)

    @torch.no_grad()
    def encode_cell_set(self, cell_set, points: bool = False, transform: str = "fixed", seed: int = 0,
                        chunk_cells: int = 4096) -> torch.Tensor:
        """Eval-mode ``encode_objects`` over a whole ``packing.PackedCellSet`` -> f32[n_cells,256] unit rows on the GPU: what
        ``eval_epoch``'s database loop (training/coarse.py:99-113) computes with one ``encode_objects`` call per ``args.batch_size``
        cells (1 by default, evaluation/args.py:11), here from the flattened dataset in chunks of ``chunk_cells`` cells. In eval mode
        a cell's embedding is a function of that cell alone (no batch statistics, no dropout), so the chunking cannot change it —
        ``tests/test_gpu_pipeline.py::test_eval_epoch_is_independent_of_batching`` holds that bit for bit.
        ``points``: the published feature mode (class_embed off) — per chunk, FixedPoints(256) under ``transform`` on the GPU
        (t2l_sample_object_points, draws keyed on (seed, chunk, object, point)) -> PointNet++ (t2l_pointnet_features) ->
        ``pn_feat``. The per-object reductions (a1) run once per dataset and device (``PackedCellSet.reduced``)."""
        dev = self.device
        if dev.type != "cuda":
            raise T2LError("encode_cell_set runs on the MI355X only (model.to('cuda')); there is no CPU fallback")
        a = self.args
        eng = self.engine()
        need_pn = "class" in a.use_features and not bool(getattr(a, "class_embed", False))
        if need_pn and not points:
            raise T2LError("class_embed is off: the cell dataset must deliver point batches (object_points='sample')")
        oe = self.object_encoder
        d = cell_set.on_device(dev)
        red = cell_set.reduced(eng, dev, oe.known_colors)
        cls = cell_set.class_idx_on(dev, oe.known_classes)
        n_cells = cell_set.n_cells
        out = torch.empty((n_cells, self.embed_dim), dtype=torch.float32, device=dev)
        offs_h = cell_set.offsets
        chunk_cells = max(1, int(chunk_cells))
        for ci, lo in enumerate(range(0, n_cells, chunk_cells)):
            hi = min(n_cells, lo + chunk_cells)
            o_lo, o_hi = int(offs_h[lo]), int(offs_h[hi])
            packed = {"offsets": (d["offsets"][lo:hi + 1] - o_lo) if o_lo else d["offsets"][lo:hi + 1],
                      "class_idx": cls[o_lo:o_hi], "color_idx": red["color_idx"][o_lo:o_hi], "rgb": red["rgb"][o_lo:o_hi],
                      "center": red["center"][o_lo:o_hi], "n_pts": red["n_pts"][o_lo:o_hi]}
            if need_pn:
                rgb_src = d["rgb"]
                if "color" not in a.use_features:  # ablation of the reference: void all colours (object_encoder.py:87-90)
                    rgb_src = d.get("rgb_zero")
                    if rgb_src is None:
                        rgb_src = d["rgb_zero"] = torch.zeros_like(d["rgb"])
                pos, col = eng.sample_object_points(d["xyz"], rgb_src, d["point_offsets"][o_lo:o_hi + 1],
                                                    (int(seed) + 0x9E3779B1 * ci) & 0xFFFFFFFF, transform=transform)
                packed["pn_feat"] = eng.pointnet_features(pos, col, offs_h[lo:hi + 1] - o_lo)
            out[lo:hi] = eng.encode_cells(packed)
        return out

    # ---- engine plumbing --------------------------------------------------------------------------------
    def engine(self) -> Engine:
        """The engine holding the CURRENT object-branch weights. Checked on every call — an optimizer steps parameters in place
        behind the module's back — but against a cached parameter list: (data_ptr, _version) over it costs ~24 us where walking
        ``state_dict(keep_vars=True)`` cost ~200 us per ``encode_objects`` call (as much host time as the GPU needs for eight
        cells). The list is rebuilt whenever the module tree may have changed hands: ``train()/eval()``, ``load_state_dict``,
        ``to()/cuda()/float()`` (``_apply``)."""
        dev = self.device
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        if self._engine is None or self._engine.device != idx:
            self._engine = Engine(idx)
            self._weights_version = None
            self._pn_weights_version = None
            self._train_bound = None
        params = self._obj_param_list
        if params is None:
            params = self._obj_param_list = list(self._object_params())
        version = (self._train_generation, tuple([(p.data_ptr(), p._version) for p in params]))
        if version != self._weights_version:
            self.sync_weights()
            self._weights_version = version
        return self._engine

    _obj_param_list = None

    def train(self, mode: bool = True):
        self._obj_param_list = None
        return super().train(mode)

    def _apply(self, fn, *args, **kwargs):
        self._obj_param_list = None
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self._obj_param_list = None
        return super().load_state_dict(*args, **kwargs)

    def _object_params(self):
        for n, p in self.state_dict(keep_vars=True).items():
            if n.startswith(("object_encoder.", "obj_inter_module.")):
                yield p

    def _pn_version(self):
        return tuple((p.data_ptr(), p._version) for n, p in self.state_dict(keep_vars=True).items()
                     if n.startswith("object_encoder.pointnet."))

    def sync_weights(self):
        """Fold BatchNorm, re-lay out and upload the object-branch weights (t2l_load_weights)."""
        a = self.args
        self._pn_weights_version = self._pn_version()
        sd = {k: v for k, v in self.state_dict().items() if k.startswith(("object_encoder.", "obj_inter_module."))}
        self._engine.load_weights(sd, class_embed=bool(getattr(a, "class_embed", False)),
                                  color_embed=bool(getattr(a, "color_embed", False)),
                                  use_features=tuple(a.use_features), num_layers=a.object_inter_module_num_layers,
                                  num_heads=a.object_inter_module_num_heads)
