"""ctypes binding of libt2l.so (include/t2l.h) — the only door between Python and the HIP kernels.

PyTorch is used here for plumbing only: device allocations (``torch.empty(..., device='cuda')``), the
current HIP stream handle and, in ``sharded.py``, ``torch.distributed`` (RCCL). No arithmetic of the
hot path runs in PyTorch, and there is NO CPU fallback: importing this module without the built
library, or calling it with CPU tensors, raises.
"""
from __future__ import annotations

import ctypes as C
import os
import os.path as osp
from typing import Dict, Optional, Tuple

import numpy as np
import torch

EMBED_DIM = 256
OBJECT_SIZE = 28
MAX_TOPK = 26
# (T2L_LIB: a dev build beside the shipped one — the stamped or an experiment library of csrc/Makefile; never a fallback)
_LIB_PATH = os.environ.get("T2L_LIB") or osp.join(osp.dirname(osp.abspath(__file__)), "libt2l.so")


class T2LError(RuntimeError):
    pass


class _WeightDesc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("numel", C.c_int64)]


class _ModelConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("class_embed", "color_embed", "use_class", "use_color", "use_position",
                                         "use_num", "num_layers", "num_heads")]


class _TrainTensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("grad", C.c_void_p), ("numel", C.c_int64)]


class _PackedCells(C.Structure):
    _fields_ = [("n_cells", C.c_int32), ("n_objects", C.c_int32), ("offsets", C.c_void_p), ("class_idx", C.c_void_p),
                ("color_idx", C.c_void_p), ("rgb", C.c_void_p), ("center", C.c_void_p), ("n_pts", C.c_void_p),
                ("pn_feat", C.c_void_p)]


EXPORTS = {
    "t2l_abi_version": (C.c_int, []),
    "t2l_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "t2l_destroy": (None, [C.c_void_p]),
    "t2l_last_error": (C.c_char_p, [C.c_void_p]),
    "t2l_load_weights": (C.c_int, [C.c_void_p, C.POINTER(_WeightDesc), C.c_int32, C.POINTER(_ModelConfig)]),
    "t2l_encode_cells": (C.c_int, [C.c_void_p, C.POINTER(_PackedCells), C.c_void_p, C.c_void_p]),
    "t2l_sample_object_points": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_uint32, C.c_int32, C.c_float,
                                           C.c_void_p, C.c_void_p, C.c_void_p]),
    "t2l_pointnet_features": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "t2l_reduce_objects": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                     C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "t2l_db_set": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "t2l_db_rows": (C.c_int64, [C.c_void_p]),
    "t2l_search": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "t2l_search_join": (C.c_int, [C.c_void_p, C.c_void_p]),
    "t2l_search_many": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "t2l_search_ordered": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "t2l_merge_topk": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                 C.c_void_p, C.c_void_p]),
    "t2l_pack_pairs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "t2l_merge_pairs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                  C.c_void_p]),
    "t2l_merge_gathered": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                     C.c_void_p]),
    "t2l_search_fallbacks": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "t2l_search_rescored": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "t2l_search_counters": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "t2l_contrastive_loss": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p]),
    "t2l_text_head_load_weights": (C.c_int, [C.c_void_p, C.POINTER(_WeightDesc), C.c_int32, C.c_char_p]),
    "t2l_text_head": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "t2l_text_inter": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "t2l_fine_load_weights": (C.c_int, [C.c_void_p, C.POINTER(_WeightDesc), C.c_int32, C.POINTER(_ModelConfig)]),
    "t2l_fine_encode_objects": (C.c_int, [C.c_void_p, C.POINTER(_PackedCells), C.c_void_p, C.c_void_p]),
    "t2l_fine_match": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                 C.c_void_p]),
    "t2l_train_bind": (C.c_int, [C.c_void_p, C.POINTER(_TrainTensor), C.c_int32, C.POINTER(_ModelConfig)]),
    "t2l_encode_cells_train": (C.c_int, [C.c_void_p, C.POINTER(_PackedCells), C.c_float, C.c_uint32, C.c_void_p, C.c_void_p]),
    "t2l_encode_cells_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "t2l_text_train_bind": (C.c_int, [C.c_void_p, C.POINTER(_TrainTensor), C.c_int32, C.c_char_p]),
    "t2l_text_head_train": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_uint32, C.c_void_p, C.c_void_p]),
    "t2l_text_head_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "t2l_pointnet_features_train": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "t2l_pointnet_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "t2l_text_adam_step": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "t2l_text_zero_grad": (C.c_int, [C.c_void_p, C.c_void_p]),
    "t2l_text_adam_state": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                      C.c_void_p]),
    "t2l_train_sync_bn_doubles": (C.c_int64, []),
    "t2l_train_sync_bn": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "t2l_zero_grad": (C.c_int, [C.c_void_p, C.c_void_p]),
    "t2l_adam_step": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "t2l_adam_state": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                 C.c_void_p]),
    "t2l_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_double]),
    "t2l_kernel_stats": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
}

_lib = None


def load_library() -> C.CDLL:
    """dlopen libt2l.so and declare every symbol of include/t2l.h. Raises if the library is absent:
    there is deliberately no fallback (build it with ``python -c 'import __graft_entry__ as g; g.build()'``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not osp.exists(_LIB_PATH):
        raise T2LError(f"{_LIB_PATH} is missing: the HIP extension is not built and there is no CPU fallback")
    lib = C.CDLL(_LIB_PATH)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _stream_ptr(device=None) -> int:
    """torch's current stream ON THE ENGINE'S DEVICE (the current device may be another GPU of the same process)."""
    return int(torch.cuda.current_stream(device).cuda_stream)


def _dev_ptr(t: Optional[torch.Tensor], dtype, name: str) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise T2LError(f"{name}: expected a CUDA (HIP) tensor; the engine has no CPU path")
    if t.dtype != dtype or not t.is_contiguous():
        raise T2LError(f"{name}: expected contiguous {dtype}, got {t.dtype} contiguous={t.is_contiguous()}")
    return t.data_ptr()


class Engine:
    """One context per GPU (one process per GPU)."""

    def _ptr(self, t: Optional[torch.Tensor], dtype, name: str) -> Optional[int]:
        """Device pointer of a contiguous tensor of ``dtype`` that lives on THIS engine's GPU (anything else raises)."""
        if t is not None and t.is_cuda and t.device.index != self.device:
            raise T2LError(f"{name}: tensor is on cuda:{t.device.index}, this engine drives cuda:{self.device}")
        return _dev_ptr(t, dtype, name)

    def __init__(self, device: Optional[int] = None):
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise T2LError("no MI355X visible: the engine needs a GPU (there is no CPU fallback)")
        self.device = torch.cuda.current_device() if device is None else int(device)
        h = C.c_void_p()
        rc = self.lib.t2l_create(C.byref(h), self.device)
        if rc != 0:
            raise T2LError(f"t2l_create failed with {rc}")
        self._h = h
        self.db_owner = None
        self.db_generation = 0
        self._lane_keepalive = []

    def close(self):
        if getattr(self, "_h", None):
            self.lib.t2l_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != 0:
            raise T2LError(f"libt2l error {rc}: {self.lib.t2l_last_error(self._h).decode()}")

    # ------------------------------------------------------------------ weights
    def load_weights(self, state_dict: Dict[str, object], class_embed: bool, color_embed: bool,
                     use_features=("class", "color", "position", "num"), num_layers: int = 2, num_heads: int = 4):
        """state_dict: name -> torch.Tensor | np.ndarray (fp32), keys as in the reference checkpoint."""
        keep, descs = [], []
        for name, v in state_dict.items():
            if not (name.startswith("object_encoder.") or name.startswith("obj_inter_module.")):
                continue
            if name.endswith("num_batches_tracked"):
                continue
            a = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            a = np.ascontiguousarray(a, dtype=np.float32)
            keep.append(a)
            descs.append(_WeightDesc(name.encode(), a.ctypes.data, a.size))
        arr = (_WeightDesc * len(descs))(*descs)
        cfg = _ModelConfig(int(class_embed), int(color_embed), int("class" in use_features),
                           int("color" in use_features), int("position" in use_features), int("num" in use_features),
                           int(num_layers), int(num_heads))
        self._check(self.lib.t2l_load_weights(self._h, arr, len(descs), C.byref(cfg)))

    # ------------------------------------------------------------------ per-object reductions (a1)
    def reduce_objects(self, xyz: torch.Tensor, rgb: torch.Tensor, point_offsets: torch.Tensor,
                       color_centers: np.ndarray, color_rows: np.ndarray) -> Dict[str, torch.Tensor]:
        """xyz, rgb f32[n_points,3] (GPU, objects concatenated), point_offsets i64[n_objects+1] (HOST: tensor or ndarray)
        -> dict(rgb f32[n,3], center f32[n,3], n_pts f32[n], color_idx i32[n]) on the GPU."""
        po = point_offsets.cpu().numpy() if isinstance(point_offsets, torch.Tensor) else np.asarray(point_offsets)
        po = np.ascontiguousarray(po, dtype=np.int64)
        n = int(po.size) - 1
        dev = xyz.device
        out = {"rgb": torch.empty((n, 3), dtype=torch.float32, device=dev),
               "center": torch.empty((n, 3), dtype=torch.float32, device=dev),
               "n_pts": torch.empty((n,), dtype=torch.float32, device=dev),
               "color_idx": torch.empty((n,), dtype=torch.int32, device=dev)}
        cc = np.ascontiguousarray(color_centers, dtype=np.float32)
        cr = np.ascontiguousarray(color_rows, dtype=np.int32)
        self._check(self.lib.t2l_reduce_objects(
            self._h, self._ptr(xyz, torch.float32, "xyz"), self._ptr(rgb, torch.float32, "rgb"),
            po.ctypes.data, n, cc.ctypes.data, cr.ctypes.data, len(cr),
            out["rgb"].data_ptr(), out["center"].data_ptr(), out["n_pts"].data_ptr(), out["color_idx"].data_ptr(),
            _stream_ptr(self.device)))
        return out

    # ------------------------------------------------------------------ PointNet++ backbone (a3)
    def sample_object_points(self, xyz: torch.Tensor, rgb: torch.Tensor, point_offsets: torch.Tensor, seed: int = 0,
                             transform: str = "fixed", rotate_deg: float = 120.0):
        """The dataloader's point batches on the GPU: xyz, rgb f32[n_points,3], point_offsets i64[n_objects+1] (all on the GPU)
        -> (pos f32[n_objects,256,3], rgb f32[n_objects,256,3]). ``transform`` (packing.POINT_TRANSFORMS): "fixed" =
        FixedPoints(256) only (`--no_pc_augment`, the published commands), "normalize" = + NormalizeScale, "rotate_normalize" =
        + RandomRotate(rotate_deg, axis=2) before it (training without the flag)."""
        from .packing import POINT_TRANSFORMS

        if transform not in POINT_TRANSFORMS:
            raise T2LError(f"sample_object_points: transform must be one of {sorted(POINT_TRANSFORMS)}, got {transform!r}")
        n = int(point_offsets.numel()) - 1
        pos = torch.empty((max(n, 0), 256, 3), dtype=torch.float32, device=xyz.device)
        col = torch.empty_like(pos)
        self._check(self.lib.t2l_sample_object_points(self._h, self._ptr(xyz, torch.float32, "xyz"), self._ptr(rgb, torch.float32, "rgb"),
                                                      self._ptr(point_offsets, torch.int64, "point_offsets"), n, int(seed) & 0xFFFFFFFF,
                                                      POINT_TRANSFORMS[transform], float(rotate_deg), pos.data_ptr(), col.data_ptr(),
                                                      _stream_ptr(self.device)))
        return pos, col

    def pointnet_features(self, pos: torch.Tensor, rgb: torch.Tensor, cell_offsets) -> torch.Tensor:
        """pos, rgb f32[n_objects,256,3] on the GPU, cell_offsets i32[n_cells+1] (HOST) -> features2 f32[n_objects,256]."""
        co = cell_offsets.cpu().numpy() if isinstance(cell_offsets, torch.Tensor) else np.asarray(cell_offsets)
        co = np.ascontiguousarray(co, dtype=np.int32)
        n = int(pos.shape[0])
        if pos.shape != (n, 256, 3) or rgb.shape != (n, 256, 3) or int(co[-1]) != n:
            raise T2LError(f"pointnet_features: expected [n,256,3] points and offsets ending at n, got {tuple(pos.shape)}, "
                           f"{tuple(rgb.shape)}, {int(co[-1])}")
        out = torch.empty((n, EMBED_DIM), dtype=torch.float32, device=pos.device)
        self._check(self.lib.t2l_pointnet_features(self._h, self._ptr(pos, torch.float32, "pos"), self._ptr(rgb, torch.float32, "rgb"),
                                                   co.ctypes.data, len(co) - 1, out.data_ptr(), _stream_ptr(self.device)))
        return out

    # ------------------------------------------------------------------ cell encoding
    def encode_cells(self, packed: Dict[str, torch.Tensor]) -> torch.Tensor:
        """packed: offsets i32[B+1], class_idx/color_idx i32[n], rgb/center f32[n,3], n_pts f32[n],
        optional pn_feat f32[n,256] — all on the GPU. Returns f32[B,256] unit rows."""
        offsets = packed["offsets"]
        n_cells = int(offsets.numel()) - 1
        n_obj = int(packed["n_pts"].numel()) if packed.get("n_pts") is not None else int(packed["class_idx"].numel())
        out = torch.empty((max(n_cells, 0), EMBED_DIM), dtype=torch.float32, device=offsets.device)
        if n_cells <= 0:
            return out
        pc = _PackedCells(
            n_cells, n_obj, self._ptr(offsets, torch.int32, "offsets"),
            self._ptr(packed.get("class_idx"), torch.int32, "class_idx"),
            self._ptr(packed.get("color_idx"), torch.int32, "color_idx"),
            self._ptr(packed.get("rgb"), torch.float32, "rgb"), self._ptr(packed.get("center"), torch.float32, "center"),
            self._ptr(packed.get("n_pts"), torch.float32, "n_pts"),
            self._ptr(packed.get("pn_feat"), torch.float32, "pn_feat"))
        self._check(self.lib.t2l_encode_cells(self._h, C.byref(pc), out.data_ptr(), _stream_ptr(self.device)))
        return out

    # ------------------------------------------------------------------ text head after T5 (f-4a)
    def text_head_load_weights(self, state_dict: Dict[str, object], prefix: str = "language_encoder."):
        """state_dict: name -> tensor / ndarray holding ``<prefix>intra_module.0.*`` and ``<prefix>inter_mlp.0.*`` (fp32)."""
        keep, descs = [], []
        for name, v in state_dict.items():
            if not name.startswith((prefix + "intra_module.", prefix + "inter_mlp.", prefix + "inter_module.")) or name.endswith("num_batches_tracked"):
                continue
            a = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            a = np.ascontiguousarray(a, dtype=np.float32)
            keep.append(a)
            descs.append(_WeightDesc(name.encode(), a.ctypes.data, a.size))
        if not descs:
            raise T2LError(f"text_head_load_weights: no tensors under {prefix}intra_module / {prefix}inter_mlp")
        arr = (_WeightDesc * len(descs))(*descs)
        self._check(self.lib.t2l_text_head_load_weights(self._h, arr, len(descs), prefix.encode()))
        self._text_head_dim = int(state_dict[prefix + "inter_mlp.0.0.weight"].shape[0])

    def text_head(self, hidden: torch.Tensor, check: bool = True):
        """hidden f32[n_sentences, n_tokens, 1024] (T5 last_hidden_state) on the GPU -> f32[n_sentences, D] = inter_mlp(max over
        tokens(intra_module(hidden))) (models/language_encoder.py:127-135). ``check``: read the overflow flag (one 4-byte
        copy + a stream sync) and return ``(out, overflowed)``; otherwise ``(out, flag tensor)``."""
        if hidden.dim() != 3 or hidden.shape[2] != 1024:
            raise T2LError(f"text_head: expected [n_sentences, n_tokens, 1024], got {tuple(hidden.shape)}")
        S, L = int(hidden.shape[0]), int(hidden.shape[1])
        if getattr(self, "_text_head_dim", None) is None:
            raise T2LError("text_head: call text_head_load_weights first")
        out = torch.empty((S, self._text_head_dim), dtype=torch.float32, device=hidden.device)
        flag = torch.zeros((1,), dtype=torch.int32, device=hidden.device)
        self._check(self.lib.t2l_text_head(self._h, self._ptr(hidden, torch.float32, "hidden"), S, L, out.data_ptr(), flag.data_ptr(),
                                           _stream_ptr(self.device)))
        return (out, bool(flag.item())) if check else (out, flag)

    def text_inter(self, sent: torch.Tensor, n_descriptions: int, check: bool = True):
        """sent f32[n_descriptions * S, 256] (t2l_text_head's output, description-major) -> f32[n_descriptions, 256] =
        max over the S sentences of x + inter_module[0](x) (models/language_encoder.py:137-147). Returns (out, overflowed) like
        ``text_head``."""
        if sent.dim() != 2 or sent.shape[1] != 256 or n_descriptions <= 0 or sent.shape[0] % n_descriptions:
            raise T2LError(f"text_inter: expected [n_descriptions * S, 256], got {tuple(sent.shape)} for {n_descriptions} descriptions")
        S = int(sent.shape[0]) // n_descriptions
        out = torch.empty((n_descriptions, 256), dtype=torch.float32, device=sent.device)
        flag = torch.zeros((1,), dtype=torch.int32, device=sent.device)
        self._check(self.lib.t2l_text_inter(self._h, self._ptr(sent, torch.float32, "sent"), n_descriptions, S, out.data_ptr(), flag.data_ptr(),
                                            _stream_ptr(self.device)))
        return (out, bool(flag.item())) if check else (out, flag)

    # ------------------------------------------------------------------ fine stage (f-1)
    def fine_load_weights(self, state_dict: Dict[str, object], class_embed: bool, color_embed: bool,
                          use_features=("class", "color", "position", "num"), num_layers: int = 2, num_heads: int = 4):
        keep, descs = [], []
        for name, v in state_dict.items():
            if name.startswith("language_encoder.") or name.endswith("num_batches_tracked"):
                continue
            a = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            a = np.ascontiguousarray(a, dtype=np.float32)
            keep.append(a)
            descs.append(_WeightDesc(name.encode(), a.ctypes.data, a.size))
        arr = (_WeightDesc * len(descs))(*descs)
        cfg = _ModelConfig(int(class_embed), int(color_embed), int("class" in use_features), int("color" in use_features),
                           int("position" in use_features), int("num" in use_features), int(num_layers), int(num_heads))
        self._check(self.lib.t2l_fine_load_weights(self._h, arr, len(descs), C.byref(cfg)))

    def fine_encode_objects(self, packed: Dict[str, torch.Tensor]) -> torch.Tensor:
        """packed cells of exactly 16 objects each -> f32[n_cells,16,128] unit-row object descriptors."""
        pc = self._packed_struct(packed)
        out = torch.empty((pc.n_cells, 16, 128), dtype=torch.float32, device=packed["offsets"].device)
        self._check(self.lib.t2l_fine_encode_objects(self._h, C.byref(pc), out.data_ptr(), _stream_ptr(self.device)))
        return out

    def fine_match(self, cell_desc: torch.Tensor, hint_desc: torch.Tensor, cell_index: Optional[torch.Tensor] = None,
                   hint_index: Optional[torch.Tensor] = None) -> torch.Tensor:
        """cell_desc f32[*,16,128], hint_desc f32[*,n_hints,128], optional i32[n_pairs] row indices -> offsets f32[n_pairs,2]."""
        n_pairs = int(cell_index.numel()) if cell_index is not None else int(cell_desc.shape[0])
        if hint_index is not None and int(hint_index.numel()) != n_pairs:
            raise T2LError("fine_match: cell_index and hint_index must have one entry per pair")
        if cell_index is None and hint_index is None and int(hint_desc.shape[0]) != n_pairs:
            raise T2LError("fine_match: without index arrays cell_desc and hint_desc must have one row block per pair")
        if cell_desc.shape[1:] != (16, 128) or hint_desc.shape[2] != 128:
            raise T2LError(f"fine_match: expected [*,16,128] and [*,n_hints,128], got {tuple(cell_desc.shape)}, {tuple(hint_desc.shape)}")
        out = torch.empty((n_pairs, 2), dtype=torch.float32, device=cell_desc.device)
        self._check(self.lib.t2l_fine_match(self._h, self._ptr(cell_desc, torch.float32, "cell_desc"),
                                            self._ptr(cell_index, torch.int32, "cell_index"),
                                            self._ptr(hint_desc, torch.float32, "hint_desc"),
                                            self._ptr(hint_index, torch.int32, "hint_index"), n_pairs, int(hint_desc.shape[1]),
                                            out.data_ptr(), _stream_ptr(self.device)))
        return out

    # ------------------------------------------------------------------ training step (a9)
    def _packed_struct(self, packed: Dict[str, torch.Tensor]) -> "_PackedCells":
        offsets = packed["offsets"]
        n_cells = int(offsets.numel()) - 1
        n_obj = int(packed["n_pts"].numel()) if packed.get("n_pts") is not None else int(packed["class_idx"].numel())
        return _PackedCells(
            n_cells, n_obj, self._ptr(offsets, torch.int32, "offsets"),
            self._ptr(packed.get("class_idx"), torch.int32, "class_idx"),
            self._ptr(packed.get("color_idx"), torch.int32, "color_idx"),
            self._ptr(packed.get("rgb"), torch.float32, "rgb"), self._ptr(packed.get("center"), torch.float32, "center"),
            self._ptr(packed.get("n_pts"), torch.float32, "n_pts"),
            self._ptr(packed.get("pn_feat"), torch.float32, "pn_feat"))

    def train_bind(self, tensors: Dict[str, Tuple[torch.Tensor, Optional[torch.Tensor]]], class_embed: bool,
                   color_embed: bool, use_features=("class", "color", "position", "num"), num_layers: int = 2,
                   num_heads: int = 4):
        """tensors: state_dict key -> (live fp32 CUDA tensor, its gradient buffer or None for BatchNorm buffers).
        The engine keeps the POINTERS (no copies): keep the tensors alive and re-bind if they are re-allocated."""
        descs, keep = [], []
        for name, (data, grad) in tensors.items():
            keep.append((data, grad))
            descs.append(_TrainTensor(name.encode(), self._ptr(data, torch.float32, name),
                                      self._ptr(grad, torch.float32, name + ".grad"), data.numel()))
        arr = (_TrainTensor * len(descs))(*descs)
        cfg = _ModelConfig(int(class_embed), int(color_embed), int("class" in use_features),
                           int("color" in use_features), int("position" in use_features), int("num" in use_features),
                           int(num_layers), int(num_heads))
        self._check(self.lib.t2l_train_bind(self._h, arr, len(descs), C.byref(cfg)))
        self._train_keepalive = keep

    def encode_cells_train(self, packed: Dict[str, torch.Tensor], dropout_p: float = 0.1, seed: int = 0) -> torch.Tensor:
        pc = self._packed_struct(packed)
        out = torch.empty((pc.n_cells, EMBED_DIM), dtype=torch.float32, device=packed["offsets"].device)
        self._check(self.lib.t2l_encode_cells_train(self._h, C.byref(pc), float(dropout_p), int(seed) & 0xFFFFFFFF,
                                                    out.data_ptr(), _stream_ptr(self.device)))
        self._train_inputs = packed  # the device arrays must outlive the backward call
        return out

    def encode_cells_backward(self, grad_emb: torch.Tensor, grad_pn_feat: Optional[torch.Tensor] = None):
        self._check(self.lib.t2l_encode_cells_backward(self._h, self._ptr(grad_emb, torch.float32, "grad_emb"),
                                                       self._ptr(grad_pn_feat, torch.float32, "grad_pn_feat"),
                                                       _stream_ptr(self.device)))

    # ------------------------------------------------------------------ the text head in training mode (f-4)
    def text_train_bind(self, tensors: Dict[str, Tuple[torch.Tensor, Optional[torch.Tensor]]], prefix: str = "language_encoder."):
        """tensors: ``<prefix>intra_module.0.* / inter_mlp.0.* / inter_module.0.*`` -> (live fp32 CUDA tensor, its .grad buffer or None
        for the BatchNorm running buffers). POINTERS are kept, not copies."""
        descs, keep = [], []
        for name, (data, grad) in tensors.items():
            keep.append((data, grad))
            descs.append(_TrainTensor(name.encode(), self._ptr(data, torch.float32, name),
                                      self._ptr(grad, torch.float32, name + ".grad"), data.numel()))
        arr = (_TrainTensor * len(descs))(*descs)
        self._check(self.lib.t2l_text_train_bind(self._h, arr, len(descs), prefix.encode()))
        self._text_train_keepalive = keep

    def text_head_train(self, hidden: torch.Tensor, n_descriptions: int, dropout_p: float = 0.1, seed: int = 0) -> torch.Tensor:
        """hidden f32[n_sentences, n_tokens, 1024] -> f32[n_descriptions, 256] (not normalised): LanguageEncoder.forward after T5 under
        model.train() (models/language_encoder.py:127-147); activations stay in the context for ``text_head_backward``."""
        if hidden.dim() != 3 or hidden.shape[2] != 1024:
            raise T2LError(f"text_head_train: expected [n_sentences, n_tokens, 1024], got {tuple(hidden.shape)}")
        out = torch.empty((int(n_descriptions), 256), dtype=torch.float32, device=hidden.device)
        self._check(self.lib.t2l_text_head_train(self._h, self._ptr(hidden, torch.float32, "hidden"), int(hidden.shape[0]), int(hidden.shape[1]),
                                                 int(n_descriptions), float(dropout_p), int(seed) & 0xFFFFFFFF, out.data_ptr(),
                                                 _stream_ptr(self.device)))
        self._text_train_input = hidden  # must outlive the backward call
        return out

    def text_head_backward(self, grad_out: torch.Tensor):
        self._check(self.lib.t2l_text_head_backward(self._h, self._ptr(grad_out, torch.float32, "grad_out"), _stream_ptr(self.device)))

    def pointnet_features_train(self, pos: torch.Tensor, rgb: torch.Tensor, cell_offsets) -> torch.Tensor:
        """The PointNet++ backbone under model.train() (per-cell BatchNorm statistics, running statistics updated once per
        cell): same arguments as pointnet_features; needs the object_encoder.pointnet.* tensors bound with gradients."""
        co = cell_offsets.cpu().numpy() if isinstance(cell_offsets, torch.Tensor) else np.asarray(cell_offsets)
        co = np.ascontiguousarray(co, dtype=np.int32)
        n = int(pos.shape[0])
        if pos.shape != (n, 256, 3) or rgb.shape != (n, 256, 3) or int(co[-1]) != n:
            raise T2LError(f"pointnet_features_train: expected [n,256,3] points and offsets ending at n, got {tuple(pos.shape)}, "
                           f"{tuple(rgb.shape)}, {int(co[-1])}")
        out = torch.empty((n, EMBED_DIM), dtype=torch.float32, device=pos.device)
        self._check(self.lib.t2l_pointnet_features_train(self._h, self._ptr(pos, torch.float32, "pos"),
                                                         self._ptr(rgb, torch.float32, "rgb"), co.ctypes.data, len(co) - 1,
                                                         out.data_ptr(), _stream_ptr(self.device)))
        self._pn_train_inputs = (pos, rgb)
        return out

    def pointnet_backward(self, grad_features2: torch.Tensor):
        self._check(self.lib.t2l_pointnet_backward(self._h, self._ptr(grad_features2, torch.float32, "grad_features2"),
                                                   _stream_ptr(self.device)))

    def train_sync_bn(self, group=None, enable: bool = True):
        """Cross-rank BatchNorm statistics for data-parallel training (t2l.h: t2l_train_sync_bn): every BatchNorm of the object branch
        and of the text head's inter_mlp normalises over the rows of ALL ranks of ``group`` — the reference's single-process batch
        (training/coarse.py:31-58) split over GPUs. The sums travel through ``torch.distributed.all_reduce`` on the current stream (RCCL
        on the node; gloo stages through the host). ``enable=False`` returns to per-rank statistics."""
        if not enable:
            self._check(self.lib.t2l_train_sync_bn(self._h, None, 0, None, None))
            self._sync_bn = None
            return
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            raise T2LError("train_sync_bn needs an initialised torch.distributed process group")
        n = int(self.lib.t2l_train_sync_bn_doubles())
        buf = torch.zeros(n, dtype=torch.float64, device=torch.device("cuda", self.device))
        base = buf.data_ptr()
        calls = [0]

        def _sum(_user, ptr, count, _stream):
            # (the engine enqueues on torch's current stream of its device, and so does all_reduce: stream order is the dependency)
            try:
                off = (int(ptr) - base) // 8
                view = buf[off:off + int(count)]
                if dist.get_backend(group) != "nccl":  # gloo (several processes on one GPU in the tests): staged through the host
                    h = view.cpu()
                    dist.all_reduce(h, group=group)
                    view.copy_(h)
                else:
                    dist.all_reduce(view, group=group)
                calls[0] += 1
                return 0
            except Exception:  # a Python exception cannot cross the C frame: the enclosing engine call fails with T2L_ESTATE
                import traceback
                traceback.print_exc()
                return 1

        cb = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)(_sum)
        self._check(self.lib.t2l_train_sync_bn(self._h, buf.data_ptr(), n, C.cast(cb, C.c_void_p), None))
        self._sync_bn = (buf, cb, calls)  # the library keeps raw pointers to both

    def sync_bn_calls(self) -> int:
        """Cross-rank sums issued since train_sync_bn (one per BatchNorm stage and direction)."""
        return 0 if getattr(self, "_sync_bn", None) is None else self._sync_bn[2][0]

    def zero_grad(self):
        self._check(self.lib.t2l_zero_grad(self._h, _stream_ptr(self.device)))

    def adam_step(self, lr: float, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8):
        self._check(self.lib.t2l_adam_step(self._h, float(lr), float(beta1), float(beta2), float(eps), _stream_ptr(self.device)))

    def adam_state(self):
        """(exp_avg f32[n], exp_avg_sq f32[n], step) of the engine-stepped tensors, flat in bind order (GPU tensors)."""
        n, step = C.c_int64(0), C.c_int64(0)
        self._check(self.lib.t2l_adam_state(self._h, 0, None, None, C.byref(step), C.byref(n), _stream_ptr(self.device)))
        dev = torch.device("cuda", self.device)
        m = torch.empty((int(n.value),), dtype=torch.float32, device=dev)
        v = torch.empty_like(m)
        self._check(self.lib.t2l_adam_state(self._h, 0, m.data_ptr(), v.data_ptr(), C.byref(step), C.byref(n), _stream_ptr(self.device)))
        return m, v, int(step.value)

    def set_adam_state(self, m: torch.Tensor, v: torch.Tensor, step: int):
        n, st = C.c_int64(0), C.c_int64(int(step))
        self._check(self.lib.t2l_adam_state(self._h, 0, None, None, C.byref(C.c_int64(0)), C.byref(n), _stream_ptr(self.device)))
        if int(m.numel()) != int(n.value) or int(v.numel()) != int(n.value):
            raise T2LError(f"adam state of {int(m.numel())} elements does not match the bound tensors ({int(n.value)})")
        dev = torch.device("cuda", self.device)
        m = m.to(dev, torch.float32).contiguous()
        v = v.to(dev, torch.float32).contiguous()
        self._check(self.lib.t2l_adam_state(self._h, 1, m.data_ptr(), v.data_ptr(), C.byref(st), C.byref(n), _stream_ptr(self.device)))
        torch.cuda.current_stream(self.device).synchronize()  # m, v may be temporaries (the ENGINE's device: not the caller's current one)

    # ---- the text head's optimizer (t2l_text_adam_step: the tensors of text_train_bind that carry a gradient buffer, bind order)
    def text_zero_grad(self):
        self._check(self.lib.t2l_text_zero_grad(self._h, _stream_ptr(self.device)))

    def text_adam_step(self, lr: float, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8):
        self._check(self.lib.t2l_text_adam_step(self._h, float(lr), float(beta1), float(beta2), float(eps), _stream_ptr(self.device)))

    def text_adam_state(self):
        n, step = C.c_int64(0), C.c_int64(0)
        self._check(self.lib.t2l_text_adam_state(self._h, 0, None, None, C.byref(step), C.byref(n), _stream_ptr(self.device)))
        dev = torch.device("cuda", self.device)
        m = torch.empty((int(n.value),), dtype=torch.float32, device=dev)
        v = torch.empty_like(m)
        self._check(self.lib.t2l_text_adam_state(self._h, 0, m.data_ptr(), v.data_ptr(), C.byref(step), C.byref(n), _stream_ptr(self.device)))
        return m, v, int(step.value)

    def set_text_adam_state(self, m: torch.Tensor, v: torch.Tensor, step: int):
        n, st = C.c_int64(0), C.c_int64(int(step))
        self._check(self.lib.t2l_text_adam_state(self._h, 0, None, None, C.byref(C.c_int64(0)), C.byref(n), _stream_ptr(self.device)))
        if int(m.numel()) != int(n.value) or int(v.numel()) != int(n.value):
            raise T2LError(f"text-head adam state of {int(m.numel())} elements does not match the bound tensors ({int(n.value)})")
        dev = torch.device("cuda", self.device)
        m = m.to(dev, torch.float32).contiguous()
        v = v.to(dev, torch.float32).contiguous()
        self._check(self.lib.t2l_text_adam_state(self._h, 1, m.data_ptr(), v.data_ptr(), C.byref(st), C.byref(n), _stream_ptr(self.device)))
        torch.cuda.current_stream(self.device).synchronize()

    # ------------------------------------------------------------------ database + search
    @staticmethod
    def _pad_width(x: torch.Tensor, what: str) -> torch.Tensor:
        """The search kernels are compiled for rows of EMBED_DIM = 256 floats. Narrower embeddings (``--coarse_embed_dim 128``: a model
        whose encoders run elsewhere) are zero-padded: zeros add nothing to a dot product, so ids and float64 scores are exactly those of
        the narrow rows (the certificate's norms and the f16 scaling see the same largest element). Wider ones have no kernel."""
        if x.dim() != 2 or x.shape[1] > EMBED_DIM or x.shape[1] < 1:
            raise T2LError(f"{what}: expected [N, D <= {EMBED_DIM}], got {tuple(x.shape)}")
        if x.shape[1] < EMBED_DIM:
            x = torch.nn.functional.pad(x.to(torch.float32), (0, EMBED_DIM - int(x.shape[1])))
        return x

    def db_set(self, emb: torch.Tensor, row_offset: int = 0, owner=None):
        """Upload this rank's database shard. ``owner`` (any object) is remembered as ``db_owner`` so that callers sharing
        the engine can tell WHOSE rows are resident (row counts alone do not: db.CellDatabase.search); every call without
        one installs a fresh anonymous token, i.e. invalidates every earlier owner."""
        self.db_owner = owner if owner is not None else object()
        self.db_generation = getattr(self, "db_generation", 0) + 1
        n = int(emb.shape[0])
        emb = self._pad_width(emb, "db_set")
        ptr = self._ptr(emb, torch.float32, "db") if n > 0 else None
        self._check(self.lib.t2l_db_set(self._h, ptr, n, int(row_offset), _stream_ptr(self.device)))

    @property
    def db_rows(self) -> int:
        return int(self.lib.t2l_db_rows(self._h))

    def search(self, queries: torch.Tensor, k: int, out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
               join: bool = True):
        """Returns (idx i32[Q,k] global row ids best-first, scores f64[Q,k]); asynchronous on the current stream.
        ``join=True`` (default): stream-ordered on the caller's stream (t2l_search_ordered — the lanes are bypassed, no
        cross-stream hop). With ``set_option("search_lanes", n)`` pass ``join=False`` for a stream of independent batches:
        consecutive calls pipeline on internal streams (t2l.h); call ``search_join()`` before touching any of their results
        or re-using their ``queries`` / ``out`` buffers (after 64 un-joined calls the engine joins by itself)."""
        queries = self._pad_width(queries, "search")
        Q = int(queries.shape[0])
        if out is None:
            idx = torch.empty((Q, k), dtype=torch.int32, device=queries.device)
            sc = torch.empty((Q, k), dtype=torch.float64, device=queries.device)
        else:
            idx, sc = out
        qp = self._ptr(queries, torch.float32, "queries") if Q > 0 else None
        fn = self.lib.t2l_search_ordered if join else self.lib.t2l_search
        self._check(fn(self._h, qp, Q, int(k), self._ptr(idx, torch.int32, "out_idx"), self._ptr(sc, torch.float64, "out_score"),
                       _stream_ptr(self.device)))
        if join:
            self._lane_keepalive.clear()  # t2l_search_ordered joined whatever was pending
        else:
            self._lane_keepalive.append((queries, idx, sc))  # the lane streams are invisible to torch's allocator
            if len(self._lane_keepalive) >= 64:
                self.search_join()
        return idx, sc

    def search_join(self):
        """Order every pipelined search issued so far into the current stream."""
        self._check(self.lib.t2l_search_join(self._h, _stream_ptr(self.device)))
        self._lane_keepalive.clear()

    def merge_topk(self, idx: torch.Tensor, score: torch.Tensor):
        """idx i32[P,Q,K], score f64[P,Q,K] (all-gathered per-shard results) -> (i32[Q,K], f64[Q,K]) by (score desc, id asc). No
        order is assumed inside a part, -1 entries may sit anywhere; ids are meant to be unique across parts (a duplicated
        (score, id) pair comes out twice)."""
        P, Q, K = (int(x) for x in idx.shape)
        out_i = torch.empty((Q, K), dtype=torch.int32, device=idx.device)
        out_s = torch.empty((Q, K), dtype=torch.float64, device=idx.device)
        self._check(self.lib.t2l_merge_topk(self._h, self._ptr(idx, torch.int32, "idx"),
                                            self._ptr(score, torch.float64, "score"), P, Q, K, out_i.data_ptr(),
                                            out_s.data_ptr(), _stream_ptr(self.device)))
        return out_i, out_s

    def pack_pairs(self, idx: torch.Tensor, score: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """(idx i32[Q,K], score f64[Q,K]) -> f64[Q,K,2] records {score, row id}: one buffer for one collective."""
        Q, K = (int(x) for x in idx.shape)
        pairs = out if out is not None else torch.empty((Q, K, 2), dtype=torch.float64, device=idx.device)
        self._check(self.lib.t2l_pack_pairs(self._h, self._ptr(idx, torch.int32, "idx"),
                                            self._ptr(score, torch.float64, "score"), Q, K, pairs.data_ptr(), _stream_ptr(self.device)))
        return pairs

    def merge_pairs(self, pairs: torch.Tensor):
        """pairs f64[P,Q,K,2] (all-gathered) -> (idx i32[Q,K], score f64[Q,K])."""
        P, Q, K, _ = (int(x) for x in pairs.shape)
        out_i = torch.empty((Q, K), dtype=torch.int32, device=pairs.device)
        out_s = torch.empty((Q, K), dtype=torch.float64, device=pairs.device)
        self._check(self.lib.t2l_merge_pairs(self._h, self._ptr(pairs, torch.float64, "pairs"), P, Q, K, out_i.data_ptr(),
                                             out_s.data_ptr(), _stream_ptr(self.device)))
        return out_i, out_s

    @staticmethod
    def result_block(Q: int, k: int, device, parts: int = 1):
        """One exchange block per part: ``(buf u8[parts, block_bytes], idx i32[Q,k] view, score f64[Q,k] view, block_bytes,
        score_offset)`` — the views alias part 0's block, so ``search(..., out=(idx, score))`` fills it in place."""
        score_offset = (Q * k * 4 + 7) // 8 * 8
        block_bytes = score_offset + Q * k * 8
        buf = torch.empty((parts, block_bytes), dtype=torch.uint8, device=device)
        idx = buf[0, :Q * k * 4].view(torch.int32).view(Q, k)
        sc = buf[0, score_offset:score_offset + Q * k * 8].view(torch.float64).view(Q, k)
        return buf, idx, sc, block_bytes, score_offset

    def merge_gathered(self, blocks: torch.Tensor, block_bytes: int, score_offset: int, parts: int, Q: int, k: int, out=None):
        """blocks u8[parts * block_bytes] (all-gathered ``result_block``s) -> (idx i32[Q,k], score f64[Q,k])."""
        if out is None:
            out_i = torch.empty((Q, k), dtype=torch.int32, device=blocks.device)
            out_s = torch.empty((Q, k), dtype=torch.float64, device=blocks.device)
        else:
            out_i, out_s = out
        self._check(self.lib.t2l_merge_gathered(self._h, self._ptr(blocks, torch.uint8, "blocks"), int(block_bytes), int(score_offset),
                                                int(parts), int(Q), int(k), out_i.data_ptr(), out_s.data_ptr(), _stream_ptr(self.device)))
        return out_i, out_s

    def search_many(self, queries: torch.Tensor, k: int, out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
        """queries f32[n_batches, Q, 256]: n_batches independent searches of Q queries each (t2l_search_many: the loop runs in C).
        Returns (idx i32[n_batches, Q, k], scores f64[n_batches, Q, k])."""
        if queries.dim() != 3 or queries.shape[2] != EMBED_DIM:
            raise T2LError(f"search_many: expected [n_batches, Q, {EMBED_DIM}], got {tuple(queries.shape)}")
        nb, Q = int(queries.shape[0]), int(queries.shape[1])
        if out is None:
            out = (torch.empty((nb, Q, k), dtype=torch.int32, device=queries.device),
                   torch.empty((nb, Q, k), dtype=torch.float64, device=queries.device))
        self._check(self.lib.t2l_search_many(self._h, self._ptr(queries, torch.float32, "queries"), nb, Q, int(k), out[0].data_ptr(),
                                             out[1].data_ptr(), _stream_ptr(self.device)))
        return out

    def search_fallbacks(self) -> int:
        c = C.c_int32(0)
        self._check(self.lib.t2l_search_fallbacks(self._h, C.byref(c)))
        return int(c.value)

    def search_counters(self) -> dict:
        """Counters of the last search (include/t2l.h: t2l_search_counters)."""
        c = (C.c_int32 * 8)()
        self._check(self.lib.t2l_search_counters(self._h, c))
        names = ("valu_exact_scans", "rescored", "to_fallback_kernel", "probe", "deferred_to_mfma_exact", "wide_repairs",
                 "mfma_exact_uncertified", "mfma_exact_served")
        return {n: int(c[i]) for i, n in enumerate(names)}

    def search_rescored(self) -> int:
        """Queries of the last search whose first certificate failed (all kept candidates re-scored in float64)."""
        c = C.c_int32(0)
        self._check(self.lib.t2l_search_rescored(self._h, C.byref(c)))
        return int(c.value)

    # ------------------------------------------------------------------ loss
    def contrastive_loss(self, anchor: torch.Tensor, positive: torch.Tensor, temperature: float, need_grad=True):
        B = int(anchor.shape[0])
        loss = torch.empty((1,), dtype=torch.float32, device=anchor.device)
        ga = torch.empty_like(anchor) if need_grad else None
        gp = torch.empty_like(positive) if need_grad else None
        self._check(self.lib.t2l_contrastive_loss(
            self._h, self._ptr(anchor, torch.float32, "anchor"), self._ptr(positive, torch.float32, "positive"), B,
            float(temperature), loss.data_ptr(), self._ptr(ga, torch.float32, "ga"), self._ptr(gp, torch.float32, "gp"),
            _stream_ptr(self.device)))
        return loss, ga, gp

    # ------------------------------------------------------------------ knobs
    def set_option(self, name: str, value: float):
        self._check(self.lib.t2l_set_option(self._h, name.encode(), float(value)))

    def kernel_stats(self, name: str):
        """(average ms, launches) of kernel ``name`` since the last call (needs option profile_events=1)."""
        ms, n = C.c_float(0), C.c_int32(0)
        self._check(self.lib.t2l_kernel_stats(self._h, name.encode(), C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)
