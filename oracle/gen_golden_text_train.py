"""TEST INFRASTRUCTURE — build-container only.

Golden vectors for the TEXT half of the training step (training/coarse.py:40-58): imports the upstream
reference, feeds its ``LanguageEncoder`` fixed T5 hidden states (a stub ``llm_model``: T5 is frozen by --fixed_embedding and its
weights do not exist in this image), runs ``model.train()`` -> ``encode_text`` -> ``ContrastiveLoss(0.1)`` against a fixed cell batch
-> ``backward()`` and writes tests/golden/train_step_text.npz (DATA only): the head's output, the loss, the gradient of every head
parameter (sampled beyond 1,024 elements as in gen_golden_train.py) and inter_mlp's BatchNorm running buffers after the step.

The dropout sites of both nn.TransformerEncoderLayers are set to p = 0 (torch's mask stream cannot be replayed; the build's
counter-based masks are pinned separately against the float64 oracle).
"""
from __future__ import annotations

import os.path as osp
import sys
import tempfile

import numpy as np

HERE = osp.dirname(osp.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as H  # noqa: E402

H.setup_reference_imports()

import torch  # noqa: E402

from gen_golden import to_torch_sd  # noqa: E402
from gen_golden_train import pack_tensor  # noqa: E402
from text2loc_amd import synth  # noqa: E402

OUT = osp.join(H.REPO, "tests", "golden")
torch.set_num_threads(4)


def main():
    from datapreparation.kitti360pose.utils import COLOR_NAMES, KNOWN_CLASS
    from models.cell_retrieval import CellRetrievalNetwork
    from training.losses import ContrastiveLoss

    W_SEED, H_SEED, B, S, L = 5, 12, 8, 6, 11
    tmp = tempfile.mkdtemp()
    hf_dir = H.make_tiny_t5(osp.join(tmp, "t5"))
    pn_path = osp.join(tmp, "pn.pth")
    H.make_pointnet_ckpt(pn_path)
    args = H.make_args(hf_dir, pn_path, class_embed=True, color_embed=True)
    model = CellRetrievalNetwork(KNOWN_CLASS, COLOR_NAMES, args)
    sd_np = synth.make_language_head_weights(W_SEED)
    model.load_state_dict(to_torch_sd(sd_np), strict=False)
    hidden = synth.make_t5_hidden(B * S, L, seed=H_SEED)

    class StubT5(torch.nn.Module):
        def forward(self, input_ids=None, attention_mask=None, output_attentions=False):
            from easydict import EasyDict
            assert input_ids.shape[0] == B * S
            return EasyDict(last_hidden_state=torch.from_numpy(hidden))

    le = model.language_encoder
    le.llm_model = StubT5()
    for layer in list(le.intra_module) + list(le.inter_module):
        layer.dropout.p = layer.dropout1.p = layer.dropout2.p = 0.0
        layer.self_attn.dropout = 0.0
    model.train()
    rng = np.random.default_rng([9, 0x7E])
    cells_np = synth.unit_rows(rng.standard_normal((B, 256))).astype(np.float32)
    texts = [" ".join(["The pose is north of a gray pole."] * S)] * B
    names = [n for n, _ in model.named_parameters() if n.startswith("language_encoder.") and ".llm_model." not in n]
    params = dict(model.named_parameters())
    for n in names:
        params[n].grad = None
    head_out = {}
    h = le.register_forward_hook(lambda mod, i, o: head_out.__setitem__("out", o.detach().clone()))
    anchor = model.encode_text(texts)                                   # training/coarse.py:44
    h.remove()
    loss = ContrastiveLoss(temperature=0.1)(anchor, torch.from_numpy(cells_np))   # :52
    loss.backward()                                                     # :55
    out = {"weight_seed": W_SEED, "hidden_seed": H_SEED, "batch": B, "n_hints": S, "n_tokens": L, "temperature": 0.1,
           "cells": cells_np, "head_out": head_out["out"].numpy(), "anchor": anchor.detach().numpy(), "loss": np.float32(loss.item())}
    used = []
    for n in names:
        g = params[n].grad
        if g is None:
            continue
        used.append(n)
        pack_tensor(out, "grad", n, g.numpy())
    for n, b in model.named_buffers():
        if n.startswith("language_encoder.inter_mlp") and "running" in n:
            out["buf/" + n] = b.numpy().copy()
    out["used_params"] = np.array(used)
    np.savez_compressed(osp.join(OUT, "train_step_text.npz"), **out)
    print("train_step_text loss", float(loss), "params with grad", len(used), [n for n in names if n not in used])


if __name__ == "__main__":
    main()
