"""``optim.Adam(model.parameters(), lr)`` of the reference's training script (training/coarse.py:258) for a
``text2loc_amd.CellRetrievalNetwork``: the object-branch parameters (with the PointNet++ backbone when it trains) are stepped by the engine's multi-tensor Adam kernel
(t2l_adam_step: one launch over every bound tensor, moments kept in HBM by the library), everything else (the language
head) by ``torch.optim.Adam`` with the same hyper-parameters. It is a ``torch.optim.Optimizer``, so the reference's
``ExponentialLR`` / ``StepLR`` schedulers (training/coarse.py:268-273) drive it unchanged through ``param_groups``.
"""
from __future__ import annotations

import torch


class Adam(torch.optim.Optimizer):
    def __init__(self, model, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        obj, rest = [], []
        for n, p in model.named_parameters():
            if not p.requires_grad:
                continue
            if n.startswith(("object_encoder.pointnet.class_classifier.", "object_encoder.pointnet.color_classifier.")):
                rest.append(p)  # not on the path (features2 is consumed): .grad stays None, torch skips them as in the reference
            elif n.startswith(("object_encoder.", "obj_inter_module.")):
                obj.append(p)   # incl. the PointNet++ backbone unless --pointnet_freeze
            else:
                rest.append(p)
        groups = [{"params": obj, "t2l_engine": True}]
        if rest:
            groups.append({"params": rest, "t2l_engine": False})
        super().__init__(groups, dict(lr=lr, betas=tuple(betas), eps=eps))
        self._model = model
        self._torch = torch.optim.Adam(rest, lr=lr, betas=tuple(betas), eps=eps) if rest else None

    def zero_grad(self, set_to_none: bool = False):
        """Object-branch gradients are zeroed IN PLACE by one kernel (the engine accumulates into fixed buffers)."""
        if self._model.device.type == "cuda":
            self._model.train_engine().zero_grad()
        if self._torch is not None:
            self._torch.zero_grad(set_to_none=set_to_none)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        g = self.param_groups[0]
        eng = self._model.train_engine()
        eng.adam_step(g["lr"], g["betas"][0], g["betas"][1], g["eps"])
        self._model._train_generation += 1  # parameters changed behind torch's back: eval weights must be re-folded
        if self._torch is not None:
            for src, dst in zip(self.param_groups[1:], self._torch.param_groups):
                dst["lr"], dst["betas"], dst["eps"] = src["lr"], src["betas"], src["eps"]
            self._torch.step()
        return loss

    # ---- checkpoint / resume: the engine-side moments are part of the optimizer state -------------------------------
    def state_dict(self):
        sd = {"param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups],
              "torch": self._torch.state_dict() if self._torch is not None else None, "t2l_engine": None}
        if self._model.device.type == "cuda":
            m, v, step = self._model.train_engine().adam_state()
            sd["t2l_engine"] = {"exp_avg": m.cpu(), "exp_avg_sq": v.cpu(), "step": step}
        return sd

    def load_state_dict(self, sd):
        for g, saved in zip(self.param_groups, sd["param_groups"]):
            g.update({k: v for k, v in saved.items() if k != "params"})
        if self._torch is not None and sd.get("torch") is not None:
            self._torch.load_state_dict(sd["torch"])
        e = sd.get("t2l_engine")
        if e is not None:
            self._model.train_engine().set_adam_state(e["exp_avg"], e["exp_avg_sq"], int(e["step"]))
