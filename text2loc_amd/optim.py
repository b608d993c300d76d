"""``optim.Adam(model.parameters(), lr)`` of the reference's training script (training/coarse.py:258) for a
``text2loc_amd.CellRetrievalNetwork``: the object-branch parameters (with the PointNet++ backbone when it trains) are stepped by the engine's multi-tensor Adam kernel
(t2l_adam_step: one launch over every bound tensor, moments kept in HBM by the library), everything else (the language
head) by ``torch.optim.Adam`` with the same hyper-parameters. It is a ``torch.optim.Optimizer``, so the reference's
``ExponentialLR`` / ``StepLR`` schedulers (training/coarse.py:268-273) drive it unchanged through ``param_groups``.

Data-parallel training (one process per GPU; not in the reference — SURVEY.md 8e): ``Adam(model, lr, group=pg)`` all-reduces
the gradients at the top of ``step()``. The engine-stepped parameters' ``.grad`` are views of ONE flat buffer
(``CellRetrievalNetwork.train_flat_grad``), so they travel as a single RCCL all_reduce (16.5 M floats = 66 MB with the
backbone: per-link bound on xGMI, one large collective instead of ~150 small ones); the language head's gradients are
flattened into a second one. ``grad_reduce="sum"`` pairs with ``ContrastiveLoss(group=pg)`` (every rank differentiates the
GLOBAL loss w.r.t. its own rows: the sum over ranks is the exact gradient); ``"mean"`` is the usual average for per-rank losses.
"""
from __future__ import annotations

import torch


def all_reduce_flat(buf: torch.Tensor, group=None, mean: bool = False) -> torch.Tensor:
    """In-place all_reduce (sum, or mean) of one flat gradient buffer: RCCL on the tensor's own device under "nccl"; under gloo a
    device tensor is staged through the host (several processes on one GPU in the tests — RCCL refuses duplicate devices)."""
    import torch.distributed as dist

    if buf.is_cuda and dist.get_backend(group) != "nccl":
        h = buf.cpu()
        dist.all_reduce(h, group=group)
        buf.copy_(h)
    else:
        dist.all_reduce(buf, group=group)
    if mean:
        buf.div_(dist.get_world_size(group))
    return buf


class Adam(torch.optim.Optimizer):
    def __init__(self, model, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, group=None, grad_reduce: str = "sum",
                 data_parallel: bool = False, sync_bn: bool = False, text_engine: bool = True):
        """``group``: a torch.distributed process group, or ``data_parallel=True`` for the default one — turns the gradient
        all-reduce of ``step()`` on. ``sync_bn``: also ``model.sync_batchnorm(group)`` — BatchNorm statistics over all ranks' batches,
        i.e. the reference's one-process step on the global batch. ``text_engine``: the language head's trained parameters (when the
        engine's training head serves this model: ``LanguageEncoder.engine_optimizer_params``) are stepped by ``t2l_text_adam_step`` — one
        launch, moments inside the library — instead of a torch.optim.Adam beside the engine."""
        if grad_reduce not in ("sum", "mean"):
            raise ValueError("grad_reduce must be 'sum' or 'mean'")
        self._group, self._grad_reduce, self._dp = group, grad_reduce, bool(data_parallel) or group is not None
        obj, rest, text = [], [], []
        le = getattr(model, "language_encoder", None)
        on_gpu = any(p.is_cuda for p in model.parameters())
        text_ids = {id(p) for _, p in le.engine_optimizer_params()} if (text_engine and on_gpu and hasattr(le, "engine_optimizer_params")) else set()
        for n, p in model.named_parameters():
            if not p.requires_grad:
                continue
            if id(p) in text_ids:
                text.append(p)  # the text head's trained parameters: t2l_text_adam_step (one launch, moments inside the library)
                continue
            if n.startswith(("object_encoder.pointnet.class_classifier.", "object_encoder.pointnet.color_classifier.")):
                rest.append(p)  # not on the path (features2 is consumed): .grad stays None, torch skips them as in the reference
            elif n.startswith(("object_encoder.", "obj_inter_module.")):
                obj.append(p)   # incl. the PointNet++ backbone unless --pointnet_freeze
            else:
                rest.append(p)
        # names in named_parameters order of everything the object-branch kernel does NOT step, and which optimizer steps each here:
        # a checkpoint written under the other ``text_engine`` setting (or before the head moved into the engine) holds the same
        # moments under the other layout — load_state_dict migrates by these names
        text_set = {id(p) for p in text}
        obj_set = {id(p) for p in obj}
        self._non_obj = [(n, p, "text" if id(p) in text_set else "rest") for n, p in model.named_parameters()
                         if p.requires_grad and id(p) not in obj_set]
        groups = [{"params": obj, "t2l_engine": True}]
        if rest:
            groups.append({"params": rest, "t2l_engine": False})
        if text:
            groups.append({"params": text, "t2l_engine": "text_head"})
        super().__init__(groups, dict(lr=lr, betas=tuple(betas), eps=eps))
        self._model = model
        self._text = bool(text)
        if sync_bn:
            model.sync_batchnorm(group)
        self._torch = torch.optim.Adam(rest, lr=lr, betas=tuple(betas), eps=eps) if rest else None

    def zero_grad(self, set_to_none: bool = False):
        """Object-branch gradients are zeroed IN PLACE by one kernel (the engine accumulates into fixed buffers)."""
        if self._model.device.type == "cuda":
            self._model.train_engine().zero_grad()
            if self._text:
                self._model.language_encoder.engine_zero_grad()  # in place: the engine accumulates into the bound buffers
        if self._torch is not None:
            self._torch.zero_grad(set_to_none=set_to_none)

    # ---- data parallel ------------------------------------------------------------------------------------------------
    def _world(self) -> int:
        import torch.distributed as dist

        if not self._dp:
            return 1
        return dist.get_world_size(self._group) if dist.is_available() and dist.is_initialized() else 1

    @torch.no_grad()
    def all_reduce_grads(self):
        """ONE all_reduce over the engine-owned flat gradient buffer + one over the flattened torch-side gradients."""
        world = self._world()
        if world == 1:
            return
        self._model.train_engine()  # binds (and so allocates) the flat buffer if nothing has yet
        bufs = []
        flat = self._model.train_flat_grad()
        if flat is not None:
            bufs.append((flat, None))
        # a rank-INDEPENDENT list: every torch-side parameter, a missing gradient (a branch this rank's batch skipped, set_to_none) as
        # zeros — ranks that disagreed on which gradients exist would issue all_reduces of different sizes (a hang, or silent garbage)
        rest = [p for g in self.param_groups[1:] for p in g["params"]]
        if rest:
            bufs.append((torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in rest]), rest))
        for buf, params in bufs:
            all_reduce_flat(buf, self._group, mean=self._grad_reduce == "mean")
            if params is not None:
                off = 0
                for p in params:
                    g = buf[off:off + p.numel()].view_as(p)
                    if p.grad is None:
                        p.grad = g.clone()  # (another rank may have contributed: every rank steps the same set)
                    else:
                        p.grad.copy_(g)
                    off += p.numel()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self.all_reduce_grads()
        g = self.param_groups[0]
        eng = self._model.train_engine()
        eng.adam_step(g["lr"], g["betas"][0], g["betas"][1], g["eps"])
        self._model._train_generation += 1  # parameters changed behind torch's back: eval weights must be re-folded
        if self._torch is not None:
            for src, dst in zip([g_ for g_ in self.param_groups[1:] if g_["t2l_engine"] is False], self._torch.param_groups):
                dst["lr"], dst["betas"], dst["eps"] = src["lr"], src["betas"], src["eps"]
            self._torch.step()
        if self._text:
            gt = self.param_groups[-1]
            self._model.language_encoder.engine_adam_step(gt["lr"], gt["betas"], gt["eps"])
        return loss

    # ---- checkpoint / resume: the engine-side moments are part of the optimizer state -------------------------------
    def state_dict(self):
        sd = {"param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups],
              "torch": self._torch.state_dict() if self._torch is not None else None, "t2l_engine": None, "t2l_text_engine": None}
        if self._model.device.type == "cuda":
            m, v, step = self._model.train_engine().adam_state()
            # the engine keeps one bias-correction step for the object branch and one for the PointNet++ backbone (which only steps
            # when its backward ran): two explicit fields (round 3 leaked the packed word step | step_pn << 32 into "step")
            sd["t2l_engine"] = {"exp_avg": m.cpu(), "exp_avg_sq": v.cpu(), "step": int(step) & 0xFFFFFFFF, "step_pn": int(step) >> 32}
            if self._text:
                le = self._model.language_encoder
                le._bind_text_train(self._model.device)
                m, v, step = le._th_train_engine.text_adam_state()
                sd["t2l_text_engine"] = {"exp_avg": m.cpu(), "exp_avg_sq": v.cpu(), "step": int(step)}
        return sd

    def _saved_moments(self, sd):
        """{parameter name: (exp_avg, exp_avg_sq, step)} of every non-object-branch parameter a checkpoint holds moments for, whichever
        layout wrote it: (i) head inside ``sd["torch"]`` (``text_engine=False``, CPU models, checkpoints older than the engine-side head);
        (ii) head as the engine's flat moments ``sd["t2l_text_engine"]`` (bind order = named_parameters order) + the rest in ``sd["torch"]``.
        A checkpoint that fits neither raises, naming the sizes — never a silent restart of Adam."""
        names_all = [n for n, _, _ in self._non_obj]
        names_rest = [n for n, _, k in self._non_obj if k == "rest"]
        text = [(n, p) for n, p, k in self._non_obj if k == "text"]
        ts, te = sd.get("torch"), sd.get("t2l_text_engine")
        n_saved = sum(len(g["params"]) for g in ts["param_groups"]) if ts is not None else 0
        out = {}
        if te is None:
            if n_saved == len(names_all):
                order = names_all                      # layout (i): the head's moments sit in the torch optimizer's state
            elif n_saved == len(names_rest) and not text:
                order = names_rest
            else:
                raise ValueError(f"optimizer checkpoint holds {n_saved} torch-side parameters; this model has {len(names_all)} outside the "
                                 f"object branch ({len(text)} of them in the text head): not a checkpoint of this model")
        else:
            head = [(n, p) for n, p in (self._model.language_encoder.engine_optimizer_params()
                                        if hasattr(getattr(self._model, "language_encoder", None), "engine_optimizer_params") else [])]
            head = [("language_encoder." + n, p) for n, p in head]
            total = sum(p.numel() for _, p in head)
            if int(te["exp_avg"].numel()) != total:
                raise ValueError(f"optimizer checkpoint holds {int(te['exp_avg'].numel())} text-head moments; this model's engine head has {total}")
            head_names = {n for n, _ in head}
            order = [n for n in names_all if n not in head_names]
            if n_saved != len(order):
                raise ValueError(f"optimizer checkpoint holds {n_saved} torch-side parameters beside the engine's text head; this model has {len(order)}")
            off = 0
            for n, p_ in head:
                k = p_.numel()
                out[n] = (te["exp_avg"][off:off + k].view_as(p_), te["exp_avg_sq"][off:off + k].view_as(p_), int(te["step"]))
                off += k
        if ts is not None:
            saved_ids = [i for g in ts["param_groups"] for i in g["params"]]
            for i, n in zip(saved_ids, order):
                st = ts["state"].get(i)
                if st:
                    out[n] = (st["exp_avg"], st["exp_avg_sq"], int(st["step"]))
        return out

    def load_state_dict(self, sd):
        # hyper-parameters go back by the group's MARKER, not its position: the two head layouts have different group lists
        # ([obj, rest] against [obj, rest, text] or [obj, text]). A group with no counterpart in the checkpoint is the text head
        # under the other layout: it takes lr / betas / eps of the group that held the head there (text <-> rest); the marker
        # itself is never taken from the checkpoint.
        by_kind = {}
        for sg in sd["param_groups"]:
            by_kind.setdefault(sg.get("t2l_engine", None), sg)
        for g in self.param_groups:
            kind = g["t2l_engine"]
            src = by_kind.get(kind)
            if src is None and kind == "text_head":
                src = by_kind.get(False)
            if src is None and kind is False:
                src = by_kind.get("text_head")
            if src is None:
                continue
            g.update({k: v for k, v in src.items() if k not in ("params", "t2l_engine")})
        saved = self._saved_moments(sd)
        if self._torch is not None:
            rest = [(n, p) for n, p, k in self._non_obj if k == "rest"]
            ref_group = (sd.get("torch") or {}).get("param_groups", [None])[0] or self._torch.state_dict()["param_groups"][0]
            state = {}
            for i, (n, p) in enumerate(rest):
                if n in saved:
                    m, v, step = saved[n]
                    state[i] = {"step": torch.tensor(float(step)), "exp_avg": m.detach().clone().view_as(p), "exp_avg_sq": v.detach().clone().view_as(p)}
            self._torch.load_state_dict({"state": state, "param_groups": [dict(ref_group, params=list(range(len(rest))))]})
        e = sd.get("t2l_engine")
        if e is not None:
            step = int(e["step"])
            step_pn = int(e["step_pn"]) if "step_pn" in e else (step >> 32 if step >> 32 else step & 0xFFFFFFFF)  # older checkpoints:
            self._model.train_engine().set_adam_state(e["exp_avg"], e["exp_avg_sq"], (step & 0xFFFFFFFF) | (step_pn << 32))  # packed, or one step for both
        if self._text:  # the head's moments go into the engine whichever layout the checkpoint used (flat, bind order)
            le = self._model.language_encoder
            le._bind_text_train(self._model.device)
            text = [(n, p) for n, p, k in self._non_obj if k == "text"]
            have = [n for n, _ in text if n in saved]
            if have and len(have) != len(text):
                raise ValueError(f"optimizer checkpoint holds moments for {len(have)} of the {len(text)} text-head tensors")
            if have:
                steps = {saved[n][2] for n, _ in text}
                if len(steps) != 1:
                    raise ValueError(f"text-head tensors of the checkpoint disagree on the Adam step ({sorted(steps)}): the engine keeps one")
                dev = self._model.device
                m = torch.cat([saved[n][0].detach().to(dev, torch.float32).reshape(-1) for n, _ in text])
                v = torch.cat([saved[n][1].detach().to(dev, torch.float32).reshape(-1) for n, _ in text])
                le._th_train_engine.set_text_adam_state(m, v, steps.pop())
