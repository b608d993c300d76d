"""CPU tier: the training-step oracle (oracle/t2l_oracle_train.py) against the goldens of the imported reference's
``model.train()`` -> ``encode_objects`` -> ``ContrastiveLoss`` -> ``backward`` -> ``Adam.step`` run,
and its analytic backward against central differences of its own forward (with the dropout masks on)."""
import numpy as np
import pytest

from oracle import t2l_oracle as O
from oracle import t2l_oracle_train as OT
from text2loc_amd import synth


def sample_index(name: str, numel: int, n_sample: int = 512) -> np.ndarray:
    seed = int.from_bytes(name.encode()[-8:].rjust(8, b"\0"), "little") % (2 ** 32)
    return np.sort(np.random.default_rng([seed, numel]).choice(numel, size=min(n_sample, numel), replace=False))


def golden_view(g, tag, name, full):
    """(expected, got) of one tensor: whole when small, else at the fixture's sample positions."""
    flat = np.asarray(full, dtype=np.float64).ravel()
    exp = g[f"{tag}/{name}"]
    return exp, (flat if flat.size <= 1024 else flat[sample_index(name, flat.size)])


def load_case(g, mode):
    shape = {k: int(g[k]) for k in ("min_obj", "max_obj") if k in g.files}  # the margin fixtures use small cells
    cells = synth.make_cells(int(g["n_cells"]), seed=int(g["cell_seed"]), with_pn_feat=True, **shape)
    for k in ("class_idx", "color_idx", "rgb", "center", "n_pts", "offsets", "counts"):
        cells[k] = g["in_" + k]  # exactly what the reference derived from its Object3d instances
    sd = synth.make_object_branch_weights(int(g["weight_seed"]))
    return cells, sd, mode == "embed"


@pytest.mark.parametrize("mode", ["embed", "pn"])
def test_train_step_oracle_matches_the_reference(golden, mode):
    g = golden(f"train_step_{mode}")
    cells, sd, embed = load_case(g, mode)
    out0, _ = OT.encode_cells_train(cells, sd, embed, embed)
    assert np.abs(out0 - g["positive"]).max() < 2e-6
    loss, d_anchor, d_pos = O.contrastive_loss(g["anchor"], out0, float(g["temperature"]), dtype=np.float64)
    assert abs(loss - float(g["loss"])) < 2e-6
    assert np.abs(d_anchor - g["grad_anchor"]).max() < 1e-6
    out, info = OT.encode_cells_train(cells, sd, embed, embed, grad_out=d_pos)
    used = [str(n) for n in g["used_params"]]
    assert sorted(used) == sorted(info["grads"].keys())
    for n in used:
        exp, got = golden_view(g, "grad", n, info["grads"][n])
        rms = float(g[f"grad_norm/{n}"]) / np.sqrt(max(np.asarray(info["grads"][n]).size, 1))
        # The reference runs in float32: one ReLU pre-activation within rounding of 0 flips sign between float32 and
        # float64 (layer-1 hidden unit 192 in the embed case) and moves the upstream gradients by ~1e-3 of their rms;
        # biases in front of a BatchNorm have true gradient 0 and carry ~1e-6 of float32 noise in the reference.
        if n.startswith("object_encoder.") and n.endswith(".0.bias"):  # Linear bias in front of a BatchNorm
            assert np.abs(got).max() < 1e-9 and np.abs(exp).max() < 1e-4, n
            continue
        err = np.abs(got - exp)
        assert (err < 1e-2 * rms + 1e-5).mean() >= 0.95 and err.max() < 0.2 * rms + 1e-5, n
        assert abs(np.sqrt((np.asarray(info["grads"][n]) ** 2).sum()) - float(g[f"grad_norm/{n}"])) < 2e-3 * float(g[f"grad_norm/{n}"]) + 2e-4, n
    # the same restatement evaluated in float32 lands on the reference's own rounding where no ReLU sits in between
    _, info32 = OT.encode_cells_train(cells, sd, embed, embed, grad_out=d_pos, dtype=np.float32)
    for n in used:
        if n.startswith(("obj_inter_module.1.linear2", "obj_inter_module.1.norm2")):  # downstream of every ReLU
            exp, got = golden_view(g, "grad", n, info32["grads"][n])
            rms = float(g[f"grad_norm/{n}"]) / np.sqrt(max(np.asarray(info32["grads"][n]).size, 1))
            assert np.abs(got - exp).max() < 2e-4 * rms + 1e-7, n
    # BatchNorm running statistics after the step
    new = OT.bn_running_update(sd, info["bn_stats"])
    n_buf = 0
    for k in g.files:
        if k.startswith("buf/"):
            name = k[4:]
            if name.rsplit(".", 1)[0] + ".running_mean" not in new:
                assert np.allclose(g[k], sd[name]) or name.endswith("num_batches_tracked")  # untouched branch
                continue
            assert np.allclose(np.asarray(new[name], dtype=np.float64), g[k], rtol=2e-5, atol=2e-6), name
            n_buf += 1
    assert n_buf >= 9
    # one Adam step from zero state, fed with the reference's own gradients (Adam's first step is lr*sign(g): feeding
    # our gradients instead would compare the sign of float32 noise wherever the true gradient is 0)
    for n in used:
        flat = np.asarray(sd[n], dtype=np.float64).ravel()
        p0 = flat if flat.size <= 1024 else flat[sample_index(n, flat.size)]
        p1, _, _ = OT.adam_step(p0, g[f"grad/{n}"].astype(np.float64), 0.0, 0.0, 1, float(g["lr"]))
        assert np.abs(p1 - g[f"adam/{n}"]).max() < 2e-7, n


def margin_grad_check(g, n, got_full, scale=1.0):
    """Gradient tensor ``n`` against a margin fixture: 1e-4 of the tensor's rms, element-wise maximum. Two families are
    not informative and keep their own bounds: Linear biases in front of a BatchNorm (true gradient 0) and the Linear
    weights in front of a BatchNorm (y = BN(w x + b) is invariant to the scale of w, so the true gradient is the residual
    of terms that cancel to ~1e-4 of their size)."""
    exp, got = golden_view(g, "grad", n, got_full)
    rms = float(g[f"grad_norm/{n}"]) / np.sqrt(max(np.asarray(got_full).size, 1))
    if n.startswith("object_encoder.") and n.endswith(".0.bias"):
        assert np.abs(got).max() < 1e-3 and np.abs(exp).max() < 1e-3, n  # float32 cancellation noise around 0
        return
    tol = 2e-3 if n.startswith("object_encoder.") and n.endswith(".0.weight") else 1e-4
    err = np.abs(got - exp).max()
    assert err < scale * tol * rms + 1e-9, (n, err / rms)


@pytest.mark.parametrize("mode", ["embed", "pn"])
def test_margin_fixture_pins_the_gradients_at_rounding_level(golden, mode):
    """tests/golden/train_step_margin_*.npz: a B=3 batch picked by seed search (oracle/gen_golden_train.py) so that every
    ReLU input of the reference's run is >= 1e-4 away from 0 — no rounding order can flip a unit, and the float64 oracle
    meets the float32 reference's parameter gradients to 1e-4 of each tensor's rms (element-wise maximum, not a quantile)."""
    g = golden(f"train_step_margin_{mode}")
    assert float(g["relu_margin"]) > 5e-5
    cells, sd, embed = load_case(g, mode)
    out0, _ = OT.encode_cells_train(cells, sd, embed, embed)
    assert np.abs(out0 - g["positive"]).max() < 2e-6
    loss, d_anchor, d_pos = O.contrastive_loss(g["anchor"], out0, float(g["temperature"]), dtype=np.float64)
    assert abs(loss - float(g["loss"])) < 1e-5 and np.abs(d_anchor - g["grad_anchor"]).max() < 1e-5  # float32 reference, |grad| ~ 0.3
    for dtype in (np.float64, np.float32):
        _, info = OT.encode_cells_train(cells, sd, embed, embed, grad_out=d_pos, dtype=dtype)
        for n in [str(x) for x in g["used_params"]]:
            margin_grad_check(g, n, info["grads"][n])


def test_backward_matches_central_differences_with_dropout():
    B = 3
    cells = synth.make_cells(B, seed=11, with_pn_feat=True, min_obj=3, max_obj=31)
    sd = {k: np.asarray(v, dtype=np.float64) for k, v in synth.make_object_branch_weights(2).items()}
    rng = np.random.default_rng(5)
    gout = rng.standard_normal((B, 256))
    for embed in (True, False):
        kw = dict(p_drop=0.1, seed=1234)
        out, info = OT.encode_cells_train(cells, sd, embed, embed, grad_out=gout, **kw)

        def f(sd2):
            o, _ = OT.encode_cells_train(cells, sd2, embed, embed, **kw)
            return float((o * gout).sum())

        names = sorted(info["grads"].keys())
        checked = 0
        for n in names:
            w = sd[n]
            for _ in range(2):
                idx = tuple(rng.integers(0, s) for s in w.shape)
                if "embedding" in n and idx[0] == 0:
                    continue
                h = 1e-6 * max(1.0, abs(w[idx]))
                sd2 = dict(sd)
                wp = w.copy(); wp[idx] += h; sd2[n] = wp
                fp = f(sd2)
                wm = w.copy(); wm[idx] -= h; sd2[n] = wm
                fm = f(sd2)
                num = (fp - fm) / (2 * h)
                ana = np.asarray(info["grads"][n])[idx]
                assert abs(num - ana) < 2e-4 * max(1.0, abs(num)) + 1e-6, (n, idx, num, ana)
                checked += 1
        assert checked > 60
        if not embed:
            dpn = info["grad_pn_feat"]
            i, j = 4, 17
            h = 1e-5
            c2 = dict(cells); pf = cells["pn_feat"].astype(np.float64)
            a = pf.copy(); a[i, j] += h; c2["pn_feat"] = a
            fp = float((OT.encode_cells_train(c2, sd, embed, embed, **kw)[0] * gout).sum())
            a = pf.copy(); a[i, j] -= h; c2["pn_feat"] = a
            fm = float((OT.encode_cells_train(c2, sd, embed, embed, **kw)[0] * gout).sum())
            assert abs((fp - fm) / (2 * h) - dpn[i, j]) < 1e-6


def test_dropout_mask_statistics():
    for p in (0.1, 0.5):
        k = OT.dropout_keep(99, 3, 1 << 20, p)
        assert abs(k.mean() - (1 - p)) < 2e-3
        assert not np.array_equal(k, OT.dropout_keep(99, 4, 1 << 20, p))
        assert not np.array_equal(k, OT.dropout_keep(100, 3, 1 << 20, p))
    assert OT.dropout_keep(1, 1, 100, 0.0).all()


def test_pointnet_train_oracle_gradients_match_central_differences():
    """oracle/t2l_oracle_pointnet_train.py: analytic backward of the train-mode PointNet++ (per-cell BatchNorm statistics,
    max aggregation, the bipartite self-loop edge) against central differences of its own float64 forward."""
    from oracle import t2l_oracle_pointnet_train as OPT

    cells = synth.make_cells(2, seed=3, min_obj=2, max_obj=2)
    pos, rgb = synth.make_sampled_points(cells, 3)
    sd = synth.make_pointnet_weights(1)
    offs = cells["offsets"]
    rng = np.random.default_rng(0)
    R = rng.standard_normal((pos.shape[0], 256))
    f2, info = OPT.forward_backward(pos, rgb, offs, sd, grad_f2=R)
    assert f2.shape == (4, 256) and (f2 >= 0).all() and f2.max() > 0
    # running statistics moved (two cells -> two momentum updates) and stay finite
    k = "object_encoder.pointnet.sa2.point_conv.local_nn.0.1.running_var"
    assert np.isfinite(info["running"][k]).all() and np.abs(info["running"][k] - sd[k]).max() > 1e-6

    def loss(sd_mod):
        out, _ = OPT.forward_backward(pos, rgb, offs, sd_mod)
        return float((out * R).sum())

    checked = 0
    for name in ["object_encoder.pointnet.lin2.weight", "object_encoder.pointnet.lin1.bias",
                 "object_encoder.pointnet.ga.mlp.1.0.weight", "object_encoder.pointnet.ga.mlp.0.1.weight",
                 "object_encoder.pointnet.sa3.point_conv.local_nn.1.0.weight", "object_encoder.pointnet.sa3.point_conv.local_nn.0.1.bias",
                 "object_encoder.pointnet.sa2.point_conv.local_nn.0.0.weight", "object_encoder.pointnet.sa1.point_conv.local_nn.1.1.weight",
                 "object_encoder.pointnet.sa1.point_conv.local_nn.0.0.weight"]:
        g = np.asarray(info["grads"][name])
        flat = np.argsort(-np.abs(g).ravel())[:3]  # the three largest entries: well above the difference quotient's noise
        for idx in flat:
            base = np.asarray(sd[name], dtype=np.float64)
            best = np.inf
            for scale in (1e-6, 1e-7):  # an arg-max or ReLU switch inside +-eps shows as an O(1e-2) disagreement that a smaller
                eps = scale * max(1.0, abs(base.ravel()[idx]))  # step does not see; smooth points agree to ~1e-8
                vals = []
                for sgn in (+1, -1):
                    mod = dict(sd)
                    w = base.copy()
                    w.ravel()[idx] += sgn * eps
                    mod[name] = w
                    vals.append(loss(mod))
                num = (vals[0] - vals[1]) / (2 * eps)
                best = min(best, abs(num - g.ravel()[idx]) / max(1.0, abs(num)))
                if best < 1e-5:
                    break
            assert best < 1e-5, (name, idx, num, g.ravel()[idx])
            checked += 1
    assert checked == 27


# ---- the TEXT half of the step (oracle/t2l_oracle_text_train.py) ---------------------------------------------------------------
def _text_case(g):
    from oracle import t2l_oracle_text_train as OTT

    B, S, L = int(g["batch"]), int(g["n_hints"]), int(g["n_tokens"])
    sd = synth.make_language_head_weights(int(g["weight_seed"]))
    hidden = synth.make_t5_hidden(B * S, L, seed=int(g["hidden_seed"]))
    return OTT, sd, hidden, B


def text_loss_grad(head_out, cells, temperature):
    """d ContrastiveLoss(F.normalize(head_out), cells) / d head_out (cell_retrieval.py:57-63 + training/losses.py:269-283)."""
    y, n = OT._normalize_fwd(np.asarray(head_out, dtype=np.float64))
    loss, d_anchor, _ = O.contrastive_loss(y, cells, temperature, dtype=np.float64)
    return loss, y, OT._normalize_bwd(d_anchor, y, n)


def test_text_head_train_oracle_matches_the_reference_step(golden):
    g = golden("train_step_text")
    OTT, sd, hidden, B = _text_case(g)
    out0, _ = OTT.text_head_train(hidden, sd, B)
    assert np.abs(out0 - g["head_out"]).max() < 2e-5 * max(1.0, np.abs(g["head_out"]).max())
    loss, anchor, d_out = text_loss_grad(out0, g["cells"], float(g["temperature"]))
    assert np.abs(anchor - g["anchor"]).max() < 2e-6 and abs(loss - float(g["loss"])) < 2e-6 * max(1.0, abs(loss))
    _, info = OTT.text_head_train(hidden, sd, B, grad_out=d_out)
    used = [str(n) for n in g["used_params"]]
    assert sorted(used) == sorted(info["grads"].keys()) and len(used) == 28
    for n in used:
        exp, got = golden_view(g, "grad", n, info["grads"][n])
        rms = float(g[f"grad_norm/{n}"]) / np.sqrt(max(np.asarray(info["grads"][n]).size, 1))
        if n.endswith(("inter_mlp.0.0.bias", "intra_module.0.norm2.bias")):  # a constant per column in front of a BatchNorm: true gradient 0
            assert np.abs(got).max() < 1e-9 and np.abs(exp).max() < 1e-4, n
            continue
        if n.endswith("in_proj_bias"):  # the key bias has true gradient 0 (softmax is shift-invariant): compare where it is not noise
            D = len(got) // 3
            sel = np.r_[0:D, 2 * D:3 * D] if len(got) <= 1024 else None
            if sel is not None:
                exp, got = exp[sel], got[sel]
        err = np.abs(got - exp)
        assert (err < 1e-2 * rms + 1e-6).mean() >= 0.95 and err.max() < 0.2 * rms + 1e-5, (n, float(err.max()), rms)
    new = OT.bn_running_update(sd, info["bn_stats"])
    for k in g.files:
        if k.startswith("buf/"):
            assert np.allclose(np.asarray(new[k[4:]], dtype=np.float64), g[k], rtol=2e-5, atol=2e-6), k


def test_text_head_train_oracle_backward_is_the_derivative_of_its_forward(golden):
    """Central differences of the float64 forward with the dropout masks ON (p = 0.1), on a small case."""
    from oracle import t2l_oracle_text_train as OTT

    sd = {k: np.asarray(v, dtype=np.float64) for k, v in synth.make_language_head_weights(7).items()}
    hidden = synth.make_t5_hidden(4 * 3, 5, seed=3).astype(np.float64)
    rng = np.random.default_rng(0)
    G = rng.standard_normal((4, 256))

    def f(sd_):
        out, _ = OTT.text_head_train(hidden, sd_, 4, p_drop=0.1, seed=11)
        return float((out * G).sum())

    _, info = OTT.text_head_train(hidden, sd, 4, grad_out=G, p_drop=0.1, seed=11)
    checked = 0
    for name in ("language_encoder.intra_module.0.linear1.weight", "language_encoder.intra_module.0.self_attn.in_proj_weight",
                 "language_encoder.inter_mlp.0.0.weight", "language_encoder.inter_mlp.0.1.weight",
                 "language_encoder.inter_module.0.self_attn.out_proj.weight", "language_encoder.inter_module.0.norm2.bias",
                 "language_encoder.intra_module.0.norm1.weight"):
        w = sd[name]
        flat = w.reshape(-1)
        for i in rng.choice(flat.size, size=3, replace=False):
            old = flat[i]
            eps = 1e-6 * max(1.0, abs(old))  # (small: a ReLU or an arg-max switching inside the interval is the only way to fail)
            flat[i] = old + eps
            fp = f(sd)
            flat[i] = old - eps
            fm = f(sd)
            flat[i] = old
            num = (fp - fm) / (2 * eps)
            ana = float(np.asarray(info["grads"][name]).reshape(-1)[i])
            assert abs(num - ana) < 1e-5 * max(1.0, abs(ana)) + 1e-7, (name, i, num, ana)
            checked += 1
    assert checked == 21
