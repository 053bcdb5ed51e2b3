"""The training LOOP (not one step) against the reference's own loop.

Golden = three consecutive iterations of what ``scripts/train.py:43-50`` runs: the reference package's ``PlannerModule.training_step``
(utils/training.py:55-61) + ``configure_optimizers()``'s ``torch.optim.RMSprop`` (utils/training.py:52-53), a different 16-map batch per
step, starting from the shipped ``mazes_032`` checkpoint (``oracle/gen_golden_trainstep.py loop`` -> ``trainloop_maze32_3steps.npz``).
Here: a DEFAULT-constructed ``NeuralAstar`` (``encoder_backend = "auto"`` -> the MFMA training kernels, asserted) driven by this package's
``PlannerModule.training_step`` + ``configure_optimizers()`` (``FusedRMSprop``).

Two passes (see the test's docstring): free-running (step 0 asserted, the later steps REPORTED) and teacher-forced (every step
restarts from the reference's parameters and BatchNorm buffers; asserted).  Bars per asserted step: loss within 1e-6, cost maps within 1e-5, histories / paths identical (every map's selection margin is > 2e-5 in the
golden), BatchNorm running statistics within 1e-5, ``num_batches_tracked`` equal.  Parameters after each update: RMSprop divides every
element's gradient by its own running magnitude (+ 1e-8), so an element's update is NOT a smooth function of the gradient tensor: where
|g| is within a few orders of 1e-7 -- or of the rounding noise of the backward pass -- a gradient error far inside the step test's bar
(1e-4 of the tensor's maximum) moves the update by up to the whole step, 0.01.  The test therefore states both: the fraction of elements
within 1e-5 of the tensor's parameter range (reported, per step), and a per-element bound that every element must meet: 1e-5 of the
range + 1.5 x the deviation RMSprop ITSELF produces when the reference's gradients of steps 0..k are all shifted by +eps or by -eps,
eps = 1e-4 of that step's tensor maximum + 5e-8 (the optimiser is simulated in float64 on the stored gradients; the absolute part is
the measured distance of single gradient elements between ANY device path and the CPU reference, see ABS_G).  Elements whose reference
gradient is below eps ("sign-noise" elements: the direction of the reference's own update is rounding noise there) are counted and
reported, not hidden; conv biases in front of a BatchNorm (true gradient exactly 0) are handled apart."""
import os
import types

import numpy as np
import pytest
import torch

import golden_util as G

pytestmark = pytest.mark.gpu
NAME = "trainloop_maze32_3steps"
EPS_G = 1e-4  # the gradient accuracy the one-step golden test grants per tensor (max |diff| / max |ref|)
# ... plus an ABSOLUTE floor: with the loss a mean over 16 x 1024 cells the weight gradients of this configuration peak at 1e-6 .. 3e-5, i.e.
# INSIDE the range where RMSprop's eps = 1e-8 matters (|g| ~ 1e-8 .. 1e-6), and the device paths (MFMA kernels and torch.nn alike: they share
# the search's dL/dcost, 2e-6 of its maximum from the reference's autograd) sit ~2e-8 from the CPU reference on single elements (measured)
ABS_G = 5e-8


def _rmsprop_paths(gs, lr, alpha=0.99, eps=1e-8):
    """parameter displacement after each step of torch.optim.RMSprop (no momentum, not centered) for the gradient sequence `gs`"""
    v = np.zeros_like(gs[0])
    p = np.zeros_like(gs[0])
    out = []
    for g in gs:
        v = alpha * v + (1.0 - alpha) * g * g
        p = p - lr * g / (np.sqrt(v) + eps)
        out.append(p.copy())
    return out


def _dev():
    assert torch.cuda.is_available(), "gpu-marked test needs a HIP device"
    return torch.device("cuda:0")


def _unpack(bits, B, H, W):
    return np.unpackbits(bits, axis=1)[:, :H * W].reshape(B, 1, H, W)


def _onehot(idx, B, H, W):
    m = np.zeros((B, H * W), np.float32)
    m[np.arange(B), idx] = 1
    return m.reshape(B, 1, H, W)


def _run_loop(backend, forced):
    """Three optimiser steps through this package's PlannerModule.training_step + configure_optimizers().  forced=False: free-running.
    forced=True: after step k's comparison the reference's parameters and BatchNorm buffers of step k are copied into the planner
    (the optimiser keeps ITS OWN state), so that every step is compared from the reference's own trajectory."""
    from neural_astar.planner import NeuralAstar
    from neural_astar.utils import training as T
    from neural_astar.utils.optim import FusedRMSprop
    from neural_astar import encoder_train as ET
    dev = _dev()
    z = np.load(os.path.join(G.GOLDEN_DIR, NAME + ".npz"))
    B, H, W, n_steps = int(z["B"]), int(z["H"]), int(z["W"]), int(z["n_steps"])
    ck = np.load(os.path.join(G.GOLDEN_DIR, "ckpt_mazes032_cnn.npz"))
    na = NeuralAstar(encoder_input="m+", encoder_arch="CNN", encoder_depth=4, Tmax=0.25)  # scripts/train.py:33-39, no backend knob
    assert na.encoder_backend == "auto"
    na.load_state_dict({k: torch.from_numpy(ck[k]) for k in ck.files}, strict=True)
    na = na.to(dev)
    na.encoder_backend = backend
    lr = float(z["lr"])
    module = T.PlannerModule(na, types.SimpleNamespace(params=types.SimpleNamespace(lr=lr)))
    module.train()
    opt = module.configure_optimizers()
    assert isinstance(opt, FusedRMSprop) and isinstance(opt, torch.optim.RMSprop)
    trunk_calls, seen = [], {}
    orig_trunk, orig_step, enc = ET.cnn_train_forward, T.fused_l1_step, na.encode

    def spy_trunk(*a, **k):  # what the default-constructed planner actually runs: the MFMA training trunk, not torch.nn
        trunk_calls.append(a[-1] if a else None)
        return orig_trunk(*a, **k)

    def spy_step(*a, **k):
        loss, out = orig_step(*a, **k)
        seen["out"] = out
        return loss, out

    def encode(*a, **k):
        c = enc(*a, **k)
        seen["cost"] = c.detach()
        return c
    ET.cnn_train_forward, T.fused_l1_step, na.encode = spy_trunk, spy_step, encode
    conv_bias_before_bn = {n + ".bias" for n, mod in na.named_modules() if isinstance(mod, torch.nn.Conv2d)}
    report, violations = [], []
    try:
        for k in range(n_steps):
            t = f"step{k}/"
            m = torch.from_numpy(_unpack(z[t + "map_bits"], B, H, W).astype(np.float32)).to(dev)
            s = torch.from_numpy(_onehot(z[t + "start_idx"], B, H, W)).to(dev)
            g = torch.from_numpy(_onehot(z[t + "goal_idx"], B, H, W)).to(dev)
            traj = torch.from_numpy(_unpack(z[t + "traj_bits"], B, H, W).astype(np.float32)).to(dev)
            assert float(z[t + "sel_margin"].min()) > 2e-5
            opt.zero_grad()
            loss = module.training_step((m, s, g, traj), k)
            loss.backward()
            opt.step()
            torch.cuda.synchronize()
            rec = dict(step=k, loss_err=abs(float(loss) - float(z[t + "loss"])),
                       cost_err=float((seen["cost"].cpu() - torch.from_numpy(z[t + "cost"])).abs().max()),
                       maps_with_other_history=int((seen["out"].histories.cpu().numpy() != _unpack(z[t + "hist_bits"], B, H, W)).any(axis=(1, 2, 3)).sum()),
                       maps_with_other_path=int((seen["out"].paths.cpu().numpy() != _unpack(z[t + "path_bits"], B, H, W)).any(axis=(1, 2, 3)).sum()))
            n_el = n_in = n_noise = 0
            worst_rel = bias_dev = buf_err = 0.0
            for name, p in na.named_parameters():
                if not p.requires_grad:
                    continue
                if name in conv_bias_before_bn:
                    # a bias in front of a BatchNorm has NO effect on the output: its true gradient is exactly 0.  The reference's autograd
                    # leaves rounding noise there (|g| ~ 1e-10, asserted), which RMSprop's division by sqrt(v) + 1e-8 turns into updates
                    # of up to 1e-4 in a direction that is noise; the HIP path returns exact zeros and leaves these biases where they were
                    assert float(z[t + "gradmax/" + name]) <= 1e-7, (name, float(z[t + "gradmax/" + name]))
                    bias_dev = max(bias_dev, float((p.detach().cpu() - torch.from_numpy(z[t + "param/" + name])).abs().max()))
                    continue
                ref = torch.from_numpy(z[t + "param/" + name]).double()
                d = (p.detach().double().cpu() - ref).abs()
                scale = float(ref.abs().max().clamp_min(1e-30))
                tol0 = 1e-5 * scale
                gs, es = [], []
                for j in range(k + 1):  # the reference's gradients of steps 0..k (fp16 of g / max|g| + the maximum: 5e-4 relative, ample here)
                    tj = f"step{j}/"
                    mx = float(z[tj + "gradmax/" + name])
                    gs.append(z[tj + "grad16/" + name].astype(np.float64) * mx)
                    # what a gradient element may be off by: the step test's bar + the absolute floor + the fp16 quantum of the STORED
                    # reference gradient itself (2^-11 relative: it matters from step 1 on, where the update is linear in g / sqrt(v))
                    es.append(EPS_G * mx + ABS_G + 1e-3 * np.abs(gs[-1]))
                import itertools
                signs = [(0.0,) * (k + 1)] + list(itertools.product((1.0, -1.0), repeat=k + 1))  # every sign pattern over the steps
                paths = [_rmsprop_paths([g_ + sg * e_ for g_, e_, sg in zip(gs, es, pat)], lr) for pat in signs]
                # (forced runs restart every step from the reference's parameters: only step k's own displacement counts)
                disp = [(pp[-1] - (pp[-2] if (forced and k > 0) else 0.0)) for pp in paths]
                spread = np.max(np.stack([np.abs(dd - disp[0]) for dd in disp[1:]]), axis=0)
                bound = tol0 + 1.5 * torch.from_numpy(spread)
                bad = d > bound
                if bool(bad.any()):
                    i = int(torch.argmax((d - bound).flatten()))
                    violations.append(dict(step=k, name=name, n=int(bad.sum()), of=d.numel(), excess=float((d - bound).flatten()[i]),
                                           d=float(d.flatten()[i]), bound=float(bound.flatten()[i]),
                                           g_ref=float(gs[-1].flatten()[i]), gradmax=float(z[t + 'gradmax/' + name])))
                n_el += d.numel()
                n_in += int((d <= tol0).sum())
                n_noise += int((np.abs(gs[-1]) < EPS_G * (es[-1].max() if hasattr(es[-1], 'max') else es[-1])).sum())
                worst_rel = max(worst_rel, float(d.max()) / scale)
            for name, b in na.named_buffers():
                ref = torch.from_numpy(np.asarray(z[t + "buffer/" + name]))
                if b.dtype.is_floating_point:
                    buf_err = max(buf_err, float((b.double().cpu() - ref.double()).abs().max() / ref.double().abs().max().clamp_min(1e-30)))
                else:
                    assert int(b) == int(ref), (k, name)  # num_batches_tracked
            rec.update(params_within_1e5=n_in / n_el, sign_noise_elements=n_noise, elements=n_el, worst_param_dev_rel=worst_rel,
                       zero_gradient_bias_dev=bias_dev, bn_buffer_err=buf_err)
            report.append(rec)
            if forced:  # continue from the reference's own trajectory
                with torch.no_grad():
                    for name, p in na.named_parameters():
                        p.copy_(torch.from_numpy(z[t + "param/" + name]).to(dev))
                    for name, b in na.named_buffers():
                        b.copy_(torch.from_numpy(np.asarray(z[t + "buffer/" + name])).to(dev))
    finally:
        ET.cnn_train_forward, T.fused_l1_step = orig_trunk, orig_step
        del na.encode
    return report, violations, trunk_calls, n_steps


@pytest.mark.parametrize("backend", ["auto", "torch"])
def test_three_steps_of_the_training_loop_match_the_reference_loop(backend):
    """see the module docstring.  Two passes: FREE-RUNNING (reported: after the first RMSprop update the optimiser's division by
    sqrt(v) + 1e-8 has amplified rounding-level gradient differences on near-zero-gradient elements into parameter differences of up
    to ~3e-4, so the cost maps of step 1 differ by ~7e-5 -- for the torch.nn encoder on the device exactly as for the MFMA kernels:
    the property is the optimiser's) and TEACHER-FORCED (asserted: every step starts from the reference's own parameters and BatchNorm
    buffers, the optimiser state stays this package's own)."""
    free, _, _, _ = _run_loop(backend, forced=False)
    print("TRAINLOOP free-running", backend, free)
    assert free[0]["cost_err"] <= 1e-5 and free[0]["loss_err"] <= 1e-6 and free[0]["maps_with_other_history"] == 0
    # measured (both backends alike): cost maps 9e-7 -> 7e-5 -> 8e-4 off the reference's after 0 / 1 / 2 updates, while EVERY map of
    # every step still takes the reference's route (margins > 2e-5 did not protect against 8e-4 by construction -- the routes simply
    # agree) and the losses are identical; asserted loosely, because this pass documents the optimiser's sensitivity
    assert all(r["cost_err"] <= 5e-3 and r["loss_err"] <= 2e-3 for r in free), free
    report, violations, trunk_calls, n_steps = _run_loop(backend, forced=True)
    print("TRAINLOOP teacher-forced", backend, report)
    print("TRAINLOOP violations", backend, violations)
    for r in report:
        assert r["cost_err"] <= 1e-5, r
        assert r["maps_with_other_history"] == 0 and r["maps_with_other_path"] == 0, r
        assert r["loss_err"] <= 1e-6, r
        assert r["bn_buffer_err"] <= 1e-5, r
        assert r["zero_gradient_bias_dev"] <= 2e-4, r
    assert not violations, violations
    if backend == "auto":
        assert len(trunk_calls) == n_steps and all(p == "f16x3" for p in trunk_calls), trunk_calls  # the MFMA training trunk ran every step
    else:
        assert not trunk_calls
