"""CPU pins of this package's re-written torch modules (planner/encoder.py, NeuralAstar.encode) against the REFERENCE's own classes.

The goldens were produced by importing the reference package itself (oracle/gen_golden_trainstep.py): its ``CNN`` / ``CNNDownSize``
(planner/encoder.py:60-97) through its ``NeuralAstar.encode`` (planner/astar.py:154-180) in eval mode, and one full training step
(utils/training.py:55-61).  Here the same state dicts must load ``strict=True`` into this package's classes (identical keys and
shapes) and the same torch ops must reproduce the reference's numbers: cost maps, and -- fed the reference's dL/dcost -- every
parameter gradient and the BatchNorm running statistics of the training step.  (The search itself has no CPU path in the product;
its parity tests are the ``-m gpu`` ones.)"""
import numpy as np
import pytest
import torch

import golden_util as G


def _planner(cfg, init):
    from neural_astar.planner import NeuralAstar
    na = NeuralAstar(**cfg)
    na.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in init.items()}, strict=True)
    return na


@pytest.mark.parametrize("name", sorted(G.ENC_CONFIGS))
def test_encoder_modules_reproduce_the_reference_cost_maps(name):
    g = G.load_enc(name)
    na = _planner(G.ENC_CONFIGS[name], g.init).eval()
    with torch.no_grad():
        cost = na.encode(torch.from_numpy(g.map_designs), torch.from_numpy(g.start_maps), torch.from_numpy(g.goal_maps))
    assert cost.shape == g.cost.shape
    err = float((cost - torch.from_numpy(g.cost)).abs().max())
    assert err <= 1e-6 * max(1.0, g.const), (name, err)


@pytest.mark.parametrize("name", ["trainstep_maze32", "trainstep_warcraft12"])
def test_encoder_training_mode_reproduces_the_reference_step(name):
    g = G.load_step(name)
    na = _planner(G.STEP_CONFIGS[name], g.init).train()
    cost = na.encode(torch.from_numpy(g.map_designs), torch.from_numpy(g.start_maps), torch.from_numpy(g.goal_maps))
    scale = float(np.abs(g.cost).max())
    assert float((cost.detach() - torch.from_numpy(g.cost)).abs().max()) <= 2e-6 * max(1.0, scale)
    cost.backward(torch.from_numpy(g.grad_cost))
    seen = 0
    for k, p in na.named_parameters():
        if k not in g.grads:
            assert p.grad is None or not p.requires_grad, k
            continue
        ref = torch.from_numpy(g.grads[k])
        tol = 1e-5 * float(ref.abs().max()) + 1e-9  # conv biases in front of a BatchNorm carry pure rounding noise (~1e-10)
        assert float((p.grad - ref).abs().max()) <= tol, (k, float((p.grad - ref).abs().max()), tol)
        seen += 1
    assert seen == len(g.grads) > 0
    for k, b in na.named_buffers():
        ref = torch.from_numpy(np.asarray(g.after[k]))
        if b.dtype.is_floating_point:
            assert float((b - ref).abs().max()) <= 1e-6 * max(1.0, float(ref.abs().max())), k
        else:
            assert int(b) == int(ref), k


def test_step_goldens_are_self_consistent():
    """loss == mean|histories - opt_trajs| (training.py:58) and dL/dcost is non-trivial on every stored step."""
    for name in G.STEP_CONFIGS:
        g = G.load_step(name)
        loss = float(np.abs(g.histories - g.opt_trajs).mean(dtype=np.float64))
        assert abs(loss - g.loss) <= 1e-6, (name, loss, g.loss)
        assert float(np.abs(g.grad_cost).max()) > 0
        assert g.histories.reshape(g.B, -1).sum(1).max() <= int(g.Tmax * g.W * g.W)
