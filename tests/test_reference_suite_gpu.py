"""The reference's own test-suite (tests/astar_test.py:5-53), re-homed on the device.

Same fixture, same four scenarios; the only change is ``.cuda()`` and that ``test_pq_astar`` compares against the
reference's CPU answer stored in tests/golden/ (the reference's assertion is ``DifferentiableAstar == pq_astar`` on this
fixture, and ``use_differentiable_astar=False`` -- the CPU-only pq_astar -- is out of scope here and must say so)."""
import numpy as np
import pytest
import torch

import golden_util as G

pytestmark = pytest.mark.gpu


@pytest.fixture
def setup():
    dev = torch.device("cuda:0")
    map_designs = torch.ones((8, 1, 64, 64), device=dev)
    map_designs[:, :, 24:48, 24:48] = 0
    start_maps = torch.zeros((8, 1, 64, 64), device=dev)
    start_maps[:, :, 0, 0] = 1
    goal_maps = torch.zeros((8, 1, 64, 64), device=dev)
    goal_maps[:, :, -1, -1] = 1
    return map_designs, start_maps, goal_maps


def test_neural_astar(setup):
    from neural_astar.planner import NeuralAstar
    map_designs, start_maps, goal_maps = setup
    planner = NeuralAstar().cuda()  # train mode, Tmax = 1.0, CNN encoder (reference defaults)
    output = planner(map_designs, start_maps, goal_maps)
    assert output.histories.shape == (8, 1, 64, 64) and output.histories.requires_grad
    assert output.paths.dtype == torch.int64 and output.intermediate_results == []


def test_vanilla_astar(setup):
    from neural_astar.planner import VanillaAstar
    map_designs, start_maps, goal_maps = setup
    output = VanillaAstar().cuda()(map_designs, start_maps, goal_maps)
    assert int(output.histories[0].sum()) == 1169 and int(output.paths[0].sum()) == 88


def test_pq_astar(setup):
    from neural_astar.planner import VanillaAstar
    map_designs, start_maps, goal_maps = setup
    output = VanillaAstar(use_differentiable_astar=True).cuda()(map_designs, start_maps, goal_maps)
    g = G.load("fixture64_g050")  # == the reference's pq_astar answer on this fixture (its own assertion)
    assert np.array_equal(output.histories.cpu().numpy(), np.repeat(g.histories[:1], 8, 0))
    assert np.array_equal(output.paths.cpu().numpy(), np.repeat(g.paths[:1], 8, 0))
    with pytest.raises(NotImplementedError):
        VanillaAstar(use_differentiable_astar=False).cuda()(map_designs, start_maps, goal_maps)


def test_astar_on_rectangle(setup):
    from neural_astar.planner import NeuralAstar, VanillaAstar
    map_designs, start_maps, goal_maps = setup
    map_designs = torch.concat((map_designs, map_designs), -1)
    start_maps = torch.concat((start_maps, torch.zeros_like(start_maps)), -1)
    goal_maps = torch.concat((torch.zeros_like(goal_maps), goal_maps), -1)
    import warnings
    planner = NeuralAstar().cuda()
    with warnings.catch_warnings():
        warnings.simplefilter("error")  # (a fall-through to torch.nn would warn)
        output = planner(map_designs, start_maps, goal_maps)
    assert output.histories.shape == (8, 1, 64, 128)
    # round 6: the reference's own rectangle scenario -- 64 x 128, training mode, gradients on -- runs on the MI355X encoder kernels (2-D conv
    # tiles beyond 126 pixels per row, weight-gradient chunks of row segments), not on torch.nn / MIOpen
    assert planner.last_encoder_route.startswith("hip:CNN-train"), planner.last_encoder_route
    output.histories.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in planner.encoder.parameters() if p.requires_grad)
    planner.eval()
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("error")
        planner(map_designs, start_maps, goal_maps)
    assert planner.last_encoder_route.startswith("hip:CNN-infer"), planner.last_encoder_route
    g = G.load("rect64x128_g050")  # the search-only part of this scenario, reference answer
    out_v = VanillaAstar().cuda()(map_designs, start_maps, goal_maps)
    assert np.array_equal(out_v.histories.cpu().numpy(), np.repeat(g.histories[:1], 8, 0))
    assert np.array_equal(out_v.paths.cpu().numpy(), np.repeat(g.paths[:1], 8, 0))
