"""GPU parity of the large-map search (maps whose state does not fit LDS: 129x129 ... 512x512; csrc/nastar_search_hybrid.hip.h -- open list
in LDS, cells in HBM; round 6: up to 1,179,648 cells -- 1024x1024 included -- with several super-chunk entries per lane), the replacement for the reference's advice to leave the differentiable search for the CPU pq_astar on large maps
(astar.py:36-37).  Against the oracle's state-machine restatement, long searches (tens of thousands of steps) included.  Bit-exact histories,
paths, step counts, logs.  (Rounds 4-5 cross-checked the long searches against the round-4 all-HBM kernel; that kernel left the library in
round 6 and the oracle, a few seconds per map here, took its place.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "gpu-marked test needs a HIP device"
    return torch.device("cuda:0")


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(_dev())


def _near_pairs(B, H, W, p, seed, reach):
    """random-obstacle maps with a start and a goal about `reach` cells apart in one 8-connected component (short searches: the oracle scans
    every cell per step)"""
    from neural_astar.utils import synthetic as syn
    rng = np.random.Generator(np.random.PCG64(seed))
    m = (rng.random((B, H, W)) > p)
    s = np.zeros((B, 1, H, W), np.float32)
    g = np.zeros((B, 1, H, W), np.float32)
    for b in range(B):
        while True:
            sr, sc = int(rng.integers(H // 8, H - H // 8)), int(rng.integers(W // 8, W - W // 8))
            gr = int(np.clip(sr + rng.integers(-reach, reach + 1), 0, H - 1))
            gc = int(np.clip(sc + rng.integers(reach // 2, reach + 1) * (1 if rng.random() < 0.5 else -1), 0, W - 1))
            if not (m[b, sr, sc] and m[b, gr, gc]) or (sr, sc) == (gr, gc):
                continue
            d = syn.geodesic_distance(m[b:b + 1], np.array([gr * W + gc]))
            if d[0, sr, sc] > 0:
                break
        s[b, 0, sr, sc] = 1
        g[b, 0, gr, gc] = 1
    return m.astype(np.float32)[:, None], s, g


def _run(cost, s, g, passable, max_iters, flags=0, log=True):
    from neural_astar import ops
    out = ops.search_nograd(_t(cost), _t(s), _t(g), _t(passable), 0.5, max_iters, want_log=log, flags=flags)
    torch.cuda.synchronize()
    return [x.cpu().numpy() if x is not None else None for x in out]


@pytest.mark.parametrize("H,W,p,Tmax,reach", [(256, 256, 0.2, 1.0, 60), (512, 512, 0.2, 1.0, 80), (512, 512, 0.25, 0.05, 200), (300, 170, 0.15, 1.0, 50),
                                               (140, 140, 0.2, 1.0, 40), (1024, 1024, 0.2, 1.0, 400), (700, 1100, 0.15, 1.0, 300)])
def test_large_maps_match_the_oracle(H, W, p, Tmax, reach):
    from neural_astar import ops
    from neural_astar.utils import synthetic as syn
    from oracle import oracle as O
    B = 2
    m, s, g = _near_pairs(B, H, W, p, seed=H * 7 + W, reach=reach)
    max_iters = ops.max_iters_for(W, Tmax, Tmax < 1.0)
    assert ops.workspace_bytes((B, H, W)) > 0  # not an LDS-resident size
    for cost in (m, syn.random_costs(B, H, W, seed=11)):
        o = O.forward(cost, s, g, m, 0.5, max_iters, mode="sm", want_log=True)
        hist, paths, iters, status, log = _run(cost, s, g, m, max_iters)
        assert (status == 0).all() and np.array_equal(iters, o.iters), (iters, o.iters)
        assert np.array_equal(hist, o.histories) and np.array_equal(paths, o.paths)
        for b in range(B):
            assert np.array_equal(log[b, :iters[b]], o.sel_log[b, :iters[b]])
        if Tmax < 1.0:
            assert (iters <= max_iters).all()


def test_long_searches_match_the_oracle_and_unsolvable_maps_are_flagged():
    """400x400 mazes (tens of thousands of steps) and 512x512 random-obstacle maps with start and goal anywhere."""
    from neural_astar import ops
    from oracle import oracle as O
    from neural_astar.planner import VanillaAstar
    from neural_astar.planner.differentiable_astar import UnsolvableMapError
    from neural_astar.utils import synthetic as syn
    longest = 0
    for pr in (syn.maze_maps(2, 400, seed=3), syn.random_obstacle_maps(4, 512, 512, 0.2, seed=3)):
        B, _, H, W = pr.map_designs.shape
        cost = syn.random_costs(B, H, W, seed=5)
        for c in (pr.map_designs, cost):
            a = _run(c, pr.start_maps, pr.goal_maps, pr.map_designs, W * W)
            o = O.forward(c, pr.start_maps, pr.goal_maps, pr.map_designs, 0.5, W * W, mode="sm", want_log=True)
            assert (a[3] == 0).all() and np.array_equal(a[2], o.iters)
            assert np.array_equal(a[0], o.histories) and np.array_equal(a[1], o.paths)
            for b in range(B):
                assert np.array_equal(a[4][b, :a[2][b]], o.sel_log[b, :o.iters[b]])
            longest = max(longest, int(a[2].max()))
    assert longest > 5000, longest
    # a wall: unsolvable, reported per map and through the status summary of the planner (stream-wait fall-back: no completion flag here)
    m = np.ones((2, 1, 200, 200), np.float32)
    m[0, 0, 100, :] = 0
    s = np.zeros_like(m)
    g = np.zeros_like(m)
    s[:, 0, 5, 5] = 1
    g[:, 0, 190, 190] = 1
    va = VanillaAstar().to(_dev()).eval()
    with pytest.raises(UnsolvableMapError):
        va(_t(m), _t(s), _t(g))
    assert va.astar.last_status.tolist() == [3, 0]
    va.astar.check_solvable = "deferred"
    va(_t(m), _t(s), _t(g))
    with pytest.raises(UnsolvableMapError):
        va.astar.raise_if_unsolvable()
    out = va(_t(m[1:]), _t(s[1:]), _t(g[1:]))
    va.astar.raise_if_unsolvable()
    assert int(out.paths.sum()) == 186  # the diagonal 5,5 -> 190,190


@pytest.mark.parametrize("H,W,reach,train", [(260, 270, 30, False), (512, 512, 25, False), (300, 260, 40, True), (1024, 1024, 25, False)])
def test_gradients_on_maps_above_65519_cells_match_the_oracle(H, W, reach, train):
    """The replay backward on maps whose history stamps need 32 bits (round 6: every size the forward takes; rounds 2-5 stopped at 65,519
    cells): DifferentiableAstar under autograd -- hybrid forward with a selection log, HBM-state replay with wide stamps -- against the
    oracle's literal reverse mode of the reference's graph (differentiable_astar.py:203-252), 1e-5 of the gradient's scale.  Short searches
    (start and goal `reach` cells apart; the training case stops at a budget of 40 steps, before either map reaches its goal): the oracle scans every cell per step."""
    from neural_astar.planner.differentiable_astar import DifferentiableAstar
    from neural_astar.utils import synthetic as syn
    from oracle import oracle as O
    B = 2
    m, s, g = _near_pairs(B, H, W, 0.15, seed=H + W, reach=reach)
    cost = syn.random_costs(B, H, W, seed=5, hi=2.0)
    Tmax = 40.5 / (W * W) if train else 1.0
    T = int(Tmax * W * W) if train else W * W
    fw = O.forward(cost, s, g, m, 0.5, T, mode="sm")
    assert fw.status == 0 if np.isscalar(fw.status) else not np.any(fw.status)
    Tref = int(fw.iters.max())  # the dense reverse mode needs the batch's steps only
    up = np.random.Generator(np.random.PCG64(3)).standard_normal((B, 1, H, W)).astype(np.float32)
    ref = O.backward(up, cost, s, g, m, 0.5, T if train else Tref)
    da = DifferentiableAstar(0.5, Tmax).to(_dev()).train(train)
    c = _t(cost).requires_grad_(True)
    out = da(c, _t(s), _t(g), _t(m))
    (out.histories * _t(up)).sum().backward()
    assert np.array_equal(out.histories[:, 0].detach().cpu().numpy(), fw.histories) and np.array_equal(da.last_iters.cpu().numpy(), fw.iters)
    err = float(np.abs(c.grad[:, 0].cpu().numpy() - ref).max())
    assert err <= 1e-5 * max(1.0, float(np.abs(ref).max())), (err, float(np.abs(ref).max()))
    assert float(np.abs(ref).max()) > 0


def test_gradient_above_65519_cells_equals_the_reference_golden():
    """wide_grad_260x270: the reference's own histories, paths and autograd gradient on 260x270 maps (oracle/gen_golden_large_grad.py) -- the
    hybrid forward with a selection log and the HBM-state replay with 32-bit history stamps, through DifferentiableAstar under autograd"""
    from neural_astar.planner.differentiable_astar import DifferentiableAstar
    from test_oracle_golden import wide_golden
    g, cost, up, grad_ref = wide_golden()
    da = DifferentiableAstar(g.g_ratio, 1.0).to(_dev()).eval()
    c = _t(cost).requires_grad_(True)
    out = da(c, _t(g.start_maps), _t(g.goal_maps), _t(g.passable))
    (out.histories * _t(up)).sum().backward()
    assert np.array_equal(out.histories.detach().cpu().numpy(), g.histories) and np.array_equal(out.paths.cpu().numpy(), g.paths)
    ref = grad_ref.reshape(c.grad.shape)
    err = float(np.abs(c.grad.cpu().numpy() - ref).max())
    assert err <= 1e-5 * max(1.0, float(np.abs(ref).max())), err
