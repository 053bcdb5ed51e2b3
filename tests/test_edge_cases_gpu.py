"""Degenerate problems through the C ABI against BOTH restatements of the reference's loop (oracle state machine; the literal dense loop where
the reference itself does not NaN): start == goal, start next to the goal, opposite corners, the goal on an obstacle, nothing free but the
start, all costs zero (every priority ties: the first-index rule decides every step), g_ratio 0, a budget of ONE step -- on 32x32
(hand-scheduled stream), 20x45 (compiled LDS loop), 90x100 (large-map kernel), one-row / one-column maps (the reference's budget is W * W:
ONE step for a column), 2x2 and 1x1.  Bit-exact histories, paths, step counts and status (reference differentiable_astar.py:150-267)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(torch.device("cuda:0"))


def _check(name, m, s, g, cost=None, gr=0.5, T=None):
    from neural_astar import ops
    from oracle import oracle as O
    B, _, H, W = m.shape
    cost = m if cost is None else cost
    T = T or W * W
    o = O.forward(cost, s, g, m, gr, T, mode="sm")
    out = ops.search_nograd(_t(cost), _t(s), _t(g), _t(m), gr, T, want_log=True)
    torch.cuda.synchronize()
    h, p, it, st = (x.cpu().numpy() for x in out[:4])
    h, p = h.reshape(o.histories.shape), p.reshape(o.paths.shape)
    assert np.array_equal(h, o.histories) and np.array_equal(p, o.paths) and np.array_equal(it, o.iters), (name, H, W)
    solvable = (st == 0)
    assert bool(o.status) == bool((~solvable).any()), (name, H, W, st.tolist())
    if H * W <= 4096:
        od = O.forward(cost, s, g, m, gr, T, mode="dense")
        if not od.status:  # (the dense loop stops where the reference would NaN: an emptied open list)
            assert np.array_equal(h, od.histories) and np.array_equal(p, od.paths), (name, H, W, "dense")
    return st, it


@pytest.mark.parametrize("H,W", [(32, 32), (20, 45), (90, 100), (1, 40), (40, 1), (2, 2), (1, 1)])
def test_degenerate_problems_match_both_restatements(H, W):
    rng = np.random.default_rng(H * 100 + W)
    B = 3
    m = (rng.random((B, 1, H, W)) > 0.15).astype(np.float32)
    s, g = np.zeros_like(m), np.zeros_like(m)
    mid = ((H // 2) % H, (W // 2) % W)
    adj = (min(1, H - 1), min(1, W - 1))
    s[0, 0][mid] = 1; g[0, 0][mid] = 1; m[0, 0][mid] = 1                         # map 0: start == goal
    s[1, 0, 0, 0] = 1; g[1, 0][adj] = 1; m[1, 0, 0, 0] = 1; m[1, 0][adj] = 1     # map 1: neighbours (or the same cell on a 1x1 map)
    s[2, 0, 0, 0] = 1; g[2, 0, H - 1, W - 1] = 1; m[2, 0, 0, 0] = 1; m[2, 0, H - 1, W - 1] = 1   # map 2: opposite corners
    st, it = _check("start == goal / neighbours / corners", m, s, g)
    assert it[0] == 1 and st[0] == 0
    m2 = m.copy()
    m2[2, 0, H - 1, W - 1] = 0
    st, _ = _check("goal on an obstacle", m2, s, g)
    if H * W > 1 and W > 1:
        assert st[2] == 3  # (a one-column map has a budget of one step: the start is selected, the budget ends, nothing is unsolvable yet)
    m3 = np.zeros_like(m)
    for b in range(B):
        m3[b][s[b] > 0] = 1
    _check("only the start is free", m3, s, g)
    _check("all costs zero", m, s, g, cost=np.zeros_like(m))
    _check("g_ratio 0, costs 3", m, s, g, cost=(m * 3).astype(np.float32), gr=0.0)
    _, it = _check("budget of one step", m, s, g, T=1)
    assert (it == 1).all()


def test_the_reference_modules_public_helpers_exist_and_agree():
    """planner.differentiable_astar.get_heuristic / expand / backtrack (reference differentiable_astar.py:26-52, :77-93, :96-125): the search kernel
    fuses them, but code that imports them from the reference's module finds them here with the same arguments and results (validated against the
    LIVE reference in the authoring container; here against the oracle's heuristic and plain numpy restatements)."""
    from neural_astar.planner import differentiable_astar as M
    from oracle import oracle as O
    dev = torch.device("cuda:0")
    B, H, W = 3, 20, 45
    goal = np.zeros((B, H, W), np.float32)
    cells = [(0, 0), (19, 44), (7, 13)]
    for b, (r, c) in enumerate(cells):
        goal[b, r, c] = 1
    h = M.get_heuristic(_t(goal)).cpu().numpy()
    for b, (r, c) in enumerate(cells):
        assert np.array_equal(h[b].view(np.uint32), O.heuristic(H, W, r, c).view(np.uint32))  # bit-exact, default tie-break factor
    h2 = M.get_heuristic(_t(goal[:, None]), 0.01).cpu().numpy()  # another factor, [B,1,H,W] in -> same shape out
    assert h2.shape == (B, 1, H, W)
    rr, cc = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    for b, (r, c) in enumerate(cells):
        dr, dc = np.abs(rr - np.float32(r)), np.abs(cc - np.float32(c))
        ref = ((dr + dc) - np.minimum(dr, dc)) + np.float32(0.01) * np.sqrt((rr - r) ** 2 + (cc - c) ** 2, dtype=np.float32)
        assert np.array_equal(h2[b, 0], ref.astype(np.float32))
    rng = np.random.default_rng(0)
    x = (rng.random((B, H, W)) > 0.7).astype(np.float32)
    nf = np.ones((B, 1, 3, 3), np.float32)
    nf[:, 0, 1, 1] = 0
    y = M.expand(_t(x), _t(nf)).cpu().numpy()
    pad = np.pad(x, ((0, 0), (1, 1), (1, 1)))
    ref = sum(pad[:, 1 + dy:1 + dy + H, 1 + dx:1 + dx + W] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dy, dx) != (0, 0))
    assert np.array_equal(y, ref)
    assert M.expand(_t(x[:1]), _t(nf[:1])).shape == (H, W)  # (the reference's squeezes: a batch of one loses its batch dimension)
    gi = goal.reshape(B, -1).argmax(1)
    parents = np.tile(gi[:, None].astype(np.float32), (1, H * W))  # the reference's initial table: every cell points at the goal
    chain = []
    for b in range(B):
        cur = int(gi[b])
        cells_b = []
        for _ in range(5):
            nxt = (cur + W + 1) % (H * W)
            parents[b, cur] = nxt
            cells_b.append(nxt)
            cur = nxt
        chain.append(cells_b)
    start = np.zeros((B, H, W), np.float32)
    for T in (0, 1, 3, 9):
        p = M.backtrack(_t(start), _t(goal), _t(parents), T)
        assert p.dtype == torch.int64 and tuple(p.shape) == (B, H, W)
        p = p.cpu().numpy().reshape(B, -1)
        for b in range(B):
            want = np.zeros(H * W, np.int64)
            want[gi[b]] = 1
            loc = int(parents[b, gi[b]])
            for _ in range(T):
                want[loc] = 1
                loc = int(parents[b, loc])
            assert np.array_equal(p[b], want), (T, b)
