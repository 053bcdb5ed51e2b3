"""Seeded synthetic planning problems (host-side data prep, numpy only).

The reference's datasets are absent (``planning-datasets`` is an un-vendored submodule,
reference ``.gitmodules:1-3``), so every input used by the tests and by ``bench.py`` is
synthesised here.  All generators guarantee an 8-connected start->goal route because the
reference crashes on unsolvable maps.

Tensor conventions follow the reference's DataLoader (``utils/data.py:137-166``):
``[B,1,H,W] float32``; 1 = passable / start / goal.
"""
from __future__ import annotations

from typing import NamedTuple, Optional

import numpy as np


class Problems(NamedTuple):
    map_designs: np.ndarray  # [B,1,H,W] float32, 1 = passable
    start_maps: np.ndarray  # [B,1,H,W] float32 one-hot
    goal_maps: np.ndarray  # [B,1,H,W] float32 one-hot


def _rng(seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64(seed))


def _one_hot(idx: np.ndarray, H: int, W: int) -> np.ndarray:
    B = idx.shape[0]
    m = np.zeros((B, H * W), np.float32)
    m[np.arange(B), idx] = 1.0
    return m.reshape(B, 1, H, W)


def _dilate8(x: np.ndarray) -> np.ndarray:
    """8-neighbour dilation of a [B,H,W] bool array (zero padded, no wrap-around)."""
    p = np.pad(x, ((0, 0), (1, 1), (1, 1)))
    out = np.zeros_like(x)
    H, W = x.shape[1:]
    for dr in (0, 1, 2):
        for dc in (0, 1, 2):
            if dr == 1 and dc == 1:
                continue
            out |= p[:, dr:dr + H, dc:dc + W]
    return out


def geodesic_distance(passable: np.ndarray, goal_idx: np.ndarray) -> np.ndarray:
    """Batched Moore-8 BFS distance (unit step cost) to ``goal_idx``; -1 where unreachable.

    passable: [B,H,W] bool.  Returns [B,H,W] int32.
    """
    B, H, W = passable.shape
    dist = np.full((B, H, W), -1, np.int32)
    frontier = np.zeros((B, H * W), bool)
    frontier[np.arange(B), goal_idx] = True
    frontier = frontier.reshape(B, H, W)
    visited = frontier.copy()
    d = 0
    while frontier.any():
        dist[frontier] = d
        frontier = _dilate8(frontier) & passable & ~visited
        visited |= frontier
        d += 1
    return dist


def fixture_block(B: int = 8, H: int = 64, W: int = 64) -> Problems:
    """The reference's only test fixture (tests/astar_test.py:5-14), scaled with H,W:
    all-passable map with a centred obstacle block rows/cols [3/8, 3/4), start (0,0), goal (H-1,W-1)."""
    m = np.ones((B, 1, H, W), np.float32)
    m[:, :, (3 * H) // 8:(3 * H) // 4, (3 * W) // 8:(3 * W) // 4] = 0
    s = np.zeros_like(m)
    s[:, :, 0, 0] = 1
    g = np.zeros_like(m)
    g[:, :, -1, -1] = 1
    return Problems(m, s, g)


def random_obstacle_maps(B: int, H: int = 32, W: Optional[int] = None, p: float = 0.25,
                         seed: int = 1234) -> Problems:
    """i.i.d. obstacles with probability ``p``; start and goal are two distinct
    passable cells of the same 8-connected component (resampled until that holds)."""
    W = H if W is None else W
    rng = _rng(seed)
    maps = np.zeros((B, H, W), bool)
    s_idx = np.zeros(B, np.int64)
    g_idx = np.zeros(B, np.int64)
    todo = np.arange(B)
    while todo.size:
        n = todo.size
        m = rng.random((n, H, W)) > p
        flat = m.reshape(n, -1)
        ok = flat.sum(1) >= 2
        # goal: uniform over passable cells
        u = rng.random((n, H * W)) * flat
        g = u.argmax(1)
        dist = geodesic_distance(m, g)
        reach = (dist.reshape(n, -1) > 0)
        ok &= reach.any(1)
        u2 = rng.random((n, H * W)) * reach
        s = u2.argmax(1)
        good = np.nonzero(ok)[0]
        maps[todo[good]] = m[good]
        s_idx[todo[good]] = s[good]
        g_idx[todo[good]] = g[good]
        todo = todo[~ok]
    return Problems(maps.astype(np.float32)[:, None], _one_hot(s_idx, H, W), _one_hot(g_idx, H, W))


def _carve_maze(rng: np.random.Generator, size: int, braid: float) -> np.ndarray:
    """Randomised depth-first maze on the odd lattice of a size x size grid (1 = passable)."""
    n = (size - 1) // 2  # lattice cells per side
    m = np.zeros((size, size), bool)
    seen = np.zeros((n, n), bool)
    stack = [(int(rng.integers(n)), int(rng.integers(n)))]
    seen[stack[0]] = True
    m[2 * stack[0][0] + 1, 2 * stack[0][1] + 1] = True
    moves = ((0, 1), (1, 0), (0, -1), (-1, 0))
    while stack:
        i, j = stack[-1]
        order = rng.permutation(4)
        for k in order:
            di, dj = moves[k]
            a, b = i + di, j + dj
            if 0 <= a < n and 0 <= b < n and not seen[a, b]:
                seen[a, b] = True
                m[2 * i + 1 + di, 2 * j + 1 + dj] = True
                m[2 * a + 1, 2 * b + 1] = True
                stack.append((a, b))
                break
        else:
            stack.pop()
    if braid > 0:  # knock out some interior walls to create loops
        walls = np.argwhere(~m[1:2 * n, 1:2 * n]) + 1
        sel = walls[((walls.sum(1) % 2) == 1)]  # wall segments between two lattice cells
        pick = sel[rng.random(len(sel)) < braid]
        m[pick[:, 0], pick[:, 1]] = True
    return m


def maze_maps(B: int, size: int = 32, seed: int = 1234, braid: float = 0.1,
              pcts=(0.55, 0.70, 0.85)) -> Problems:
    """maze-like stand-in for ``mazes_032_moore_c8``.

    Goal uniform over passable cells; start drawn from the cells whose true distance to the
    goal lies in a random one of the 55-70 / 70-85 / 85-100 percentile bands, mirroring
    ``MazeDataset.get_random_start_map`` (reference ``utils/data.py:200-221``)."""
    rng = _rng(seed)
    maps = np.stack([_carve_maze(rng, size, braid) for _ in range(B)])
    flat = maps.reshape(B, -1)
    g_idx = (rng.random((B, size * size)) * flat).argmax(1)
    dist = geodesic_distance(maps, g_idx).reshape(B, -1)
    s_idx = np.zeros(B, np.int64)
    th = np.array(list(pcts) + [1.0])
    for b in range(B):
        d = dist[b]
        vals = d[d > 0]
        # reference works on negative distances (opt_dist <= 0); same bands on positive distances
        q = np.percentile(vals, 100.0 * th)
        r = int(rng.integers(0, len(th) - 1))
        cand = np.nonzero((d >= q[r]) & (d <= q[r + 1]) & (d > 0))[0]
        if cand.size == 0:
            cand = np.nonzero(d > 0)[0]
        s_idx[b] = cand[int(rng.integers(cand.size))]
    return Problems(maps.astype(np.float32)[:, None], _one_hot(s_idx, size, size), _one_hot(g_idx, size, size))


def random_costs(B: int, H: int, W: int, seed: int = 4321, lo: float = 0.0, hi: float = 1.0) -> np.ndarray:
    """Encoder-like cost maps: U(lo,hi) fp32 (the reference's encoders emit sigmoid outputs in (0,1))."""
    rng = _rng(seed)
    return (lo + (hi - lo) * rng.random((B, 1, H, W))).astype(np.float32)


# ---- datasets in the reference's on-disk format (utils/data.py:130-150) --------------------------------------------------
# action index -> (dy, dx), the planning-datasets "moore" order that MazeDataset.next_loc decodes (utils/data.py:232-241)
ACTION_MOVES = ((-1, 0), (0, 1), (0, -1), (1, 0), (-1, 1), (-1, -1), (1, 1), (1, -1))


def optimal_policies(passable: np.ndarray, dist: np.ndarray) -> np.ndarray:
    """One-hot optimal action per cell, [N, 8, 1, H, W] float32: the move to the reachable neighbour with the smallest
    distance-to-goal (first action in ``ACTION_MOVES`` order on ties); all-zero on obstacles, unreachable cells and the goal."""
    N, H, W = passable.shape
    big = np.iinfo(np.int32).max
    d = np.where((dist >= 0) & passable, dist, big).astype(np.int64)
    pad = np.full((N, H + 2, W + 2), big, np.int64)
    pad[:, 1:-1, 1:-1] = d
    nb = np.stack([pad[:, 1 + dy:1 + dy + H, 1 + dx:1 + dx + W] for dy, dx in ACTION_MOVES], 1)  # [N,8,H,W]
    best = nb.argmin(1)
    ok = (d < big) & (d > 0) & (nb.min(1) < d)
    pol = np.zeros((N, 8, H, W), np.float32)
    n, y, x = np.nonzero(ok)
    pol[n, best[n, y, x], y, x] = 1.0
    return pol[:, :, None]


def write_maze_npz(path: str, n_train: int = 32, n_valid: int = 8, n_test: int = 8, size: int = 32, seed: int = 7) -> None:
    """A small dataset file in the layout ``MazeDataset._process`` reads: arr_{0,4,8} map_designs [N,W,W], arr_{1,5,9}
    goal_maps [N,1,W,W], arr_{2,6,10} opt_policies [N,8,1,W,W], arr_{3,7,11} opt_dists [N,1,W,W] (NEGATIVE distances to the
    goal, the map's minimum on obstacles / unreachable cells, as in the planning-datasets files)."""
    rng = _rng(seed)
    arrs = []
    for n in (n_train, n_valid, n_test):
        maps = np.stack([_carve_maze(rng, size, 0.1) for _ in range(n)])
        flat = maps.reshape(n, -1)
        g_idx = (rng.random((n, size * size)) * flat).argmax(1)
        dist = geodesic_distance(maps, g_idx)
        od = np.where(dist >= 0, -dist.astype(np.float32), -(dist.max((1, 2), keepdims=True) + 1.0)).astype(np.float32)
        goal = np.zeros((n, size * size), np.float32)
        goal[np.arange(n), g_idx] = 1
        arrs += [maps.astype(np.float32), goal.reshape(n, 1, size, size), optimal_policies(maps, dist), od[:, None]]
    np.savez_compressed(path, *arrs)
