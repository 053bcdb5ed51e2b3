"""CPU restatement of the reference's data-path functions -- TEST INFRASTRUCTURE ONLY (tests/ and oracle/gen_golden.py).

Follows /root/reference/src/neural_astar/utils/data.py:
  * ``opt_traj``            <- MazeDataset.get_opt_traj   (:171-199)  follow the optimal policy from the start until the goal is
                               reached; marks the start and every intermediate cell, NOT the goal;
  * ``start_candidates``    <- MazeDataset.get_random_start_map (:201-220) WITHOUT its two random draws: the three candidate masks
                               (55-70 / 70-85 / 85-100 percentile bands of the distance-to-goal) the draws choose from.
Pinned against the reference's own methods by oracle/gen_golden.py (tests/golden/data_maze32.npz).
"""
from __future__ import annotations

import numpy as np

ACTION_MOVES = ((-1, 0), (0, 1), (0, -1), (1, 0), (-1, 1), (-1, -1), (1, 1), (1, -1))  # data.py:232-241 (y, x)


def opt_traj(start_map: np.ndarray, goal_map: np.ndarray, opt_policy: np.ndarray) -> np.ndarray:
    """start_map, goal_map [1,H,W] one-hot; opt_policy [A,1,H,W] one-hot actions -> [1,H,W] 0/1 float32 (data.py:171-199)."""
    traj = np.zeros_like(start_map, dtype=np.float32)
    _, H, W = start_map.shape
    cur = tuple(int(v) for v in np.argwhere(start_map[0] != 0)[0])
    goal = tuple(int(v) for v in np.argwhere(goal_map[0] != 0)[0])
    steps = 0
    while cur != goal:
        traj[0, cur[0], cur[1]] = 1.0
        a = int(np.argmax(opt_policy[:, 0, cur[0], cur[1]]))           # :243 argmax of the one-hot action
        nxt = (cur[0] + ACTION_MOVES[a][0], cur[1] + ACTION_MOVES[a][1])
        if traj[0, nxt[0], nxt[1]] != 0.0:                              # :193-195
            raise AssertionError("Revisiting the same position while following the optimal policy")
        cur = nxt
        steps += 1
        if steps > H * W:
            raise AssertionError("policy does not reach the goal")
    return traj


def start_candidates(opt_dist: np.ndarray, pcts=(0.55, 0.70, 0.85, 1.0)) -> np.ndarray:
    """opt_dist [1,H,W] -> bool [3, H*W]: candidate start cells of band r = 0, 1, 2 (data.py:212-216)."""
    v = opt_dist.reshape(-1)
    vals = v[v > v.min()]                                               # :212-213 drop obstacles (the minimum)
    th = np.percentile(vals, 100.0 * (1.0 - np.asarray(pcts)))         # :214 (numpy's linear interpolation)
    return np.stack([(v >= th[r + 1]) & (v <= th[r]) for r in range(len(th) - 1)])  # :216
