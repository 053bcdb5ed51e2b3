"""Loader for tests/golden/*.npz (written by oracle/gen_golden.py from the reference itself)."""
from __future__ import annotations

import glob
import os
from typing import NamedTuple, Optional

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Golden(NamedTuple):
    name: str
    B: int
    H: int
    W: int
    g_ratio: float
    Tmax: float
    training: bool
    map_designs: np.ndarray  # [B,1,H,W] f32
    start_maps: np.ndarray
    goal_maps: np.ndarray
    cost_maps: np.ndarray  # [B,1,H,W] f32 (== map_designs for the VanillaAstar convention)
    passable: np.ndarray  # [B,1,H,W] f32
    histories: np.ndarray  # [B,1,H,W] f32 exact 0/1
    paths: np.ndarray  # [B,1,H,W] int64
    grad_up: Optional[np.ndarray]
    grad_cost: Optional[np.ndarray]
    sel_log: Optional[np.ndarray]

    @property
    def max_iters(self) -> int:
        t = self.Tmax if self.training else 1.0
        return int(t * self.W * self.W)  # differentiable_astar.py:200-202


def _unpack(bits, B, H, W):
    return np.unpackbits(bits, axis=1)[:, :H * W].reshape(B, 1, H, W)


def _onehot(idx, B, H, W):
    m = np.zeros((B, H * W), np.float32)
    m[np.arange(B), idx] = 1
    return m.reshape(B, 1, H, W)


def names():
    """search fixtures (ckpt_*.npz are weight fixtures, data_*.npz dataset fixtures: not search cases)"""
    return sorted(n for n in (os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))) if not n.startswith(("ckpt_", "data_")))


def load(name: str) -> Golden:
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    B, H, W = int(z["B"]), int(z["H"]), int(z["W"])
    maps = _unpack(z["map_bits"], B, H, W).astype(np.float32)
    cost = z["cost"].astype(np.float32) if "cost" in z else maps
    passable = _unpack(z["passable_bits"], B, H, W).astype(np.float32) if "passable_bits" in z else maps
    return Golden(
        name, B, H, W, float(z["g_ratio"]), float(z["Tmax"]), bool(z["training"]), maps,
        _onehot(z["start_idx"], B, H, W), _onehot(z["goal_idx"], B, H, W), cost, passable,
        _unpack(z["hist_bits"], B, H, W).astype(np.float32), _unpack(z["path_bits"], B, H, W).astype(np.int64),
        z["grad_up"] if "grad_up" in z else None, z["grad_cost"] if "grad_cost" in z else None,
        z["sel_log"] if "sel_log" in z else None)
