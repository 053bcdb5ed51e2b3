"""Loader for tests/golden/*.npz (written by oracle/gen_golden.py from the reference itself)."""
from __future__ import annotations

import glob
import os
from typing import NamedTuple, Optional

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NON_SEARCH_PREFIXES = ("ckpt_", "data_", "trainstep_", "trainloop_", "enc_", "tieclass_", "coupled_", "gradnoise_", "wide_")  # weight / dataset / training-step / encoder fixtures; tieclass_* / coupled_* / gradnoise_*: tests/test_tie_class.py, tests/test_coupled_forward.py, tests/test_oracle_golden.py; wide_*: tests/test_large_maps_gpu.py + tests/test_oracle_golden.py


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
    return sorted(n for n in (os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))) if not n.startswith(NON_SEARCH_PREFIXES))


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


# ---- round 3: full-training-step and encoder goldens (oracle/gen_golden_trainstep.py, the reference package run end to end) ----



class StepGolden(NamedTuple):
    """One reference training step (utils/training.py:55-61 through the reference's NeuralAstar): inputs, initial weights and
    everything the step produced."""
    name: str
    B: int
    H: int
    W: int
    Tmax: float
    g_ratio: float
    map_designs: np.ndarray  # [B,C,Hm,Wm] f32 (C = 1 mazes, 3 WarCraft images)
    start_maps: np.ndarray  # [B,1,H,W]
    goal_maps: np.ndarray
    opt_trajs: np.ndarray  # [B,1,H,W] f32 0/1
    init: dict  # planner state dict before the step
    loss: float
    cost: np.ndarray  # [B,1,H,W] f32 -- the encoder's training-mode output
    grad_cost: np.ndarray  # dL/dcost
    histories: np.ndarray
    paths: np.ndarray
    grads: dict  # parameter name -> gradient (empty for the *_tight file)
    after: dict  # buffer name -> value after the step (BatchNorm running statistics, counters)
    sel_margin: np.ndarray  # [B] smallest priority gap best vs runner-up over the search


def load_step(name: str) -> StepGolden:
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    B, H, W = int(z["B"]), int(z["H"]), int(z["W"])
    if "image_u8" in z:  # WarCraft: 96x96 RGB, start top-left, goal bottom-right
        maps = z["image_u8"].astype(np.float32) / np.float32(255.0)
        s = np.zeros((B, 1, H, W), np.float32); g = np.zeros((B, 1, H, W), np.float32)
        s[:, 0, 0, 0] = 1; g[:, 0, -1, -1] = 1
    else:
        maps = _unpack(z["map_bits"], B, H, W).astype(np.float32)
        s, g = _onehot(z["start_idx"], B, H, W), _onehot(z["goal_idx"], B, H, W)
    init = {k[5:]: z[k] for k in z.files if k.startswith("init/")}
    if not init:
        ck = np.load(os.path.join(GOLDEN_DIR, "ckpt_mazes032_cnn.npz"))
        init = {k: ck[k] for k in ck.files}
    return StepGolden(name, B, H, W, float(z["Tmax"]), float(z["g_ratio"]), maps, s, g,
                      _unpack(z["traj_bits"], B, H, W).astype(np.float32), init, float(z["loss"]), z["cost"], z["grad_cost"],
                      _unpack(z["hist_bits"], B, H, W).astype(np.float32), _unpack(z["path_bits"], B, H, W).astype(np.int64),
                      {k[5:]: z[k] for k in z.files if k.startswith("grad/")}, {k[6:]: z[k] for k in z.files if k.startswith("after/")},
                      z["sel_margin"])


STEP_CONFIGS = {
    # constructor arguments of the reference scripts (scripts/train.py:33-39, scripts/train_warcraft.py:33-40)
    "trainstep_maze32": dict(encoder_input="m+", encoder_arch="CNN", encoder_depth=4, Tmax=0.25),
    "trainstep_maze32_tight": dict(encoder_input="m+", encoder_arch="CNN", encoder_depth=4, Tmax=0.25),
    "trainstep_warcraft12": dict(encoder_input="rgb+", encoder_arch="CNNDownSize", encoder_depth=3, const=10.0, learn_obstacles=True, Tmax=0.25),
}

ENC_CONFIGS = {
    "enc_cnndownsize_rgbp_d3_96": dict(encoder_input="rgb+", encoder_arch="CNNDownSize", encoder_depth=3, const=10.0, learn_obstacles=True),
    "enc_cnndownsize_mp_d2_64x32": dict(encoder_input="m+", encoder_arch="CNNDownSize", encoder_depth=2, const=None, learn_obstacles=True),
    "enc_cnn_mp_d3_20x45": dict(encoder_input="m+", encoder_arch="CNN", encoder_depth=3, const=None),
    "enc_cnn_mp_d2_24x24": dict(encoder_input="m+", encoder_arch="CNN", encoder_depth=2, const=2.0),
    "enc_cnn_m_d1_32": dict(encoder_input="m", encoder_arch="CNN", encoder_depth=1, const=None),
}


class EncGolden(NamedTuple):
    name: str
    map_designs: np.ndarray
    start_maps: np.ndarray
    goal_maps: np.ndarray
    init: dict
    cost: np.ndarray
    const: float


def load_enc(name: str) -> EncGolden:
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    m = z["map_designs"]
    B, h, w = m.shape[0], int(z["h"]), int(z["w"])
    return EncGolden(name, m, _onehot(z["start_idx"], B, h, w), _onehot(z["goal_idx"], B, h, w),
                     {k[5:]: z[k] for k in z.files if k.startswith("init/")}, z["cost"], float(z["const"]))
