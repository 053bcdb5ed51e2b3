"""ctypes front-end of the CPU oracle (oracle/nastar_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
``cpu_baseline`` leg.  The product package (neural-astar_amd/) never imports this module.

All functions take/return numpy arrays shaped like the reference's tensors with the channel
dimension dropped: maps are ``[B,H,W] float32``; ``histories [B,H,W] float32``;
``paths [B,H,W] int64``.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import NamedTuple, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libnastar_oracle.so")
_lib = None

OK = 0
ERR_UNSOLVABLE = 3


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (a few hundred ms)."""
    src = os.path.join(_HERE, "nastar_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libnastar_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        fp = ctypes.POINTER(ctypes.c_float)
        i64p = ctypes.POINTER(ctypes.c_int64)
        i32p = ctypes.POINTER(ctypes.c_int32)
        ci, cd = ctypes.c_int, ctypes.c_double
        _lib.nastar_oracle_forward_dense.argtypes = [fp, fp, fp, fp, ci, ci, ci, cd, ci, fp, i64p, i32p, i32p, i32p]
        _lib.nastar_oracle_forward_dense.restype = ci
        _lib.nastar_oracle_backward_dense.argtypes = [fp, fp, fp, fp, fp, ci, ci, ci, cd, ci, fp]
        _lib.nastar_oracle_backward_dense.restype = ci
        _lib.nastar_oracle_forward_sm.argtypes = [fp, fp, fp, fp, ci, ci, ci, cd, ci, fp, i64p, i32p, i32p, i32p]
        _lib.nastar_oracle_forward_sm.restype = ci
        _lib.nastar_oracle_heuristic.argtypes = [ci, ci, ci, ci, fp]
        _lib.nastar_oracle_heuristic.restype = None
    return _lib


class OracleOutput(NamedTuple):
    histories: np.ndarray  # [B,H,W] float32 (exact 0/1)
    paths: np.ndarray  # [B,H,W] int64
    sel_log: Optional[np.ndarray]  # [B,max_iters] int32, -1 = not executed
    iters: np.ndarray  # [B] int32
    t_batch: int  # last executed loop index of the whole batch (dense) / max(iters)-1 (sm)
    status: int


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim == 4:
        assert a.shape[1] == 1
        a = a[:, 0]
    assert a.ndim == 3
    return np.ascontiguousarray(a)


def _p(a, ty):
    return a.ctypes.data_as(ctypes.POINTER(ty))


def max_iters_for(W: int, Tmax: float = 1.0, training: bool = False) -> int:
    """differentiable_astar.py:200-202 -- int(Tmax_eff * W * W), Tmax_eff = 1.0 in eval mode."""
    t = Tmax if training else 1.0
    return int(t * W * W)


def forward(cost, start, goal, passable, g_ratio: float = 0.5, max_iters: Optional[int] = None,
            mode: str = "dense", want_log: bool = False) -> OracleOutput:
    lib = _load()
    cost, start, goal, passable = map(_f32, (cost, start, goal, passable))
    B, H, W = cost.shape
    if max_iters is None:
        max_iters = W * W
    hist = np.empty((B, H, W), np.float32)
    paths = np.empty((B, H, W), np.int64)
    log = np.empty((B, max_iters), np.int32) if want_log else None
    iters = np.empty((B,), np.int32)
    logp = _p(log, ctypes.c_int32) if want_log else None
    if mode == "dense":
        tb = ctypes.c_int32(0)
        rc = lib.nastar_oracle_forward_dense(
            _p(cost, ctypes.c_float), _p(start, ctypes.c_float), _p(goal, ctypes.c_float),
            _p(passable, ctypes.c_float), B, H, W, float(g_ratio), int(max_iters),
            _p(hist, ctypes.c_float), _p(paths, ctypes.c_int64), logp, _p(iters, ctypes.c_int32),
            ctypes.byref(tb))
        t_batch = int(tb.value)
    elif mode == "sm":
        status = np.empty((B,), np.int32)
        rc = lib.nastar_oracle_forward_sm(
            _p(cost, ctypes.c_float), _p(start, ctypes.c_float), _p(goal, ctypes.c_float),
            _p(passable, ctypes.c_float), B, H, W, float(g_ratio), int(max_iters),
            _p(hist, ctypes.c_float), _p(paths, ctypes.c_int64), logp, _p(iters, ctypes.c_int32),
            _p(status, ctypes.c_int32))
        t_batch = int(iters.max()) - 1
    else:
        raise ValueError(mode)
    return OracleOutput(hist, paths, log, iters, t_batch, rc)


def backward(grad_hist, cost, start, goal, passable, g_ratio: float = 0.5,
             max_iters: Optional[int] = None) -> np.ndarray:
    lib = _load()
    grad_hist, cost, start, goal, passable = map(_f32, (grad_hist, cost, start, goal, passable))
    B, H, W = cost.shape
    if max_iters is None:
        max_iters = W * W
    out = np.zeros((B, H, W), np.float32)
    rc = lib.nastar_oracle_backward_dense(
        _p(grad_hist, ctypes.c_float), _p(cost, ctypes.c_float), _p(start, ctypes.c_float),
        _p(goal, ctypes.c_float), _p(passable, ctypes.c_float), B, H, W, float(g_ratio),
        int(max_iters), _p(out, ctypes.c_float))
    if rc:
        raise RuntimeError(f"oracle backward failed rc={rc}")
    return out


def heuristic(H: int, W: int, goal_r: int, goal_c: int) -> np.ndarray:
    lib = _load()
    out = np.empty((H, W), np.float32)
    lib.nastar_oracle_heuristic(H, W, goal_r, goal_c, _p(out, ctypes.c_float))
    return out
