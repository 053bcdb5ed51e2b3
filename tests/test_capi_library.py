"""CPU: the C-ABI shared library builds for gfx950, loads, and exports every symbol include/nastar.h declares.
No compute call is made (there is no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "nastar.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(nastar_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_agree():
    from neural_astar import _native
    assert _declared_symbols() == sorted(_native.EXPORTED_SYMBOLS)


def test_library_builds_loads_and_exports_everything():
    from neural_astar import _native
    if not os.path.exists(_native.LIB_PATH):
        _native.build()
    lib = _native.load()
    for sym in _declared_symbols():
        assert hasattr(lib, sym), sym
    hdr = open(os.path.join(ROOT, "include", "nastar.h")).read()
    assert lib.nastar_version() == int(re.search(r"#define NASTAR_VERSION (\d+)", hdr).group(1)) >= 300  # the library matches the header
    import neural_astar
    major, minor, patch = (int(x) for x in neural_astar.__version__.split("."))
    assert major * 1000 + minor * 100 + patch == lib.nastar_version()  # ... and the package names the same version
    assert lib.nastar_workspace_bytes(4096, 32, 32, 0) == 0
    assert lib.nastar_workspace_bytes(2, 79, 79, 0) == 0  # 9 B/cell compact state in LDS up to 6399 cells ...
    assert lib.nastar_workspace_bytes(2, 80, 80, 0) >= 2 * 80 * 80 * 5  # ... beyond, the large-map kernel (faster from 80x80 on: profiles/r06/probe_mid.jsonl)
    assert lib.nastar_workspace_bytes(2, 128, 128, 0) >= 2 * 128 * 128 * 5
    assert 2 * 200 * 150 * 5 <= lib.nastar_workspace_bytes(2, 200, 150, 0) < 2 * 200 * 150 * 6  # larger maps: 5 B/cell in the workspace (open list in LDS)
    # round 6: the marks of NASTAR_FLAG_MARK_COUPLED (B int32, 16-byte aligned, + the t_end cell) behind the slabs; the probe bitmaps of the finish call behind those
    assert lib.nastar_workspace_bytes(6, 32, 32, 32768) == 32 + 16 and lib.nastar_workspace_bytes(6, 32, 32, 32768 | 256) == 32 + 16 + 16
    assert lib.nastar_batchloop_workspace_bytes(6, 32, 32, 1024) == 32 + 16 + 16 + 6 * 32 * 4
    assert lib.nastar_batchloop_workspace_bytes(2, 200, 150, 22500) == lib.nastar_workspace_bytes(2, 200, 150, 32768 | 256) + 2 * 704 * 4
    assert lib.nastar_completion_supported(32, 32) == 1 and lib.nastar_completion_supported(200, 150) == 0
    assert lib.nastar_host_wait_nonzero(None, 10) == 0
    assert lib.nastar_backward_workspace_bytes(4, 32, 32, 256) >= 4 * 258 * 16  # replay backward: 16 B of history per step
    assert lib.nastar_last_error() == b""


def test_argument_validation_needs_no_gpu():
    """Status codes for bad arguments are produced before any HIP call."""
    from neural_astar import _native
    lib = _native.load()
    one = 16  # any non-NULL pointer value; never dereferenced on these paths
    assert lib.nastar_forward(None, one, one, one, 1, 8, 8, 0.5, 64, one, one, None, one, one, None, 0, 0, None) == _native.NASTAR_ERR_NULL
    assert lib.nastar_forward(one, one, one, one, 0, 8, 8, 0.5, 64, one, one, None, one, one, None, 0, 0, None) == _native.NASTAR_ERR_BAD_SHAPE
    assert lib.nastar_forward(one, one, one, one, 1, 8, 8, 0.5, 0, one, one, None, one, one, None, 0, 0, None) == _native.NASTAR_ERR_BAD_SHAPE
    assert lib.nastar_forward(one, one, one, one, 1, 2048, 2048, 0.5, 64, one, one, None, one, one, None, 0, 0, None) == _native.NASTAR_ERR_UNSUPPORTED
    assert lib.nastar_forward(one, one, one, one, 1, 1024, 1024, 0.5, 64, one, one, None, one, one, None, 0, 0, None) == _native.NASTAR_ERR_NULL  # (round 6: a supported size, it wants its workspace)
    assert lib.nastar_workspace_bytes(1, 1024, 1024, 0) >= 5 * 1024 * 1024 and lib.nastar_workspace_bytes(1, 1024, 1153, 0) == 0
    assert lib.nastar_forward(one, one, one, one, 1, 200, 150, 0.5, 64, one, one, None, one, one, one, 16, 0, None) == _native.NASTAR_ERR_WORKSPACE
    assert lib.nastar_forward(one, one, one, one, 1, 200, 150, 0.5, 64, one, one, None, one, one, None, 0, 0, None) == _native.NASTAR_ERR_NULL
    # placement: argument checks of nastar_forward carry over; maps whose state lives in HBM take none
    assert lib.nastar_forward_ordered(None, one, one, one, 1, 8, 8, 0.5, 64, one, one, None, one, one, None, None, 0, 0, one, one, None) == _native.NASTAR_ERR_NULL
    assert lib.nastar_forward_ordered(one, one, one, one, 0, 8, 8, 0.5, 64, one, one, None, one, one, None, None, 0, 0, one, None, None) == _native.NASTAR_ERR_BAD_SHAPE
    assert lib.nastar_forward_ordered(one, one, one, one, 1, 200, 150, 0.5, 64, one, one, None, one, one, None, one, 1 << 30, 0, one, None, None) == _native.NASTAR_ERR_UNSUPPORTED
    assert lib.nastar_forward_ordered(one, one, one, one, 1, 200, 150, 0.5, 64, one, one, None, one, one, None, one, 1 << 30, 0, None, one, None) == _native.NASTAR_ERR_UNSUPPORTED
    # nastar_forward_ex (0.5.0): the same checks; a CHECKED order needs its 16-byte verdict word in the workspace
    assert lib.nastar_forward_ex(None, one, one, one, 1, 8, 8, 0.5, 64, one, one, None, one, one, None, None, 0, 0, None, None, None, None, None) == _native.NASTAR_ERR_NULL
    assert lib.nastar_forward_ex(one, one, one, one, 1, 200, 150, 0.5, 64, one, one, None, one, one, None, one, 1 << 30, 0, one, None, None, None, None) == _native.NASTAR_ERR_UNSUPPORTED
    assert lib.nastar_workspace_bytes(4096, 32, 32, 256) == 16 and lib.nastar_workspace_bytes(4096, 32, 32, 0) == 0
    assert lib.nastar_forward_ex(one, one, one, one, 4, 32, 32, 0.5, 64, one, one, None, one, one, None, None, 0, 256, one, None, None, None, None) == _native.NASTAR_ERR_NULL
    assert lib.nastar_forward_ex(one, one, one, one, 4, 32, 32, 0.5, 64, one, one, None, one, one, None, one, 8, 256, one, None, None, None, None) == _native.NASTAR_ERR_WORKSPACE
    assert lib.nastar_placement_from_levels(None, 4, one, None) == _native.NASTAR_ERR_NULL
    assert lib.nastar_placement_from_levels(one, 0, one, None) == _native.NASTAR_ERR_BAD_SHAPE
    assert lib.nastar_backward_replay(None, one, one, one, one, one, 1, 8, 8, 0.5, 64, one, None, one, one, 64, 0, None) == _native.NASTAR_ERR_NULL
    # round 6: the A/B switches of earlier rounds (8 = NO_ASM, 16 / 128 = older streams, 32 = NO_DIVE, 512 = the round-4 large-map kernel, 2048.. = hybrid variants)
    # left the product ABI: unknown flag bits are refused (csrc/nastar_dev_flags.h, `make dev`)
    for bit in (8, 16, 32, 128, 512, 2048, 4096, 8192, 16384):
        assert lib.nastar_forward(one, one, one, one, 1, 8, 8, 0.5, 64, one, one, None, one, one, None, 0, bit, None) == _native.NASTAR_ERR_UNSUPPORTED, bit
    assert lib.nastar_forward_batchloop_finish(one, one, one, one, 2, 8, 8, 0.5, 64, one, one, None, one, one, None, 0, None) == _native.NASTAR_ERR_NULL
    assert lib.nastar_forward_batchloop_finish(one, one, one, one, 2, 8, 8, 0.5, 64, one, one, None, one, one, one, 16, None) == _native.NASTAR_ERR_WORKSPACE
    for sym in ("nastar_backward", "nastar_backward_l1", "nastar_has_dev_kernels"):  # rounds 1-3 legacy entry points: gone in 0.4.0
        assert not hasattr(lib, sym), sym
    assert lib.nastar_heuristic(None, 1, 8, 8, one, None) == _native.NASTAR_ERR_NULL
    # training / data-path / encoder entry points
    assert lib.nastar_l1_loss(one, None, 64, one, one, 2048, None) == _native.NASTAR_ERR_NULL
    assert lib.nastar_l1_loss(one, one, 0, one, one, 2048, None) == _native.NASTAR_ERR_BAD_SHAPE
    assert lib.nastar_l1_loss(one, one, 64, one, one, 8, None) == _native.NASTAR_ERR_WORKSPACE
    assert lib.nastar_policy_rollout(one, None, one, 1, 1, 8, 8, 8, one, one, None) == _native.NASTAR_ERR_NULL
    assert lib.nastar_policy_rollout(one, one, one, 1, 1, 9, 8, 8, one, one, None) == _native.NASTAR_ERR_BAD_SHAPE
    import ctypes
    arr = (ctypes.c_void_p * 5)(*([one] * 5))
    assert lib.nastar_encoder_cnn_forward(None, one, one, 1, 1, 32, 32, arr, arr, arr, 1.0, one, one, 1 << 20, None) == _native.NASTAR_ERR_NULL
    assert lib.nastar_encoder_cnn_forward(one, one, one, 1, 1, 30, 32, arr, arr, arr, 1.0, one, one, 1 << 20, None) == _native.NASTAR_ERR_UNSUPPORTED
    assert lib.nastar_encoder_cnn_forward(one, one, one, 1, 1, 32, 32, arr, arr, arr, 1.0, one, one, 16, None) == _native.NASTAR_ERR_WORKSPACE
    assert lib.nastar_encoder_workspace_bytes(4096, 32, 32) == 4096 * 1024 * 800 and lib.nastar_encoder_workspace_bytes(0, 32, 32) == 0
    assert lib.nastar_conv3x3_bf16(one, one, one, one, None, 1, 32, 32, 32, 64, 1, None) == _native.NASTAR_ERR_NULL
    assert lib.nastar_pack_outputs(None, one, 1, 8, 8, one, None) == _native.NASTAR_ERR_NULL
    assert lib.nastar_unpack_outputs(one, 0, 8, 8, one, one, None) == _native.NASTAR_ERR_BAD_SHAPE


def test_product_has_no_cpu_fallback_and_never_touches_the_oracle():
    import torch
    from neural_astar.planner import VanillaAstar
    x = torch.ones(1, 1, 8, 8)
    with pytest.raises(RuntimeError, match="no CPU"):
        VanillaAstar()(x, x, x)
    pkg = os.path.join(ROOT, "neural-astar_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "nastar_oracle" not in src, f
