"""ctypes binding of the C ABI in ``include/nastar.h`` (``lib/libnastar_hip.so``).

This is the ONLY compute path of the package: there is no CPU or PyTorch fallback.  If the shared library is
missing the import of the planner still succeeds (so that CPU-only tooling can inspect the modules) but the
first call fails loudly with instructions to build it.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # neural-astar_amd/
# NASTAR_LIB: development switch -- another build of the same C ABI (make BUILD=build_x OUT=../lib/libnastar_hip_x.so)
LIB_PATH = os.environ.get("NASTAR_LIB") or os.path.join(_PKG_ROOT, "lib", "libnastar_hip.so")
CSRC_DIR = os.path.join(_PKG_ROOT, "csrc")

NASTAR_OK = 0
NASTAR_ERR_BAD_SHAPE = 1
NASTAR_ERR_UNSUPPORTED = 2
NASTAR_ERR_UNSOLVABLE = 3
NASTAR_ERR_HIP = 4
NASTAR_ERR_NULL = 5
NASTAR_ERR_WORKSPACE = 6

_ERR_NAMES = {
    NASTAR_ERR_BAD_SHAPE: "bad shape (B, H, W and max_iters must be positive)",
    NASTAR_ERR_UNSUPPORTED: "map size not supported by the implemented kernels",
    NASTAR_ERR_UNSOLVABLE: "unsolvable map",
    NASTAR_ERR_HIP: "HIP runtime error",
    NASTAR_ERR_NULL: "NULL pointer argument",
    NASTAR_ERR_WORKSPACE: "workspace too small",
}

# every symbol include/nastar.h declares -- tests check the library exports all of them
EXPORTED_SYMBOLS = (
    "nastar_version",
    "nastar_last_error",
    "nastar_workspace_bytes",
    "nastar_forward",
    "nastar_forward_packed",
    "nastar_forward_ordered",
    "nastar_forward_ex",
    "nastar_batchloop_workspace_bytes",
    "nastar_forward_batchloop_finish",
    "nastar_completion_supported",
    "nastar_host_wait_nonzero",
    "nastar_placement_from_levels",
    "nastar_placement_predict",
    "nastar_backward_workspace_bytes",
    "nastar_backward_replay",
    "nastar_backward_replay_ordered",
    "nastar_backward_l1_replay",
    "nastar_l1_loss",
    "nastar_policy_rollout",
    "nastar_heuristic",
    "nastar_debug_occupancy",
    "nastar_pack_outputs",
    "nastar_unpack_outputs",
    "nastar_encoder_workspace_bytes",
    "nastar_encoder_cnn_forward",
    "nastar_encoder_workspace_bytes_f16x3",
    "nastar_encoder_cnn_forward_f16x3",
    "nastar_encoder_workspace_bytes_f16",
    "nastar_encoder_cnn_forward_f16",
    "nastar_conv3x3_bf16",
    "nastar_encoder_downsize_workspace_bytes",
    "nastar_encoder_cnn_downsize_forward",
    "nastar_conv3x3_f16",
    "nastar_conv3x3_img32_f16",
    "nastar_maxpool2x2_f16",
    "nastar_encoder_prep_f16",
    "nastar_conv3x3_wgrad_workspace_bytes",
    "nastar_conv3x3_wgrad_f16",
    "nastar_chan_stats_f16",
    "nastar_chan_stats_workspace_bytes",
    "nastar_absmax_multi_f32",
    "nastar_pack_conv_weights_multi_f16",
    "nastar_rmsprop_multi_f32",
    "nastar_bn1_parts",
    "nastar_bn1_fwd_partial",
    "nastar_bn1_sigmoid_fwd",
    "nastar_bn1_sigmoid_bwd_partial",
    "nastar_bn1_sigmoid_bwd",
    "nastar_chan_stats_f16_ws",
    "nastar_chan_affine_f16",
    "nastar_pack_conv_weight_f16",
    "nastar_bn_coef_fwd",
    "nastar_bn_coef_bwd",
    "nastar_bn_coef_bwd_io",
    "nastar_bn_stats_coef_fwd_f16",
    "nastar_bn_stats_coef_bwd_f16",
    "nastar_grad_seed_f16",
    "nastar_conv3x3_co1_workspace_bytes",
    "nastar_conv3x3_co1_f16",
    "nastar_conv3x3_co1_wgrad_f16",
    "nastar_grad_scale_f32",
    "nastar_bn_stats_coef_bwd_u1_f16",
    "nastar_chan_affine_u1_f16",
    "nastar_chan_stats_u1_f16_ws",
    "nastar_maxpool2x2_bwd_f16",
    "nastar_upcat_f16",
    "nastar_upcat_bwd_f16",
    "nastar_grad_add_f16",
)


class NativeLibraryMissing(RuntimeError):
    pass


_lib: Optional[ctypes.CDLL] = None


def build(verbose: bool = False) -> str:
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", CSRC_DIR], stdout=out)
    return LIB_PATH


DEV_LIB_PATH = os.path.join(_PKG_ROOT, "lib", "libnastar_hip_dev.so")
_dev_lib: Optional[ctypes.CDLL] = None


def _bind_search(lib: ctypes.CDLL) -> None:
    """argument types of the search entry points (shared by the product library and the development build)"""
    vp, ci, cd, cz = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_size_t
    lib.nastar_workspace_bytes.restype = cz
    lib.nastar_workspace_bytes.argtypes = [ci, ci, ci, ci]
    lib.nastar_forward_ex.restype = ci
    lib.nastar_forward_ex.argtypes = [vp, vp, vp, vp, ci, ci, ci, cd, ci, vp, vp, vp, vp, vp, vp, vp, cz, ci, vp, vp, vp, vp, vp]
    lib.nastar_batchloop_workspace_bytes.restype = cz
    lib.nastar_batchloop_workspace_bytes.argtypes = [ci, ci, ci, ci]
    lib.nastar_forward_batchloop_finish.restype = ci
    lib.nastar_forward_batchloop_finish.argtypes = [vp, vp, vp, vp, ci, ci, ci, cd, ci, vp, vp, vp, vp, vp, vp, cz, vp]
    lib.nastar_backward_workspace_bytes.restype = cz
    lib.nastar_backward_workspace_bytes.argtypes = [ci, ci, ci, ci]
    lib.nastar_backward_replay.restype = ci
    lib.nastar_backward_replay.argtypes = [vp, vp, vp, vp, vp, vp, ci, ci, ci, cd, ci, vp, vp, vp, vp, cz, ci, vp]


def load_dev() -> ctypes.CDLL:
    """The DEVELOPMENT build of the search translation unit (``make -C csrc dev``: the A/B switches of csrc/nastar_dev_flags.h -- older
    instruction streams, the compiled step).  Test / probe infrastructure: nothing in the package calls it; ``ops.search_nograd(lib=...)``
    runs a search through it."""
    global _dev_lib
    if _dev_lib is None:
        if not os.path.exists(DEV_LIB_PATH):
            raise NativeLibraryMissing(f"{DEV_LIB_PATH} not found: build it with `make -C {CSRC_DIR} dev`")
        _dev_lib = ctypes.CDLL(DEV_LIB_PATH)
        _bind_search(_dev_lib)
    return _dev_lib


FASTLANE_PATH = os.path.join(_PKG_ROOT, "lib", "_nastar_fastlane.so")
_fastlane = False  # False: not looked for yet; None: absent / switched off


def load_fastlane():
    """(module, address of nastar_forward_ex, address of nastar_placement_from_levels) of the native host lane (csrc/nastar_fastlane.cpp: output allocation, the launch and the poll of
    its completion flag in C++), or None when lib/_nastar_fastlane.so has not been built (`make -C csrc fastlane`) or NASTAR_FASTLANE=0 --
    the Python lane then issues the SAME launch through ctypes (slower on the host, identical on the device)."""
    global _fastlane
    if _fastlane is False:
        _fastlane = None
        if os.environ.get("NASTAR_FASTLANE", "1") != "0" and os.path.exists(FASTLANE_PATH) and not os.environ.get("NASTAR_LIB"):
            import importlib.util
            try:
                spec = importlib.util.spec_from_file_location("_nastar_fastlane", FASTLANE_PATH)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                fn = ctypes.cast(load().nastar_forward_ex, ctypes.c_void_p).value
                sort_fn = ctypes.cast(load().nastar_placement_from_levels, ctypes.c_void_p).value
                _fastlane = (mod, int(fn), int(sort_fn))
            except Exception as e:  # noqa: BLE001 -- an ABI mismatch of the extension must not take the package down
                import warnings
                warnings.warn(f"neural_astar: {FASTLANE_PATH} could not be loaded ({type(e).__name__}: {e}); using the Python host lane", RuntimeWarning)
    return _fastlane


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryMissing(
            f"{LIB_PATH} not found: the MI355X HIP extension is the only compute path of this package "
            f"(no CPU fallback). Build it with `make -C {CSRC_DIR}` or `python __graft_entry__.py build`.")
    lib = ctypes.CDLL(LIB_PATH)
    vp, ci, cd, cz = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_size_t
    _bind_search(lib)
    lib.nastar_version.restype = ci
    lib.nastar_version.argtypes = []
    lib.nastar_last_error.restype = ctypes.c_char_p
    lib.nastar_last_error.argtypes = []
    lib.nastar_workspace_bytes.restype = cz
    lib.nastar_workspace_bytes.argtypes = [ci, ci, ci, ci]
    lib.nastar_forward.restype = ci
    lib.nastar_forward.argtypes = [vp, vp, vp, vp, ci, ci, ci, cd, ci, vp, vp, vp, vp, vp, vp, cz, ci, vp]
    lib.nastar_forward_ordered.restype = ci
    lib.nastar_forward_ordered.argtypes = [vp, vp, vp, vp, ci, ci, ci, cd, ci, vp, vp, vp, vp, vp, vp, vp, cz, ci, vp, vp, vp]
    lib.nastar_forward_ex.restype = ci
    lib.nastar_forward_ex.argtypes = [vp, vp, vp, vp, ci, ci, ci, cd, ci, vp, vp, vp, vp, vp, vp, vp, cz, ci, vp, vp, vp, vp, vp]
    lib.nastar_completion_supported.restype = ci
    lib.nastar_completion_supported.argtypes = [ci, ci]
    lib.nastar_host_wait_nonzero.restype = ci
    lib.nastar_host_wait_nonzero.argtypes = [vp, ci]
    lib.nastar_placement_from_levels.restype = ci
    lib.nastar_placement_from_levels.argtypes = [vp, ci, vp, vp]
    lib.nastar_placement_predict.restype = ci
    lib.nastar_placement_predict.argtypes = [vp, vp, vp, ci, ci, ci, vp, vp, cz, vp]
    lib.nastar_forward_packed.restype = ci
    lib.nastar_forward_packed.argtypes = [vp, vp, vp, vp, ci, ci, ci, cd, ci, vp, vp, vp, vp, vp, vp, vp, cz, ci, vp]
    lib.nastar_backward_workspace_bytes.restype = cz
    lib.nastar_backward_workspace_bytes.argtypes = [ci, ci, ci, ci]
    lib.nastar_backward_replay.restype = ci
    lib.nastar_backward_replay.argtypes = [vp, vp, vp, vp, vp, vp, ci, ci, ci, cd, ci, vp, vp, vp, vp, cz, ci, vp]
    lib.nastar_backward_replay_ordered.restype = ci
    lib.nastar_backward_replay_ordered.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, cd, ci, vp, vp, vp, vp, cz, ci, vp, vp]
    lib.nastar_backward_l1_replay.restype = ci
    lib.nastar_backward_l1_replay.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, cd, ci, vp, vp, vp, vp, cz, vp]
    lib.nastar_l1_loss.restype = ci
    lib.nastar_l1_loss.argtypes = [vp, vp, ctypes.c_longlong, vp, vp, cz, vp]
    lib.nastar_policy_rollout.restype = ci
    lib.nastar_policy_rollout.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, vp, vp, vp]
    lib.nastar_heuristic.restype = ci
    lib.nastar_heuristic.argtypes = [vp, ci, ci, ci, vp, vp]
    lib.nastar_pack_outputs.restype = ci
    lib.nastar_pack_outputs.argtypes = [vp, vp, ci, ci, ci, vp, vp]
    lib.nastar_unpack_outputs.restype = ci
    lib.nastar_unpack_outputs.argtypes = [vp, ci, ci, ci, vp, vp, vp]
    lib.nastar_encoder_workspace_bytes.restype = cz
    lib.nastar_encoder_workspace_bytes.argtypes = [ci, ci, ci]
    lib.nastar_encoder_cnn_forward.restype = ci
    lib.nastar_encoder_cnn_forward.argtypes = [vp, vp, vp, ci, ci, ci, ci, ctypes.POINTER(vp), ctypes.POINTER(vp),
                                               ctypes.POINTER(vp), ctypes.c_float, vp, vp, cz, vp]
    lib.nastar_encoder_workspace_bytes_f16x3.restype = cz
    lib.nastar_encoder_workspace_bytes_f16x3.argtypes = [ci, ci, ci]
    lib.nastar_encoder_cnn_forward_f16x3.restype = ci
    lib.nastar_encoder_cnn_forward_f16x3.argtypes = [vp, vp, vp, ci, ci, ci, ci, vp, ctypes.POINTER(vp), ctypes.POINTER(vp),
                                                     ctypes.POINTER(vp), ctypes.c_float, vp, vp, cz, vp]
    lib.nastar_encoder_workspace_bytes_f16.restype = cz
    lib.nastar_encoder_workspace_bytes_f16.argtypes = [ci, ci, ci]
    lib.nastar_encoder_cnn_forward_f16.restype = ci
    lib.nastar_encoder_cnn_forward_f16.argtypes = lib.nastar_encoder_cnn_forward_f16x3.argtypes
    lib.nastar_encoder_downsize_workspace_bytes.restype = cz
    lib.nastar_encoder_downsize_workspace_bytes.argtypes = [ci, ci, ci, ci, ci]
    lib.nastar_encoder_cnn_downsize_forward.restype = ci
    lib.nastar_encoder_cnn_downsize_forward.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, ctypes.POINTER(vp),
                                                        ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.c_float, vp, vp, cz, vp]
    lib.nastar_conv3x3_bf16.restype = ci
    lib.nastar_conv3x3_bf16.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]
    lib.nastar_conv3x3_f16.restype = ci
    lib.nastar_conv3x3_f16.argtypes = [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ctypes.c_float, vp]
    lib.nastar_conv3x3_img32_f16.restype = ci
    lib.nastar_conv3x3_img32_f16.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, vp]
    lib.nastar_maxpool2x2_f16.restype = ci
    lib.nastar_maxpool2x2_f16.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp]
    lib.nastar_encoder_prep_f16.restype = ci
    lib.nastar_encoder_prep_f16.argtypes = [vp, vp, vp, ci, ctypes.c_longlong, ci, ci, vp, vp]
    lib.nastar_conv3x3_wgrad_f16.restype = ci
    lib.nastar_conv3x3_wgrad_f16.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, ctypes.c_float, vp, vp, cz, vp]
    lib.nastar_conv3x3_wgrad_workspace_bytes.restype = cz
    lib.nastar_conv3x3_wgrad_workspace_bytes.argtypes = [ci, ci, ci, ci, ci]
    lib.nastar_chan_stats_f16.restype = ci
    lib.nastar_chan_stats_f16.argtypes = [vp, vp, vp, vp, vp, vp, ctypes.c_longlong, ci, ci, vp]
    cll = ctypes.c_longlong
    lib.nastar_bn1_parts.restype = ci
    lib.nastar_bn1_parts.argtypes = [cll]
    lib.nastar_bn1_fwd_partial.restype = ci
    lib.nastar_bn1_fwd_partial.argtypes = [vp, cll, vp, vp]
    lib.nastar_bn1_sigmoid_fwd.restype = ci
    lib.nastar_bn1_sigmoid_fwd.argtypes = [vp, cll, vp, ci, cd, vp, vp, cd, vp, cd, vp, vp, vp, vp, vp]
    lib.nastar_bn1_sigmoid_bwd_partial.restype = ci
    lib.nastar_bn1_sigmoid_bwd_partial.argtypes = [vp, vp, cll, vp, vp, vp, vp, vp, vp]
    lib.nastar_bn1_sigmoid_bwd.restype = ci
    lib.nastar_bn1_sigmoid_bwd.argtypes = [vp, vp, cll, vp, vp, vp, vp, vp, ci, cd, vp, vp, vp, vp, vp]
    lib.nastar_absmax_multi_f32.restype = ci
    lib.nastar_absmax_multi_f32.argtypes = [vp, ci, vp, vp]
    lib.nastar_pack_conv_weights_multi_f16.restype = ci
    lib.nastar_pack_conv_weights_multi_f16.argtypes = [vp, ci, ci, ci, vp, vp, vp, vp]
    lib.nastar_rmsprop_multi_f32.restype = ci
    lib.nastar_rmsprop_multi_f32.argtypes = [vp, ci, ctypes.c_float, ctypes.c_float, ctypes.c_float, vp]
    lib.nastar_chan_stats_workspace_bytes.restype = cz
    lib.nastar_chan_stats_workspace_bytes.argtypes = [ctypes.c_longlong, ci]
    lib.nastar_chan_stats_f16_ws.restype = ci
    lib.nastar_chan_stats_f16_ws.argtypes = [vp, vp, vp, vp, vp, vp, ctypes.c_longlong, ci, ci, vp, cz, vp]
    lib.nastar_conv3x3_co1_workspace_bytes.restype = cz
    lib.nastar_conv3x3_co1_workspace_bytes.argtypes = [ci, ci, ci, ci]
    lib.nastar_conv3x3_co1_f16.restype = ci
    lib.nastar_conv3x3_co1_f16.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, vp, vp, vp, vp, cz, vp]
    lib.nastar_conv3x3_co1_wgrad_f16.restype = ci
    lib.nastar_conv3x3_co1_wgrad_f16.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp, vp, vp, vp, cz, vp]
    lib.nastar_grad_scale_f32.restype = ci
    lib.nastar_grad_scale_f32.argtypes = [vp, ctypes.c_longlong, vp, vp, vp]
    lib.nastar_bn_stats_coef_bwd_u1_f16.restype = ci
    lib.nastar_bn_stats_coef_bwd_u1_f16.argtypes = [vp, vp, ci, ci, ci, vp, vp, vp, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, cz, vp]
    lib.nastar_chan_stats_u1_f16_ws.restype = ci
    lib.nastar_chan_stats_u1_f16_ws.argtypes = [vp, vp, vp, ci, ci, ci, vp, vp, vp, vp, vp, ci, ci, vp, cz, vp]
    lib.nastar_chan_affine_u1_f16.restype = ci
    lib.nastar_chan_affine_u1_f16.argtypes = [vp, vp, vp, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, ci, ci, vp]
    lib.nastar_chan_affine_f16.restype = ci
    lib.nastar_chan_affine_f16.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_longlong, ci, ci, ci, vp]
    lib.nastar_pack_conv_weight_f16.restype = ci
    lib.nastar_pack_conv_weight_f16.argtypes = [vp, ci, ci, ci, ci, vp, vp, vp, vp, vp, ci, vp]
    lib.nastar_bn_stats_coef_fwd_f16.restype = ci
    lib.nastar_bn_stats_coef_fwd_f16.argtypes = [vp, ctypes.c_longlong, ci, ci, vp, vp, cd, cd, vp, vp, vp, vp, vp, vp, vp, vp, cz, vp]
    lib.nastar_bn_stats_coef_bwd_f16.restype = ci
    lib.nastar_bn_stats_coef_bwd_f16.argtypes = [vp, vp, vp, vp, ctypes.c_longlong, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, cz, vp]
    lib.nastar_bn_coef_bwd_io.restype = ci
    lib.nastar_bn_coef_bwd_io.argtypes = [vp, vp, vp, vp, vp, ctypes.c_longlong, vp, vp, vp, vp, vp, vp, vp, ci, vp]
    lib.nastar_bn_coef_fwd.restype = ci
    lib.nastar_bn_coef_fwd.argtypes = [vp, vp, vp, cd, ctypes.c_longlong, cd, vp, vp, vp, vp, vp, vp, ci, vp]
    lib.nastar_bn_coef_bwd.restype = ci
    lib.nastar_bn_coef_bwd.argtypes = [vp, vp, vp, vp, vp, ctypes.c_longlong, vp, vp, vp, vp, vp, vp, ci, vp]
    lib.nastar_upcat_f16.restype = ci
    lib.nastar_upcat_f16.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]
    lib.nastar_upcat_bwd_f16.restype = ci
    lib.nastar_upcat_bwd_f16.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]
    lib.nastar_grad_add_f16.restype = ci
    lib.nastar_grad_add_f16.argtypes = [vp, vp, vp, vp, vp, vp, ctypes.c_longlong, ci, ci, vp]
    lib.nastar_maxpool2x2_bwd_f16.restype = ci
    lib.nastar_maxpool2x2_bwd_f16.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, vp]
    lib.nastar_grad_seed_f16.restype = ci
    lib.nastar_grad_seed_f16.argtypes = [vp, ctypes.c_longlong, ci, vp, vp, vp, vp]
    lib.nastar_debug_occupancy.restype = ci
    lib.nastar_debug_occupancy.argtypes = [ci, ci, ctypes.POINTER(ci)]
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc == NASTAR_OK:
        return
    msg = _ERR_NAMES.get(rc, f"error {rc}")
    if rc == NASTAR_ERR_HIP and _lib is not None:
        msg += ": " + _lib.nastar_last_error().decode("utf-8", "replace")
    raise RuntimeError(f"{what} failed: {msg}")
