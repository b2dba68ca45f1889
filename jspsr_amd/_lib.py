"""ctypes binding of lib/libjspsr_hip.so (C ABI: include/jspsr_hip.h).

The library is loaded lazily and loudly: a missing or stale .so is an error, never a fallback.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("JSPSR_LAB_LIB") or os.path.join(_HERE, "lib", "libjspsr_hip.so")  # JSPSR_LAB_LIB: kernel-lab builds only
CSRC = os.path.join(_HERE, "csrc")
ABI_VERSION = 15

_lock = threading.Lock()
_lib = None

c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_f = ctypes.c_float
c_ll = ctypes.c_longlong

# name -> (restype, argtypes); must list every symbol include/jspsr_hip.h declares
SIGNATURES = {
    "jspsr_abi_version": (c_i, []),
    "jspsr_launch_count": (c_ll, [ctypes.c_char_p]),
    "jspsr_conv_dynamic_queue": (c_i, [c_i]),
    "jspsr_last_error": (ctypes.c_char_p, []),
    "jspsr_prop_forward_f32": (c_i, [c_p, c_p, c_p, c_i, c_p, c_p, c_f, c_p, c_i, c_i, c_i, c_p]),
    "jspsr_prop_backward_workspace_bytes": (ctypes.c_size_t, [c_i, c_i, c_i]),
    "jspsr_prop_backward_f32": (c_i, [c_p] * 4 + [c_i] + [c_p] * 6 + [c_i, c_i, c_i, c_p]),
    "jspsr_prop_backward_fold_f32": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p, c_p]),
    "jspsr_prop_step_forward_f32": (c_i, [c_p, c_p, c_p, c_i, c_p, c_p, c_f, c_i, c_p, c_i, c_i, c_i, c_p]),
    "jspsr_prop_step_backward_workspace_bytes": (ctypes.c_size_t, [c_i, c_i, c_i]),
    "jspsr_prop_step_backward_f32": (c_i, [c_p] * 4 + [c_i, c_p, c_f, c_i, c_i] + [c_p] * 6 + [c_i, c_i, c_i, c_p]),
    "jspsr_prop_head_forward": (c_i, [c_i, c_p, c_p, c_p, c_p, c_f, c_p, c_i, c_i, c_i, c_p]),
    "jspsr_prop_head_backward_workspace_bytes": (ctypes.c_size_t, [c_i, c_i, c_i]),
    "jspsr_prop_head_backward": (c_i, [c_i] + [c_p] * 8 + [c_i, c_i, c_i, c_p]),
    "jspsr_prop_logits_forward_f32": (c_i, [c_p, c_p, c_p, c_p, c_f, c_p, c_i, c_i, c_i, c_p]),
    "jspsr_prop_logits_backward_f32": (c_i, [c_p] * 8 + [c_i, c_i, c_i, c_p]),
    "jspsr_head_ok": (c_i, [c_i] * 5),
    "jspsr_head_forward": (c_i, [c_i, c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    "jspsr_head_backward_workspace_bytes": (ctypes.c_size_t, [c_i, c_i, c_i]),
    "jspsr_head_backward": (c_i, [c_i, c_p, c_p, c_i, c_p, c_i, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    "jspsr_pack_weight": (c_i, [c_i, c_p, c_p] + [c_i] * 6 + [c_p]),
    "jspsr_pack_chunk": (c_i, []),
    "jspsr_pack_weights_multi": (c_i, [c_p, c_i, c_ll, c_p]),
    "jspsr_conv2d_stats_rows": (c_i, [c_i, c_i, c_i]),
    "jspsr_conv2d_in_affine_ok": (c_i, [c_i] * 5),
    "jspsr_conv2d_forward": (c_i, [c_i] + [c_p] * 4 + [c_i] * 14 + [c_p, c_p, c_p, c_i, c_p, c_i, c_p]),
    "jspsr_bn_fold": (c_i, [c_p, c_p, c_p, c_p, c_f, c_f, c_i, c_p, c_p, c_p]),
    "jspsr_conv2d_dgrad_reduce_ok": (c_i, [c_i] * 10),
    "jspsr_conv2d_dgrad": (c_i, [c_i] + [c_p] * 4 + [c_i] * 16 + [c_p, c_i, c_p, c_p, c_i, c_p, c_p, c_p]),
    "jspsr_conv2d_wgrad_workspace_bytes": (ctypes.c_size_t, [c_i] * 8),
    "jspsr_conv2d_wgrad_x_affine_ok": (c_i, [c_i] * 10),
    "jspsr_conv2d_wgrad": (c_i, [c_i, c_p, c_i, c_i, c_i, c_p, c_i, c_i, c_i, c_p] + [c_i] * 12 + [c_p, c_i, c_p, c_p]),
    "jspsr_reduce_workspace_bytes": (ctypes.c_size_t, [c_i, c_i, c_i]),
    "jspsr_bn_forward": (c_i, [c_i, c_p, c_i, c_i, c_p, c_i, c_i, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_f, c_f, c_i,
                               c_i, c_f, c_p, c_p, c_ll, c_i, c_p, c_i, c_p, c_p, c_p, c_p]),
    "jspsr_bn_reduce_params": (c_i, [c_p, c_p, c_p, c_p, c_i, c_p, c_p]),
    "jspsr_bn_backward": (c_i, [c_i, c_p, c_i, c_i, c_p, c_i, c_i, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_i, c_i, c_f,
                                c_p, c_p, c_p, c_p, c_i, c_ll, c_i, c_p, c_p, c_i, c_p]),
    "jspsr_act_backward": (c_i, [c_i, c_p, c_i, c_i, c_p, c_i, c_i, c_p, c_i, c_p, c_ll, c_i, c_p, c_p]),
    "jspsr_gate_pool": (c_i, [c_i, c_p, c_i, c_ll, c_i, c_p, c_p, c_p, c_p, c_p]),
    "jspsr_gate_scale": (c_i, [c_i, c_p, c_p, c_p, c_i, c_ll, c_i, c_p]),
    "jspsr_gate_backward_reduce": (c_i, [c_i, c_p, c_p, c_p, c_i, c_ll, c_i, c_p, c_p]),
    "jspsr_gate_backward_apply": (c_i, [c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_ll, c_i, c_p]),
    "jspsr_loss_workspace_bytes": (ctypes.c_size_t, [c_i, c_i, c_i]),
    "jspsr_loss_forward": (c_i, [c_p, c_p, c_f, c_f, c_f, c_p, c_p, c_i, c_i, c_i, c_p]),
    "jspsr_loss_backward": (c_i, [c_p, c_p, c_p, c_f, c_f, c_f, c_p, c_p, c_i, c_i, c_i, c_p]),
    "jspsr_metrics_workspace_bytes": (ctypes.c_size_t, [c_i, c_i]),
    "jspsr_metrics_forward": (c_i, [c_p, c_p, c_i, c_i, c_f, c_f, c_f, c_i, c_p, c_p, c_p]),
    "jspsr_adamw_step": (c_i, [c_p, c_p, c_p, c_p, c_ll, c_f, c_f, c_f, c_f, c_f, c_i, c_p]),
    "jspsr_adamw_step_dev": (c_i, [c_p, c_p, c_p, c_p, c_ll, c_p, c_p]),
    "jspsr_tiles_crop_f32": (c_i, [c_p, c_p] + [c_i] * 6 + [c_p]),
    "jspsr_tiles_merge_f32": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "jspsr_mirror_pad_f32": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "jspsr_elev_scale_f32": (c_i, [c_p, c_p, c_ll, c_i, c_i, ctypes.c_double, ctypes.c_double, ctypes.c_double, c_p]),
    "jspsr_gate_mlp_forward": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p]),
    "jspsr_gate_mlp_backward_workspace_bytes": (ctypes.c_size_t, [c_i, c_i, c_i]),
    "jspsr_gate_mlp_backward": (c_i, [c_p] * 7 + [c_i, c_i, c_i] + [c_p] * 6),
    "jspsr_nchw_to_nhwc": (c_i, [c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
}


class JspsrHipError(RuntimeError):
    pass


def build(verbose: bool = False) -> str:
    """Compile every HIP source under csrc/ for gfx950 into lib/libjspsr_hip.so (hipcc)."""
    cmd = ["make", "-C", CSRC, "-j4"] + ([] if verbose else ["-s"])
    subprocess.check_call(cmd)
    return SO_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(SO_PATH):
            raise JspsrHipError(
                f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). jspsr_amd has no fallback path."
            )
        # torch first: its bundled HIP runtime must be the one this process initialises -- libjspsr_hip.so binds to whatever
        # libamdhip64 is loaded already, and loaded BEFORE torch it would pull in the system copy, leaving the process with
        # two runtimes (seen as "no ROCm-capable device" on the first launch)
        import torch  # noqa: F401
        lib = ctypes.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the .so is stale
            fn.restype = res
            fn.argtypes = args
        v = lib.jspsr_abi_version()
        if v != ABI_VERSION:
            raise JspsrHipError(f"libjspsr_hip.so ABI {v} != binding ABI {ABI_VERSION}: rebuild")
        _lib = lib
    return _lib


calls = 0   # C-ABI calls checked so far (bench.py reports calls per step; an entry point launches 1-3 kernels)


def check(code: int, what: str):
    global calls
    calls += 1
    if code != 0:
        msg = load().jspsr_last_error().decode(errors="replace")
        raise JspsrHipError(f"{what} failed (code {code}): {msg}")
