"""ctypes binding of libmvsnet_hip.so (the C ABI declared in include/mvsnet_hip.h).

There is no CPU fallback: if the shared library is missing or a call returns non-zero this
module raises.  torch is used only to own device memory and the current stream.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MVS_LIB_PATH") or os.path.join(_HERE, "libmvsnet_hip.so")      # MVS_LIB_PATH: A/B builds of the library

_f = C.c_float
_i = C.c_int
_p = C.c_void_p
_sz = C.c_size_t
_pp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); must list every symbol include/mvsnet_hip.h declares
SIGNATURES = {
    "mvs_abi_version": (_i, []),
    "mvs_error_string": (C.c_char_p, [_i]),
    "mvs_set_conv_impl": (_i, [_i]),
    "mvs_get_conv_impl": (_i, []),
    "mvs_homography_transforms_f32": (_i, [_p, _i, _i, _f, _f, _f, _i, _p, _p, _p]),
    "mvs_cost_volume_f32": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p]),
    "mvs_warp_f32": (_i, [_p, _p, _i, _i, _i, _i, _p, _p]),
    "mvs_conv3d_f32": (_i, [_p] * 7 + [_i] * 6 + [_p, _p, _p]),
    "mvs_deconv3d_f32": (_i, [_p] * 7 + [_i] * 5 + [_p, _p, _p]),
    "mvs_conv3d_pair_f32": (_i, [_p] * 3 + [_i] * 6 + [_p] * 5),
    "mvs_gn_stat_slots": (_i, []),
    "mvs_conv2d_prepared_floats": (_sz, [_i, _i, _i, _i]),
    "mvs_conv2d_prepare_f32": (_i, [_p, _i, _i, _i, _i, _p, _p]),
    "mvs_conv2d_prepare_dgrad_f32": (_i, [_p, _i, _i, _i, _p, _p]),
    "mvs_conv2d_gn_f32": (_i, [_p, _p, _p, _p, _i, _i, _p, _p, _p, _p, _i, _i, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p]),
    "mvs_deconv2d_prepared_floats": (_sz, [_i, _i]),
    "mvs_deconv2d_prepare_f32": (_i, [_p, _i, _i, _p, _p]),
    "mvs_deconv2d_gn_f32": (_i, [_p, _p, _p, _p, _i, _i, _p, _p, _i, _i, _i, _i, _p, _p, _p]),
    "mvs_profile_dominant": (_i, [_i]),
    "mvs_profile_dominant_ms": (_i, [_p, _p]),
    "mvs_profile_stages": (_i, [_i]),
    "mvs_profile_stages_ms": (_i, [_p, _p]),
    "mvs_profile_layers": (_i, [_i]),
    "mvs_profile_layers_ms": (_i, [_p, _p]),
    "mvs_bn_finalize_f32": (_i, [_p, _i, C.c_double, _p, _p, _f, _p, _p, _p]),
    "mvs_zero_f64": (_i, [_p, _sz, _p]),
    "mvs_regnet_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "mvs_regnet_us0_f32": (_i, [_p, _i, _i, _i, _i, _i, _pp, _pp, _pp, _f, _p, _sz, _p, _p]),
    "mvs_regnet_prepared_floats": (_sz, [_i, _i]),
    "mvs_regnet_prepare_f32": (_i, [_pp, _i, _i, _p, _p]),
    "mvs_depth_from_features_f32": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _f, _f, _f, _i, _i, _pp, _p, _pp, _pp, _f,
                                         _p, _p, _p, _sz, _p, _p, _p, _p]),
    "mvs_regnet_us0_batch_f32": (_i, [_p, _i, _i, _i, _i, _i, _i, _pp, _p, _pp, _pp, _f, _p, _sz, _p, _p]),
    "mvs_regnet_us0_prepared_f32": (_i, [_p, _i, _i, _i, _i, _i, _pp, _p, _pp, _pp, _f, _p, _sz, _p, _p]),
    "mvs_softargmin_prob_f32": (_i, [_p, _i, _i, _i, _f, _f, _i, _p, _p, _p]),
    "mvs_conv2d_cat_f32": (_i, [_p, _i, _p, _i, _p, _p, _i, _i, _i, _p, _p, _i, _p]),
    "mvs_gru_gates_f32": (_i, [_p] * 7 + [_i, _i, _i, _p, _p, _p]),
    "mvs_gru_blend_f32": (_i, [_p] * 5 + [_i, _i, _i, _p, _p]),
    "mvs_wta_update_f32": (_i, [_p, _f, _i, _i, _p, _p, _p, _p]),
    "mvs_wta_finish_f32": (_i, [_p, _p, _i, _i, _p, _p]),
    "mvs_gru_workspace_bytes": (_sz, [_i] * 6),
    "mvs_softargmin_bwd_f32": (_i, [_p, _p, _p, _i, _i, _i, _f, _f, _i, _p, _p]),
    "mvs_bn_relu_f32": (_i, [_p] * 6 + [_sz, _i, _p, _p]),
    "mvs_bn_bwd_sum_slots": (_i, []),
    "mvs_bn_bwd_reduce_f32": (_i, [_p, _p, C.c_double, _f, _p, _p, _p, _p, _sz, _i, _p, _p]),
    "mvs_bn_bwd_apply_f32": (_i, [_p, _p, C.c_double, _f, _p, _p, _p, _p, _p, _p, _sz, _i, _p, _p, _p, _p]),
    "mvs_conv3d_wgrad_workspace_bytes": (_sz, [_i] * 6),
    "mvs_conv3d_wgrad_f32": (_i, [_p, _p] + [_i] * 6 + [_p, _sz, _p, _p]),
    "mvs_cost_volume_bwd_f32": (_i, [_p, _p, _p] + [_i] * 5 + [_p, _p, _p, _p, _p]),
    "mvs_cost_volume_bwd_workspace_bytes": (_sz, [_i] * 5),
    "mvs_cost_volume_bwd_gather_f32": (_i, [_p, _p, _p] + [_i] * 5 + [_p, _p, _p, _sz, _p, _p, _p]),
    "mvs_rmsprop_step_f32": (_i, [_p, _p, _p, _p, _sz, _f, _f, _f, _f, _f, _p]),
    "mvs_gn_stats_f32": (_i, [_p, _i, _sz, _i, _p, _p]),
    "mvs_unet_prepare_many_f32": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "mvs_transpose_add_many_f32": (_i, [_i, _p, _p, _p, _p]),
    "mvs_add_f64_many_f32": (_i, [_i, _p, _p, _p, _p]),
    "mvs_add_many_f32": (_i, [_i, _p, _p, _p, _p]),
    "mvs_center_images_workspace_bytes": (_sz, [_i]),
    "mvs_center_images_u8_f32": (_i, [_p, _i, _i, _i, _p, _p, _p]),
    "mvs_gn_slots_to_channel_sums_many_f64": (_i, [_i, _p, _p, _p, _i, _i, _p, _p, _p]),
    "mvs_gn_bwd_sums_doubles": (_sz, [_i, _i]),
    "mvs_gn_slots_to_channel_sums_f64": (_i, [_p, _i, _i, _i, _p, _p]),
    "mvs_gn_apply_f32": (_i, [_p, _p, _p, _p, _f, _i, _i, _sz, _i, _p, _p]),
    "mvs_gn_bwd_reduce_f32": (_i, [_p, _p, _p, _p, _f, _i, _p, _i, _sz, _i, _p, _p]),
    "mvs_gn_bwd_apply_f32": (_i, [_p, _p, _p, _p, _f, _i, _p, _p, _i, _sz, _i, _p, _p]),
    "mvs_gn_bwd_apply_tot_f32": (_i, [_p, _p, _p, _p, _f, _i, _p, _p, _p, _i, _sz, _i, _p, _p]),
    "mvs_gn_bwd_sum_slots": (_i, []),
    "mvs_momentum_step_f32": (_i, [_p, _p, _p, _sz, _f, _f, _f, _p]),
    "mvs_adam_step_f32": (_i, [_p, _p, _p, _p, _sz, _f, _f, _f, _f, _f, _p]),
    "mvs_gru_train_slots": (_i, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mvs_gru_train_cell_fwd_f32": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p]),
    "mvs_gru_train_cell_bwd_f32": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p]),
    "mvs_conv2d_wgrad_workspace_bytes": (_sz, [_i] * 5),
    "mvs_conv2d_wgrad_f32": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _sz, _p, _p]),
    "mvs_gru_wta_f32": (_i, [_p, _p, _p] + [_i] * 8 + [_pp, C.POINTER(C.c_float), _p, _sz, _p, _p, _p]),
    "mvs_gru_wta_batch_f32": (_i, [_pp, _pp, _pp] + [_i] * 9 + [_pp, C.POINTER(C.c_float), _p, _sz, _p, _p, _p]),
    "mvs_gru_set_formulation": (_i, [_i]),
    "mvs_gru_fused_route": (_i, [_i, _i, _i, _i, _sz]),
    "mvs_set_test_hook": (_i, [_i, _i]),
    "mvs_get_test_hook": (_i, [_i]),
    "mvs_gru_fused_trace": (_i, [_p, _i]),
    "mvs_gru_prepare": (_i, [_p]),
    "mvs_gru_release": (_i, [_p]),
    "mvs_gru_stream_layout": (_i, [_p, C.POINTER(C.c_int), C.POINTER(C.c_float)]),
    "mvs_regnet_filler_shares": (_i, [C.POINTER(C.c_int)]),
}

CONV_IMPL = {"auto": 0, "scalar": 1, "mfma": 2, "bf16x3": 3}
# ids of mvs_set_test_hook (include/mvsnet_hip.h MVS_HOOK_*): the library reads nothing from the environment
HOOKS = {"cv_tile_rows_log2": 0, "conv_no_span": 1, "conv_no_fuse2": 2, "s2_planes": 3, "gru_one_stream": 4,
         "gru_producer_threads": 5, "unet_persistent": 6, "unet_grid": 7, "fuse2_planes": 8, "regnet_side_branch": 9}
HOOK_DEFAULTS = {"cv_tile_rows_log2": -1, "conv_no_span": 0, "conv_no_fuse2": 0, "s2_planes": 0, "gru_one_stream": 0,
                 "gru_producer_threads": 128, "unet_persistent": 1, "unet_grid": 0, "fuse2_planes": 0, "regnet_side_branch": 0}

_lib = None


class MvsnetHipError(RuntimeError):
    pass


def load():
    """Loads (once) and returns the ctypes handle; raises if the HIP library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MvsnetHipError(
                "libmvsnet_hip.so not found at %s -- build it with `python -m mvsnet_amd.build` "
                "(there is no CPU fallback)" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)      # AttributeError if a declared symbol is missing
            fn.restype = res
            fn.argtypes = args
        if lib.mvs_abi_version() != 1:
            raise MvsnetHipError("libmvsnet_hip.so ABI version mismatch")
        _lib = lib
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().mvs_error_string(rc).decode()
        raise MvsnetHipError("%s failed: %s (code %d)" % (what or "mvsnet_hip call", msg, rc))


def ptr(t):
    """Device pointer of a contiguous float32/float64 CUDA(HIP) tensor, or NULL for None."""
    if t is None:
        return None
    if not t.is_cuda:
        raise MvsnetHipError("expected a device tensor (no CPU path exists)")
    if not t.is_contiguous():
        raise MvsnetHipError("expected a contiguous tensor")
    # Kernels launch on HIP's CURRENT device and on torch's current stream of that device
    # (stream_ptr): a tensor living on another GPU would be dereferenced across devices (a memory
    # fault without peer access).  One process drives one GPU: shard.bind_device() selects it.
    cur = torch.cuda.current_device()
    if t.device.index != cur:
        raise MvsnetHipError("tensor is on cuda:%s but the current device is cuda:%d -- call "
                             "torch.cuda.set_device (mvsnet_amd.shard.bind_device) before any library call"
                             % (t.device.index, cur))
    return C.c_void_p(t.data_ptr())


def f32(t, name="tensor"):
    if t.dtype != torch.float32:
        raise MvsnetHipError("%s must be float32, got %s" % (name, t.dtype))
    return t


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


import threading

_GRU_LOCK = threading.Lock()   # the table below is touched from garbage-collection finalizers as well
_GRU_PREPARED = {}             # (device index, stream handle) -> [generation, live users (DepthPlan objects)] of the set mvs_gru_prepare made
_GRU_NO_SLOT = set()           # (device, stream) keys for which mvs_gru_prepare answered MVS_E_NO_SLOT: not asked again until a set is released
_GRU_GENERATION = [0]
MVS_E_NO_SLOT = -5


def gru_prepare():
    """mvs_gru_prepare for torch's CURRENT stream of the current device, once per (device, stream): the only call of the
    recurrent path that creates streams / events and synchronises (include/mvsnet_hip.h).  Skipped under hipGraph capture:
    a captured sweep runs on the capture stream alone and needs no set.  Returns the token to hand to gru_unref() (or None): the
    key plus the GENERATION of the set, so that a stale token -- its set was released with gru_release() and the stream
    prepared again by somebody else -- can no longer give the new set away.  When all 16 sets of the process are taken
    (MVS_E_NO_SLOT) it warns once per (device, stream), remembers the answer and carries on: the sweep runs without a set."""
    key = (torch.cuda.current_device(), int(torch.cuda.current_stream().cuda_stream))
    if torch.cuda.is_current_stream_capturing():
        return None
    with _GRU_LOCK:
        if key in _GRU_NO_SLOT:
            return None
        if key not in _GRU_PREPARED:
            rc = load().mvs_gru_prepare(stream_ptr())
            if rc == MVS_E_NO_SLOT:
                _GRU_NO_SLOT.add(key)
                import warnings
                warnings.warn("mvsnet_amd: all 16 stream sets of mvs_gru_prepare are in use; the recurrent sweep of this stream runs on the "
                              "stream alone (drop DepthPlan objects of streams you no longer use, or call _lib.gru_release())", RuntimeWarning)
                return None
            check(rc, "mvs_gru_prepare")
            _GRU_GENERATION[0] += 1
            _GRU_PREPARED[key] = [_GRU_GENERATION[0], 0]
        _GRU_PREPARED[key][1] += 1
        return key + (_GRU_PREPARED[key][0],)


def gru_unref(token):
    """One user of the (device, stream) set less; the last one releases it (mvs_gru_release waits for the side streams).  Called by
    DepthPlan.close() / its finalizer: a set must not outlive the stream it was calibrated for (a later stream may receive the
    same handle on another hardware queue).  Tokens of an earlier generation of the key are ignored."""
    if token is None:
        return
    key, gen = token[:2], token[2]
    with _GRU_LOCK:
        ent = _GRU_PREPARED.get(key)
        if ent is None or ent[0] != gen:
            return
        ent[1] -= 1
        if ent[1] > 0:
            return
        del _GRU_PREPARED[key]
        _GRU_NO_SLOT.clear()                       # a slot is free again: streams that were refused may ask once more
        try:
            with torch.cuda.device(key[0]):
                load().mvs_gru_release(C.c_void_p(key[1]))
        except Exception:       # interpreter shutdown: the library or torch may be gone
            pass


def gru_release():
    """mvs_gru_release for torch's current stream, whatever its user count (before the stream object is dropped); tokens handed
    out for this set become stale."""
    key = (torch.cuda.current_device(), int(torch.cuda.current_stream().cuda_stream))
    with _GRU_LOCK:
        if key in _GRU_PREPARED:
            del _GRU_PREPARED[key]
            _GRU_NO_SLOT.clear()
            check(load().mvs_gru_release(stream_ptr()), "mvs_gru_release")


def ptr_array(tensors):
    arr = (C.c_void_p * len(tensors))()
    for k, t in enumerate(tensors):
        arr[k] = ptr(t).value if t is not None else None
    return arr


def set_conv_impl(name: str):
    check(load().mvs_set_conv_impl(CONV_IMPL[name]), "mvs_set_conv_impl")


def set_test_hook(name: str, value: int):
    """mvs_set_test_hook by name (tests and measurement tools only; HOOKS)."""
    check(load().mvs_set_test_hook(HOOKS[name], int(value)), "mvs_set_test_hook(%s, %d)" % (name, value))


class test_hooks:
    """Context manager: sets hooks by name, restores the defaults on exit.  ``with _lib.test_hooks(conv_no_span=1): ...``"""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        for k, v in self.kw.items():
            set_test_hook(k, v)
        return self

    def __exit__(self, *exc):
        for k in self.kw:
            set_test_hook(k, HOOK_DEFAULTS[k])
        return False
