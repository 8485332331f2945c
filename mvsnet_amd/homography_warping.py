"""Host-side mirror of mvsnet/homography_warping.py on top of the HIP library.

Same function names and argument meaning as the reference; tensors are torch device tensors
instead of TensorFlow graph nodes, batch dimension fixed at 1 (squeezed away).
"""
from __future__ import annotations

import torch

from . import _lib


def homography_transforms(cams, depth_num, depth_start, depth_interval=0.0, depth_end=0.0,
                          inverse_depth=False, want_homographies=False):
    """cams (N,2,4,4) device float32 -> transforms (N-1,D,8) [, homographies (N-1,D,3,3)].

    One launch for what the reference builds per view with get_homographies
    (homography_warping.py:10-58) or get_homographies_inv_depth (:60-106) and then per
    (view, plane) with the coefficient algebra of tf_transform_homography (:216-250)."""
    lib = _lib.load()
    cams = _lib.f32(cams, "cams")
    n = cams.shape[0]
    D = int(depth_num)
    T = torch.empty((n - 1, D, 8), device=cams.device, dtype=torch.float32)
    Hm = torch.empty((n - 1, D, 3, 3), device=cams.device, dtype=torch.float32) if want_homographies else None
    _lib.check(lib.mvs_homography_transforms_f32(
        _lib.ptr(cams), n, D, float(depth_start), float(depth_interval), float(depth_end),
        int(bool(inverse_depth)), _lib.ptr(Hm), _lib.ptr(T), _lib.stream_ptr()),
        "mvs_homography_transforms_f32")
    return (T, Hm) if want_homographies else T


def get_homographies(left_cam, right_cam, depth_num, depth_start, depth_interval):
    """mvsnet/homography_warping.py:10-58.  left_cam/right_cam (2,4,4) -> (D,3,3)."""
    cams = torch.stack([left_cam, right_cam]).contiguous()
    return homography_transforms(cams, depth_num, depth_start, depth_interval,
                                 want_homographies=True)[1][0]


def get_homographies_inv_depth(left_cam, right_cam, depth_num, depth_start, depth_end):
    """mvsnet/homography_warping.py:60-106.  -> (D,3,3)."""
    cams = torch.stack([left_cam, right_cam]).contiguous()
    return homography_transforms(cams, depth_num, depth_start, depth_end=depth_end,
                                 inverse_depth=True, want_homographies=True)[1][0]


def homography_to_transform8(homography):
    """Coefficient algebra of tf_transform_homography (homography_warping.py:216-250) on torch
    tensors: (...,3,3) -> (...,8).  Tiny host-side helper for callers that bring their own H."""
    h = homography.reshape(-1, 9)
    a0, a1, a2, b0, b1, b2, c0, c1, c2 = [h[:, i] for i in range(9)]
    a_0 = a0 - c0 / 2
    a_1 = a1 - c1 / 2
    a_2 = (a0 + a1) / 2 + a2 - (c0 + c1) / 4 - c2 / 2
    b_0 = b0 - c0 / 2
    b_1 = b1 - c1 / 2
    b_2 = (b0 + b1) / 2 + b2 - (c0 + c1) / 4 - c2 / 2
    c_2 = c2 + (c0 + c1) / 2
    lin = torch.stack([a_0, a_1, a_2, b_0, b_1, b_2, c0, c1], dim=1) / c_2[:, None]
    return lin.reshape(homography.shape[:-2] + (8,)).contiguous()


def transform_image(input_image, transform8, border="zeros"):
    """tf.contrib.image.transform(image, t8, 'BILINEAR') (homography_warping.py:251-252).
    input_image (H,W,C) device float32, C % 4 == 0."""
    lib = _lib.load()
    img = _lib.f32(input_image, "input_image")
    H, W, C = img.shape
    out = torch.empty_like(img)
    _lib.check(lib.mvs_warp_f32(_lib.ptr(img), _lib.ptr(_lib.f32(transform8)), H, W, C,
                                0 if border == "zeros" else 1, _lib.ptr(out), _lib.stream_ptr()),
               "mvs_warp_f32")
    return out


def tf_transform_homography(input_image, homography, border="zeros"):
    """mvsnet/homography_warping.py:211-253 (the active warp).  homography (3,3)."""
    return transform_image(input_image, homography_to_transform8(homography), border)


def homography_warping(input_image, homography):
    """mvsnet/homography_warping.py:176-210 (dead code in the reference: clamp-to-border taps)."""
    return tf_transform_homography(input_image, homography, border="clamp")
