"""Prediction glue mirroring mvsnet/predictlib.py: regulariser dispatch and the on-disk outputs.

The reference reads its settings from global tf.app.flags (predictlib.py:21-26, inference.py:19-78);
here they live in an explicit `InferenceConfig`.
"""
from __future__ import annotations

import os
import time
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import preprocess as pp


@dataclass
class InferenceConfig:
    """The flags of mvsnet/inference.py:19-78 that matter for this path (same names/defaults)."""
    input_dir: Optional[str] = None
    output_dir: Optional[str] = None
    view_num: int = 8
    max_d: int = 256
    width: int = 1024
    height: int = 768
    sample_scale: float = 0.25
    interval_scale: float = 1.0
    base_image_size: int = 8
    batch_size: int = 1
    regularization: str = "3DCNN"
    inverse_depth: bool = False
    network_mode: str = "normal"
    refinement: bool = False
    refinement_network: str = "original"              # 'original' | 'unet' (inference.py:57-59)
    upsample_before_refinement: bool = False
    refine_with_confidence: bool = False
    visualize: bool = False
    max_clusters_per_session: Optional[int] = None


def setup_output_dir(input_dir, output_dir):
    """predictlib.py:59-66: default <input_dir>/depths_mvsnet."""
    if output_dir is None:
        output_dir = os.path.join(input_dir, "depths_mvsnet")
    os.makedirs(output_dir, exist_ok=True)
    return output_dir


def get_depth_and_prob_map(full_images, scaled_cams, depth_start, depth_interval, config, weights,
                           depth_num=None, depth_end=None, features=None, ref_image=None):
    """predictlib.py:79-99.  Returns (depth_map, prob_map, None).  The reference's GRU branch
    raises NameError as shipped (undefined depth_num / depth_end, predictlib.py:95-96); here they
    are explicit arguments (default: config.max_d and start + (D-1)*interval).  `features`
    (N,H/4,W/4,C) skips the 2D towers (used by the per-image feature cache of inference.py); then
    `ref_image` (1,Himg,Wimg,3) supplies the reference image the refinement tower looks at.
    With config.refinement the third result is the residual depth map (predictlib.py:86-92)."""
    from .model import inference_mem, inference_winner_take_all
    D = int(depth_num if depth_num is not None else config.max_d)
    if config.regularization == "3DCNN":
        d, p = inference_mem(full_images, scaled_cams, D, depth_start, depth_interval,
                             config.network_mode, inverse_depth=config.inverse_depth,
                             weights=weights, view_num=config.view_num, features=features)
        if config.refinement:
            from .refine import depth_refine
            if weights.refine is None:
                raise ValueError("config.refinement needs weights.refine (MVSNetWeights.from_numpy(refine=...))")
            if ref_image is None:
                ref_image = full_images[:, 0] if full_images.dim() == 5 else full_images[0:1]
            d, residual = depth_refine(d, ref_image.to(d.device), p, D, depth_start, depth_interval, weights.refine,
                                       upsample_depth=config.upsample_before_refinement,
                                       refine_with_confidence=config.refine_with_confidence)
            return d, p, residual
    elif config.regularization == "GRU":
        if depth_end is None:
            depth_end = float(depth_start) + (D - 1) * float(depth_interval)
        d, p = inference_winner_take_all(full_images, scaled_cams, D, depth_start, depth_end,
                                         network_mode=config.network_mode, reg_type="GRU",
                                         inverse_depth=config.inverse_depth, weights=weights,
                                         view_num=config.view_num, features=features)
    else:
        raise NotImplementedError(config.regularization)          # predictlib.py:97-98
    return d, p, None


def write_output_slice(output_dir, out_depth_map, out_prob_map, out_ref_image, out_ref_cam, out_index,
                       visualize=False, prob_upsample=None):
    """predictlib.py:105-159: <idx>_init.pfm, <idx>_prob.pfm, <idx>_depth.png (uint16 mm),
    <idx>_prob.png (x65535), <idx>.jpg, <idx>.txt with <idx> the un-padded reference index.
    out_ref_image: (H,W,3) in the pipeline's BGR order (write_reference_image swaps back to RGB)."""
    from PIL import Image
    depth = np.squeeze(np.asarray(out_depth_map)).astype(np.float32)
    prob = np.squeeze(np.asarray(out_prob_map)).astype(np.float32)
    if prob_upsample:                 # depth was refined at input resolution: nearest-neighbour prob (predictlib.py:110-115)
        from .mvs_data_generation import scale_image
        prob = scale_image(prob, prob_upsample, "nearest")
    idx = int(np.squeeze(out_index))
    pp.write_pfm(os.path.join(output_dir, "{}_init.pfm".format(idx)), depth)
    pp.write_pfm(os.path.join(output_dir, "{}_prob.pfm".format(idx)), prob)
    pp.write_png16(os.path.join(output_dir, "{}_depth.png".format(idx)), pp.depth_to_uint16(depth))
    pp.write_png16(os.path.join(output_dir, "{}_prob.png".format(idx)), pp.confidence_to_uint16(prob))
    if out_ref_image is not None:
        img = np.asarray(out_ref_image)
        img = np.clip(img, 0, 255).astype(np.uint8)[:, :, ::-1]      # BGR -> RGB (preprocess.py:208-212)
        Image.fromarray(np.ascontiguousarray(img)).save(os.path.join(output_dir, "{}.jpg".format(idx)))
    if out_ref_cam is not None:
        pp.write_cam(os.path.join(output_dir, "{}.txt".format(idx)), np.asarray(out_ref_cam))
    return idx


def write_output(output_dir, depth_batch, prob_batch, images_batch, cams_batch, index_batch):
    """predictlib.py:162-177 for batch_size 1..B."""
    start = time.time()
    for i in range(len(index_batch)):
        write_output_slice(output_dir, depth_batch[i], prob_batch[i], images_batch[i][0], cams_batch[i][0],
                           index_batch[i])
    return time.time() - start
