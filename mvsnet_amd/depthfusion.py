"""Post-processing hand-off to Gipuma / fusibile, mirroring mvsnet/depthfusion.py:

    python -m mvsnet_amd.depthfusion --dense_folder <dir> [--fusibile_exe_path <exe>]
        [--prob_threshold 0.8] [--disp_threshold 0.25] [--num_consistent 3]

1. probability filter: <idx>_init.pfm with depth 0 where <idx>_prob.pfm < threshold ->
   <idx>_prob_filtered.pfm (depthfusion.py:171-189);
2. Gipuma layout under <dense_folder>/points_mvsnet: cams/<name>.P (3x4 projection K[R|t]),
   images/<name>, 2333__<idx>/disp.dmb + normals.dmb (constant 1/sqrt(3) normals masked by depth > 0)
   (depthfusion.py:28-168);
3. the external `fusibile` binary is run when it exists (depthfusion.py:192-213); it is not part of
   this package, so without it the converted folder is the result.
File formats are byte-compatible with the reference's writers (.dmb: int32 header 1,H,W,C +
float32 data in the reference's element order).
"""
from __future__ import annotations

import argparse
import glob
import os
import shutil
import struct
import subprocess

import numpy as np

from .preprocess import load_cam, load_pfm, write_pfm


def read_gipuma_dmb(path):
    """depthfusion.py:28-40: header (type, H, W, C) then float32 in column-major (W,H,C) order."""
    with open(path, "rb") as f:
        _type, height, width, channel = struct.unpack("<iiii", f.read(16))
        array = np.frombuffer(f.read(), np.float32)
    array = array.reshape((width, height, channel), order="F")
    return np.transpose(array, (1, 0, 2)).squeeze()


def write_gipuma_dmb(path, image):
    """depthfusion.py:43-64 (3-channel images are stored plane by plane, as the reference does)."""
    image = np.asarray(image)
    height, width = image.shape[0], image.shape[1]
    channels = image.shape[2] if image.ndim == 3 else 1
    if image.ndim == 3:
        image = np.transpose(image, (2, 0, 1)).squeeze()
    with open(path, "wb") as f:
        f.write(struct.pack("<iiii", 1, height, width, channels))
        np.ascontiguousarray(image).tofile(f)


def mvsnet_to_gipuma_dmb(in_path, out_path):
    write_gipuma_dmb(out_path, load_pfm(in_path))


def mvsnet_to_gipuma_cam(in_path, out_path):
    """depthfusion.py:76-99: P = K_4x4(last row zeroed) @ E, first three rows, str() formatted."""
    cam = load_cam(in_path)
    extrinsic = cam[0]
    intrinsic = cam[1].copy()
    intrinsic[3, :] = 0
    projection = np.matmul(intrinsic, extrinsic)[0:3]
    with open(out_path, "w") as f:
        for i in range(3):
            for j in range(4):
                f.write(str(projection[i][j]) + " ")
            f.write("\n")
        f.write("\n")


def fake_colmap_normal(in_depth_path, out_normal_path):
    """depthfusion.py:102-123."""
    depth = read_gipuma_dmb(in_depth_path)
    normal = np.ones(depth.shape + (3,), dtype=depth.dtype) / 1.732050808
    mask = np.float32(depth > 0)[..., None]
    write_gipuma_dmb(out_normal_path, np.float32(normal * mask))


def _depth_image_names(depth_folder):
    return sorted(os.path.basename(p) for p in glob.glob(os.path.join(depth_folder, "*.jpg")))


def probability_filter(dense_folder, prob_threshold):
    """depthfusion.py:171-189."""
    depth_folder = os.path.join(dense_folder, "depths_mvsnet")
    for name in _depth_image_names(depth_folder):
        prefix = os.path.splitext(name)[0]
        depth = load_pfm(os.path.join(depth_folder, prefix + "_init.pfm")).copy()
        prob = load_pfm(os.path.join(depth_folder, prefix + "_prob.pfm"))
        depth[prob < prob_threshold] = 0
        write_pfm(os.path.join(depth_folder, prefix + "_prob_filtered.pfm"), depth)


def mvsnet_to_gipuma(dense_folder, gipuma_point_folder):
    """depthfusion.py:126-168."""
    depth_folder = os.path.join(dense_folder, "depths_mvsnet")
    names = _depth_image_names(depth_folder)
    cam_folder = os.path.join(gipuma_point_folder, "cams")
    image_folder = os.path.join(gipuma_point_folder, "images")
    for d in (gipuma_point_folder, cam_folder, image_folder):
        os.makedirs(d, exist_ok=True)
    for name in names:
        prefix = os.path.splitext(name)[0]
        mvsnet_to_gipuma_cam(os.path.join(depth_folder, prefix + ".txt"), os.path.join(cam_folder, name + ".P"))
        shutil.copy(os.path.join(depth_folder, name), os.path.join(image_folder, name))
        sub = os.path.join(gipuma_point_folder, "2333__" + prefix)
        os.makedirs(sub, exist_ok=True)
        mvsnet_to_gipuma_dmb(os.path.join(depth_folder, prefix + "_prob_filtered.pfm"), os.path.join(sub, "disp.dmb"))
        fake_colmap_normal(os.path.join(sub, "disp.dmb"), os.path.join(sub, "normals.dmb"))
    return names


def fusibile_command(point_folder, fusibile_exe_path, disp_thresh, num_consistent):
    """depthfusion.py:192-211 as an argument list."""
    return [fusibile_exe_path, "-input_folder", point_folder + "/", "-p_folder", os.path.join(point_folder, "cams") + "/",
            "-images_folder", os.path.join(point_folder, "images") + "/", "--depth_min=0.001", "--depth_max=100000",
            "--normal_thresh=360", "--disp_thresh=" + str(disp_thresh), "--num_consistent=" + str(num_consistent)]


def depth_map_fusion(point_folder, fusibile_exe_path, disp_thresh, num_consistent):
    cmd = fusibile_command(point_folder, fusibile_exe_path, disp_thresh, num_consistent)
    print(" ".join(cmd))
    if not (fusibile_exe_path and os.path.isfile(fusibile_exe_path)):
        print("fusibile executable not found: skipping the fusion step (external tool, not part of this package)")
        return None
    return subprocess.run(cmd, check=False).returncode


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--dense_folder", type=str, required=True)
    ap.add_argument("--fusibile_exe_path", type=str, default="")
    ap.add_argument("--prob_threshold", type=float, default=0.8)
    ap.add_argument("--disp_threshold", type=float, default=0.25)
    ap.add_argument("--num_consistent", type=float, default=3)
    a = ap.parse_args(argv)
    point_folder = os.path.join(a.dense_folder, "points_mvsnet")
    os.makedirs(point_folder, exist_ok=True)
    print("filter depth map with probability map")
    probability_filter(a.dense_folder, a.prob_threshold)
    print("Convert mvsnet output to gipuma input")
    mvsnet_to_gipuma(a.dense_folder, point_folder)
    print("Run depth map fusion & filter")
    depth_map_fusion(point_folder, a.fusibile_exe_path, a.disp_threshold, a.num_consistent)


if __name__ == "__main__":
    main()
