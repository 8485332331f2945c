"""Camera-text and PFM readers/writers, byte-compatible with the reference's
mvsnet/preprocess.py:116-155 (load_cam), :273-292 (write_cam), :294-325 (load_pfm),
:327-356 (write_pfm), plus the uint16 PNG depth/confidence quantisation of :253-270.

Pure numpy + stdlib (the reference's cv2 / imageio / tf file_io are not needed for the formats).
"""
from __future__ import annotations

import re
import sys

import numpy as np


def load_cam(file, interval_scale=1, max_d=None):
    """Reads a camera txt (file object or path) into a (2,4,4) float64 array:
    cam[0] = extrinsic, cam[1][:3,:3] = intrinsic, cam[1][3] = (depth_min, depth_interval,
    depth_num, depth_max).  29/30/31-word variants as preprocess.py:132-154; for the 29-word
    form ``max_d`` replaces the reference's global FLAGS.max_d and must be given."""
    if isinstance(file, (str, bytes)):
        with open(file) as f:
            return load_cam(f, interval_scale, max_d)
    cam = np.zeros((2, 4, 4))
    words = file.read().split()
    for i in range(4):
        for j in range(4):
            cam[0][i][j] = float(words[4 * i + j + 1])
    for i in range(3):
        for j in range(3):
            cam[1][i][j] = float(words[3 * i + j + 18])
    if len(words) == 29:
        if max_d is None:
            raise ValueError("29-word camera file needs max_d (FLAGS.max_d in the reference)")
        cam[1][3][0] = float(words[27])
        cam[1][3][1] = float(words[28]) * interval_scale
        cam[1][3][2] = max_d
        cam[1][3][3] = cam[1][3][0] + cam[1][3][1] * cam[1][3][2]
    elif len(words) == 30:
        cam[1][3][0] = float(words[27])
        cam[1][3][1] = float(words[28]) * interval_scale
        cam[1][3][2] = float(words[29])
        cam[1][3][3] = cam[1][3][0] + cam[1][3][1] * cam[1][3][2]
    elif len(words) == 31:
        cam[1][3][0] = float(words[27])
        cam[1][3][1] = float(words[28]) * interval_scale
        cam[1][3][2] = float(words[29])
        cam[1][3][3] = float(words[30])
    else:
        cam[1][3][:] = 0
    return cam


def cam_text(cam):
    """The exact text write_cam emits (preprocess.py:277-290)."""
    out = ["extrinsic\n"]
    for i in range(4):
        out.append("".join(str(cam[0][i][j]) + " " for j in range(4)) + "\n")
    out.append("\n")
    out.append("intrinsic\n")
    for i in range(3):
        out.append("".join(str(cam[1][i][j]) + " " for j in range(3)) + "\n")
    out.append("\n" + str(cam[1][3][0]) + " " + str(cam[1][3][1]) + " " + str(cam[1][3][2]) + " "
               + str(cam[1][3][3]) + "\n")
    return "".join(out)


def write_cam(file, cam):
    with open(file, "w") as f:
        f.write(cam_text(cam))


def load_pfm(file):
    """Reads a PFM (binary file object or path) -> float32 array, top row first
    (preprocess.py:294-325; the reference's cv2.flip(data, 0) is np.flipud)."""
    if isinstance(file, (str,)):
        with open(file, "rb") as f:
            return load_pfm(f)
    header = file.readline().decode("latin-1").rstrip()
    if header == "PF":
        color = True
    elif header == "Pf":
        color = False
    else:
        raise Exception("Not a PFM file.")
    dim_match = re.match(r"^(\d+)\s(\d+)\s$", file.readline().decode("latin-1"))
    if dim_match:
        width, height = map(int, dim_match.groups())
    else:
        raise Exception("Malformed PFM header.")
    scale = float(file.readline().decode("latin-1").rstrip())
    data_type = "<f" if scale < 0 else ">f"
    data = np.frombuffer(file.read(), data_type)
    shape = (height, width, 3) if color else (height, width)
    data = np.reshape(data, shape)
    return np.ascontiguousarray(np.flipud(data)).astype(np.float32)


def pfm_encode(image, scale=1):
    """Bytes of write_pfm (preprocess.py:327-356): 'Pf\\n'|'PF\\n', '%d %d\\n' % (W,H),
    '%f\\n' % (-scale for little-endian data), raw rows bottom-to-top."""
    if image.dtype.name != "float32":
        raise Exception("Image dtype must be float32.")
    image = np.flipud(image)
    if len(image.shape) == 3 and image.shape[2] == 3:
        color = True
    elif len(image.shape) == 2 or (len(image.shape) == 3 and image.shape[2] == 1):
        color = False
    else:
        raise Exception("Image must have H x W x 3, H x W x 1 or H x W dimensions.")
    endian = image.dtype.byteorder
    if endian == "<" or (endian == "=" and sys.byteorder == "little"):
        scale = -scale
    head = ("PF\n" if color else "Pf\n") + "%d %d\n" % (image.shape[1], image.shape[0]) + "%f\n" % scale
    return head.encode("ascii") + image.tobytes()


def write_pfm(file, image, scale=1):
    data = pfm_encode(np.asarray(image), scale)
    with open(file, "wb") as f:
        f.write(data)


def depth_to_uint16(image):
    """preprocess.py:253-256: clip to [0, 65535] and truncate to uint16 (mm)."""
    return np.clip(image, 0, 65535).astype(np.uint16)


def confidence_to_uint16(image):
    """preprocess.py:261-270: probabilities in [0,1] scaled by 65535, clipped, uint16."""
    return np.clip(np.asarray(image, np.float32) * 65535, 0, 65535).astype(np.uint16)


def write_png16(file, image16):
    """Greyscale 16-bit PNG via Pillow (imageio.imsave in the reference).  zlib level 1 (round 4): the same pixels in a
    ~20 % larger file for a quarter of the encoding time (2.2 -> 0.5 ms per 160 x 128 map; the encoder was 60 % of a
    reference view's write time)."""
    from PIL import Image
    Image.fromarray(np.asarray(image16, np.uint16)).save(file, compress_level=1)
