"""Host-side input pipeline: session directories -> (images, cams) batches for the hot path.

Restates the inference branch of the reference's data model without cv2 / imageio / TensorFlow:
  * session format: covisibility.json, images/<i>.jpg, cameras/<i>.json
    (mvsnet/mvs_data_generation/cluster_generator.py:140-156, mvs_cluster.py:63-127)
  * per cluster: load -> scale-to-cover -> centre-crop -> per-image standardise -> cams scaled to
    the feature resolution (cluster_generator.py:234-286, mvs_data_generation/utils.py:33-153).
Pure numpy + Pillow; this is plumbing outside the HIP hot path.
"""
from __future__ import annotations

import json
import math
import os

import numpy as np


def center_image(img):
    """mvs_data_generation/utils.py:33-38: per-image, per-channel standardisation."""
    img = img.astype(np.float32)
    var = np.var(img, axis=(0, 1), keepdims=True)
    mean = np.mean(img, axis=(0, 1), keepdims=True)
    return (img - mean) / (np.sqrt(var) + 0.00000001)


def scale_camera(cam, scale=1):
    """mvs_data_generation/utils.py:61-71: focal lengths and principal point times scale."""
    new_cam = np.copy(cam)
    new_cam[1][0][0] = cam[1][0][0] * scale
    new_cam[1][1][1] = cam[1][1][1] * scale
    new_cam[1][0][2] = cam[1][0][2] * scale
    new_cam[1][1][2] = cam[1][1][2] * scale
    return new_cam


def scale_image(image, scale=1, interpolation="linear"):
    """cv2.resize(image, None, fx=scale, fy=scale, INTER_LINEAR | INTER_NEAREST)
    (mvs_data_generation/utils.py:81-86) restated: output size = round(n*scale), sample position
    src = (dst + 0.5)/scale - 0.5 with edge replication (linear) or floor(dst/scale) (nearest).
    uint8 inputs are interpolated in float and rounded (cv2 uses 11-bit fixed point: results may
    differ by one grey level on rare pixels)."""
    img = np.asarray(image)
    h, w = img.shape[:2]
    nh, nw = int(round(h * scale)), int(round(w * scale))
    if scale == 1:                    # every sample position is a pixel centre: the identity for both interpolations
        return img
    if interpolation == "nearest":
        ys = np.minimum((np.arange(nh) / scale).astype(np.int64), h - 1)
        xs = np.minimum((np.arange(nw) / scale).astype(np.int64), w - 1)
        return img[ys][:, xs]
    inv = 1.0 / scale

    def taps(n_out, n_in):
        s = (np.arange(n_out, dtype=np.float64) + 0.5) * inv - 0.5
        i0 = np.floor(s).astype(np.int64)
        f = s - i0
        lo = np.clip(i0, 0, n_in - 1)
        hi = np.clip(i0 + 1, 0, n_in - 1)
        return lo, hi, f

    y0, y1, fy = taps(nh, h)
    x0, x1, fx = taps(nw, w)
    src = img if img.ndim == 3 else img[..., None]
    # only the rows / columns that are sampled are converted to float64 (same arithmetic as on the whole image)
    r0, r1 = src[y0], src[y1]
    top = r0[:, x0].astype(np.float64) * (1 - fx)[None, :, None] + r0[:, x1].astype(np.float64) * fx[None, :, None]
    bot = r1[:, x0].astype(np.float64) * (1 - fx)[None, :, None] + r1[:, x1].astype(np.float64) * fx[None, :, None]
    out = top * (1 - fy)[:, None, None] + bot * fy[:, None, None]
    if img.ndim == 2:
        out = out[..., 0]
    if img.dtype == np.uint8:
        return np.clip(np.floor(out + 0.5), 0, 255).astype(np.uint8)
    return out.astype(img.dtype)


def mask_depth_image(depth_image, min_depth, max_depth):
    """mvs_data_generation/utils.py:156-163: two cv2.threshold calls -- THRESH_TOZERO keeps values
    strictly above min_depth, THRESH_TOZERO_INV zeroes values strictly above max_depth -- then a
    trailing channel axis."""
    d = np.asarray(depth_image)
    d = np.where(d > min_depth, d, 0)
    d = np.where(d > max_depth, 0, d).astype(np.asarray(depth_image).dtype)
    return d[:, :, None]


def flip_cams(cams, depth_num):
    """mvs_data_generation/utils.py:166-171: reverse the reference view's depth range (start at the far
    plane, negative interval) -- the GRU training augmentation.  Returns a copy."""
    cams = np.copy(cams)
    cams[0][1, 3, 0] = cams[0][1, 3, 0] + (depth_num - 1) * cams[0][1, 3, 1]
    cams[0][1, 3, 1] = -cams[0][1, 3, 1]
    return cams


def scale_mvs_input(images, cams, scale=1, depth_image=None):
    """mvs_data_generation/utils.py:107-118; the GT depth is resized with nearest neighbour."""
    images, cams = [scale_image(i, scale) for i in images], [scale_camera(c, scale) for c in cams]
    if depth_image is None:
        return images, cams
    return images, cams, scale_image(depth_image, scale, "nearest")


def crop_mvs_input(images, cams, width, height, base_image_size, depth_image=None):
    """mvs_data_generation/utils.py:121-153: centre-crop to at most (height, width), otherwise
    round the size UP to a multiple of base_image_size (as the reference does), shifting the
    principal point.  A GT depth image is cropped with the window of the last view (as the reference)."""
    images, cams = list(images), [np.copy(c) for c in cams]
    for view in range(len(images)):
        h, w = images[view].shape[0:2]
        new_h = height if h > height else int(math.ceil(h / base_image_size) * base_image_size)
        new_w = width if w > width else int(math.ceil(w / base_image_size) * base_image_size)
        start_h = int(math.ceil((h - new_h) / 2))
        start_w = int(math.ceil((w - new_w) / 2))
        images[view] = images[view][start_h:start_h + new_h, start_w:start_w + new_w]
        cams[view][1][0][2] = cams[view][1][0][2] - start_w
        cams[view][1][1][2] = cams[view][1][1][2] - start_h
    if depth_image is not None:
        return images, cams, depth_image[start_h:start_h + new_h, start_w:start_w + new_w]
    return images, cams


_CAMERA_JSON = {}          # path -> (mtime_ns, size, parsed): a session's cameras are listed by up to view_num clusters each


def _camera_json(path):
    st = os.stat(path)
    hit = _CAMERA_JSON.get(path)
    if hit is None or hit[0] != st.st_mtime_ns or hit[1] != st.st_size:
        with open(path) as f:
            hit = (st.st_mtime_ns, st.st_size, json.load(f))
        if len(_CAMERA_JSON) > 4096:
            _CAMERA_JSON.clear()
        _CAMERA_JSON[path] = hit
    return hit[2]


class Cluster:
    """One reference view and its covisible source views (mvs_cluster.py:27-207)."""

    def __init__(self, session_dir, ref_index, views, min_depth, max_depth, view_num,
                 image_width=1024, image_height=768, depth_num=256, interval_scale=1.0):
        self.session_dir = session_dir
        self.ref_index = int(ref_index)
        self.views = views
        self.min_depth, self.max_depth = min_depth, max_depth
        self.view_num = view_num
        self.image_width, self.image_height = image_width, image_height
        self.depth_num, self.interval_scale = depth_num, interval_scale
        indices = [int(self.ref_index)] + [int(v) for v in views]
        indices += [int(self.ref_index)] * max(0, view_num - len(indices))   # pad with the reference
        self.indices = indices[:view_num]                                    # mvs_cluster.py:128-140
        self.rescale = 1.0

    def image_path(self, index):
        return os.path.join(self.session_dir, "images", "{}.jpg".format(index))

    def camera_path(self, index):
        return os.path.join(self.session_dir, "cameras", "{}.json".format(index))

    def depth_path(self, index):
        return os.path.join(self.session_dir, "depths", "{}.png".format(index))

    def load_depth(self, index):
        """uint16 millimetre depth PNG, or None when missing (mvs_cluster.py:78-89)."""
        from PIL import Image
        try:
            return np.asarray(Image.open(self.depth_path(index))).astype(np.uint16)
        except Exception:
            return None

    def masked_reference_depth(self):
        """mvs_cluster.py:163-177: GT depth brought to the input image's scale (nearest), values
        outside (min_depth, max_depth] zeroed; call after images()."""
        depth = self.load_depth(self.ref_index)
        if depth is None:
            return None
        scale = float(self.original_image_shape[0]) / float(depth.shape[0])
        depth = scale_image(depth, scale, "nearest")
        return mask_depth_image(depth, self.min_depth, self.max_depth)

    def load_image(self, index):
        """RGB decode then RGB->BGR, as mvs_cluster.py:72-76."""
        from PIL import Image
        rgb = np.asarray(Image.open(self.image_path(index)).convert("RGB"))
        return np.ascontiguousarray(rgb[:, :, ::-1])

    def load_camera(self, index):
        """(2,4,4): pose (translation metres -> mm), intrinsics, (min, interval, num, max)
        (mvs_cluster.py:91-127)."""
        data = _camera_json(self.camera_path(index))
        interval = ((self.max_depth - self.min_depth) / (self.depth_num - 1)) * self.interval_scale
        cam = np.zeros((2, 4, 4))
        for i in range(4):
            for j in range(4):
                cam[0, i, j] = data["pose"]["matrix"]["{},{}".format(i, j)]
        cam[0, 0:3, 3] *= 1000
        intr = data["intrinsics"]
        cam[1, 0, 0], cam[1, 1, 1] = intr["fx"], intr["fy"]
        cam[1, 0, 2], cam[1, 1, 2], cam[1, 2, 2] = intr["px"], intr["py"], 1.0
        cam[1, 3, 0], cam[1, 3, 1] = self.min_depth, interval
        cam[1, 3, 2], cam[1, 3, 3] = self.depth_num, self.max_depth
        return cam

    def images(self):
        imgs = [self.load_image(i) for i in self.indices]
        self.original_image_shape = imgs[0].shape                            # mvs_cluster.py:154
        h_scale = max(float(self.image_height) / im.shape[0] for im in imgs)
        w_scale = max(float(self.image_width) / im.shape[1] for im in imgs)
        self.rescale = max(h_scale, w_scale)                                 # mvs_cluster.py:178-192
        return imgs

    def cameras(self):
        return [self.load_camera(i) for i in self.indices]


class ClusterGenerator:
    """Iterator of cluster_generator.py:27-286.  'inference' / 'test': `data_dir` is one session; yields
    (output_images, input_images, output_cams, full_cams, image_index[, depth]) once per cluster.
    'train' / 'val' (cluster_generator.py:61-64,166-223): `data_dir`/train|val holds one sub-directory per
    session; clusters of all sessions are shuffled (`seed`; the reference uses the global `random` state) and
    each yields (images (N,H,W,3) centred, cams (N,2,4,4) scaled to the output, depth (H/4,W/4,1) nearest-
    neighbour down-sampled masked GT, full depth (H,W,1)); clusters that fail to load are skipped as in the
    reference (:217-220); with `flip_cams` (GRU training) every cluster is also yielded with its depth range
    reversed."""

    def __init__(self, data_dir, view_num=3, image_width=1024, image_height=768, depth_num=256,
                 interval_scale=1, base_image_size=1, include_empty=False, mode="inference",
                 output_scale=0.25, max_clusters_per_session=None, flip_cams=False, sessions_frac=1.0, seed=0):
        if mode not in ("inference", "test", "train", "val"):
            raise ValueError("mode must be one of inference, test, train, val")
        self.mode = mode              # 'test' also yields the masked GT depth (cluster_generator.py:244-251)
        self.data_dir = data_dir
        self.view_num = view_num
        self.image_width, self.image_height = image_width, image_height
        self.depth_num, self.interval_scale = depth_num, interval_scale
        self.base_image_size = base_image_size
        self.include_empty = include_empty
        self.output_scale = output_scale
        self.max_clusters_per_session = max_clusters_per_session
        self.flip_cams = flip_cams
        self.clusters = []
        # per-image cache of the inference path (_prepare_cached), shared by the loader threads: created HERE, not lazily
        # from the threads (a thread could see the cache before the lock existed); image sizes (never evicted, a tuple per
        # image) live apart from the decoded images (LRU of `image_cache_limit` entries)
        import collections
        import threading
        self._img_cache, self._img_sizes, self._img_lock = collections.OrderedDict(), {}, threading.Lock()
        self.image_cache_limit = 192
        if mode in ("train", "val"):
            sessions_dir = os.path.join(data_dir, mode)
            sessions = sorted(f for f in os.listdir(sessions_dir)
                              if not f.startswith(".") and not f.endswith(".txt") and not f.endswith(".pickle"))
            for session in sessions[:int(len(sessions) * sessions_frac)]:
                try:
                    self.load_clusters(os.path.join(sessions_dir, session), self.clusters)
                except (OSError, ValueError, KeyError):
                    continue                                   # cluster_generator.py:110-113
            import random
            random.Random(seed).shuffle(self.clusters)         # :124-125
        else:
            self.load_clusters(data_dir, self.clusters)

    def prepare_training(self, c, center=True):
        """cluster_generator.py:171-196.  center=False: the images come back as the cropped uint8 BGR stack and the caller
        standardises them (on the device: `Trainer.loss` takes uint8 -- a quarter of the bytes through the loader's pipe and
        the upload, no float32 reductions on the loader)."""
        images, cams = c.images(), c.cameras()
        depth = c.masked_reference_depth()
        if depth is None:
            raise IOError("no ground-truth depth for reference view %d" % c.ref_index)
        images, cams, depth = scale_mvs_input(images, cams, scale=c.rescale, depth_image=depth[:, :, 0])
        images, cams, depth = crop_mvs_input(images, cams, self.image_width, self.image_height,
                                             self.base_image_size, depth)
        images = np.stack([center_image(i) if center else np.ascontiguousarray(i) for i in images], axis=0)
        depth = depth.astype(np.float32)
        rescaled = scale_image(depth, self.output_scale, "nearest")[:, :, None]
        cams = np.stack([scale_camera(cam, self.output_scale) for cam in cams], axis=0)
        return images, cams, rescaled, depth[:, :, None]

    def load_clusters(self, session_dir, clusters):
        with open(os.path.join(session_dir, "covisibility.json")) as f:
            data = json.load(f)
        added = 0
        max_clusters = len(data) if self.max_clusters_per_session is None else self.max_clusters_per_session
        for d in data:
            if not self.include_empty and not data[d]["views"]:
                continue
            if added < max_clusters:
                clusters.append(Cluster(session_dir, int(d), data[d]["views"], data[d]["min_depth"],
                                        data[d]["max_depth"], self.view_num, self.image_width,
                                        self.image_height, self.depth_num, self.interval_scale))
                added += 1

    def _prepare_cached(self, c, center=True):
        """Inference mode: what prepare() does per image -- decode, rescale, crop, centre, output-scale -- depends only on the
        image file and on (rescale, crop window), so it is kept per image across the reference views of a session that list
        the image as a source (the reference decodes and resizes every image once per cluster, cluster_generator.py:234-286).
        Same values as the uncached path; an LRU of `image_cache_limit` (192) images per generator, shared by the loader threads.
        center=False: the input images come back as cropped uint8 BGR (N,H,W,3) and the caller standardises them (on the
        device, inference.center_images_device): a quarter of the bytes to upload and no 1.3 M-element float32 reductions per image
        on the loader threads."""
        raw = {}

        def raw_image(i):
            if i not in raw:
                raw[i] = c.load_image(i)
            return raw[i]
        # the cluster-level rescale needs every image's size (mvs_cluster.py:178-192): sizes are cached with the images
        sizes = []
        for i in c.indices:
            with self._img_lock:
                hit = self._img_sizes.get((c.session_dir, i))
            if hit is None:
                hit = raw_image(i).shape
                with self._img_lock:
                    self._img_sizes[(c.session_dir, i)] = hit
            sizes.append(hit)
        c.original_image_shape = sizes[0]
        c.rescale = max(max(float(self.image_height) / s_[0] for s_ in sizes), max(float(self.image_width) / s_[1] for s_ in sizes))
        cams = c.cameras()
        full_cams, out_cams = self.cluster_cameras(c, cams, sizes)
        ins, outs = [], []
        for v, i in enumerate(c.indices):
            key = (c.session_dir, i, round(float(c.rescale), 12), bool(center))
            with self._img_lock:
                hit = self._img_cache.get(key)
                if hit is not None:
                    self._img_cache.move_to_end(key)
            if hit is None:
                im, _ = scale_mvs_input([raw_image(i)], [cams[v]], scale=c.rescale)
                cr, _ = crop_mvs_input(im, [cams[v]], self.image_width, self.image_height, self.base_image_size)
                hit = (center_image(cr[0]) if center else np.ascontiguousarray(cr[0]), scale_image(cr[0], self.output_scale))
                with self._img_lock:
                    self._img_cache[key] = hit
                    while len(self._img_cache) > self.image_cache_limit:
                        self._img_cache.popitem(last=False)
            ins.append(hit[0]); outs.append(hit[1])
        return (np.stack(outs, axis=0), np.stack(ins, axis=0), out_cams, full_cams, c.ref_index)

    def cluster_cameras(self, c, cams, sizes):
        """Cameras of a cluster after scale-to-cover (c.rescale), centre-crop and output scaling, from the ORIGINAL image
        sizes alone (crop_mvs_input only shifts the principal point by the crop offset): (full_cams (N,2,4,4), out_cams)."""
        full_cams, out_cams = [], []
        for v in range(len(c.indices)):
            _, cm = scale_mvs_input([], [cams[v]], scale=c.rescale)
            h0, w0 = int(round(sizes[v][0] * c.rescale)), int(round(sizes[v][1] * c.rescale))
            dummy = [np.empty((h0, w0, 0), np.uint8)]
            _, cc = crop_mvs_input(dummy, cm, self.image_width, self.image_height, self.base_image_size)
            _, oc = scale_mvs_input([], cc, scale=self.output_scale)
            full_cams.append(cc[0]); out_cams.append(oc[0])
        return np.stack(full_cams, axis=0), np.stack(out_cams, axis=0)

    def prepare(self, c: Cluster, center=True):
        if self.mode == "inference" and type(c) is Cluster:
            return self._prepare_cached(c, center)
        if not center:
            raise NotImplementedError("center=False is the cached session-format inference path")
        images = c.images()
        cams = c.cameras()
        depth = None
        if self.mode == "test":
            depth = c.masked_reference_depth()
            if depth is None:
                raise IOError("no ground-truth depth for reference view %d" % c.ref_index)
            images, cams, depth = scale_mvs_input(images, cams, scale=c.rescale, depth_image=depth[:, :, 0])
            cropped_images, cropped_cams, depth = crop_mvs_input(images, cams, self.image_width, self.image_height,
                                                                 self.base_image_size, depth)
            depth = depth.astype(np.float32)[:, :, None]
        else:
            images, cams = scale_mvs_input(images, cams, scale=c.rescale)
            cropped_images, cropped_cams = crop_mvs_input(images, cams, self.image_width, self.image_height,
                                                          self.base_image_size)
        full_cams = np.stack(cropped_cams, axis=0)
        input_images = np.stack([center_image(i) for i in cropped_images], axis=0)
        output_images, output_cams = scale_mvs_input(cropped_images, cropped_cams, scale=self.output_scale)
        out = (np.stack(output_images, axis=0), input_images, np.stack(output_cams, axis=0), full_cams, c.ref_index)
        return out + (depth,) if self.mode == "test" else out

    def __len__(self):
        return len(self.clusters)

    def __iter__(self):
        for c in self.clusters:
            if self.mode in ("train", "val"):
                try:
                    images, cams, rescaled, depth = self.prepare_training(c)
                except (OSError, ValueError, KeyError):
                    continue                                   # bad clusters are skipped (:217-220)
                yield images, cams, rescaled, depth
                if self.flip_cams:
                    yield images, flip_cams(cams, self.depth_num), rescaled, depth
            else:
                yield self.prepare(c)


# ------------------------------------------------------------------------------------------------
# upstream MVSNet project folders (README.md:165-215): images/%08d.jpg, cams/%08d_cam.txt, pair.txt
# ------------------------------------------------------------------------------------------------


def gen_pipeline_mvs_list(dense_folder, view_num):
    """mvsnet/preprocess.py:547-579: [[ref_image, ref_cam, view_image, view_cam, ...], ...] from pair.txt
    (the 10 best source views per reference image, best first; at most view_num - 1 are used)."""
    image_folder = os.path.join(dense_folder, "images")
    cam_folder = os.path.join(dense_folder, "cams")
    words = open(os.path.join(dense_folder, "pair.txt")).read().split()
    mvs_list, pos = [], 1
    for _ in range(int(words[0])):
        ref_index = int(words[pos]); pos += 1
        paths = [os.path.join(image_folder, "%08d.jpg" % ref_index),
                 os.path.join(cam_folder, "%08d_cam.txt" % ref_index)]
        all_view_num = int(words[pos]); pos += 1
        for view in range(min(view_num - 1, all_view_num)):
            view_index = int(words[pos + 2 * view])
            paths += [os.path.join(image_folder, "%08d.jpg" % view_index),
                      os.path.join(cam_folder, "%08d_cam.txt" % view_index)]
        pos += 2 * all_view_num
        mvs_list.append((ref_index, paths))
    return mvs_list


class PairCluster:
    """One reference view of an upstream-format project folder; duck-types Cluster."""

    def __init__(self, dense_folder, ref_index, paths, view_num, image_width, image_height, depth_num,
                 interval_scale):
        self.session_dir = dense_folder
        self.ref_index = int(ref_index)
        self.paths = list(paths)
        while len(self.paths) < 2 * view_num:           # pad missing views with the reference
            self.paths += self.paths[:2]
        self.view_num = view_num
        self.image_width, self.image_height = image_width, image_height
        self.depth_num, self.interval_scale = depth_num, interval_scale
        self.rescale = 1.0

    def images(self):
        from PIL import Image
        imgs = []
        for v in range(self.view_num):
            rgb = np.asarray(Image.open(self.paths[2 * v]).convert("RGB"))
            imgs.append(np.ascontiguousarray(rgb[:, :, ::-1]))          # BGR, as cv2.imread upstream
        h_scale = max(float(self.image_height) / im.shape[0] for im in imgs)
        w_scale = max(float(self.image_width) / im.shape[1] for im in imgs)
        self.rescale = max(h_scale, w_scale)
        return imgs

    def cameras(self):
        from .preprocess import load_cam
        return [load_cam(self.paths[2 * v + 1], self.interval_scale, self.depth_num)
                for v in range(self.view_num)]


class PairClusterGenerator(ClusterGenerator):
    """ClusterGenerator over an upstream MVSNet project folder (pair.txt)."""

    def load_clusters(self, session_dir, clusters):
        for ref_index, paths in gen_pipeline_mvs_list(session_dir, self.view_num):
            if self.max_clusters_per_session is not None and len(clusters) >= self.max_clusters_per_session:
                break
            clusters.append(PairCluster(session_dir, ref_index, paths, self.view_num, self.image_width,
                                        self.image_height, self.depth_num, self.interval_scale))


def make_generator(data_dir, *args, **kwargs):
    """Session format (covisibility.json) or upstream format (pair.txt), whichever the folder holds."""
    if os.path.isfile(os.path.join(data_dir, "covisibility.json")):
        return ClusterGenerator(data_dir, *args, **kwargs)
    if os.path.isfile(os.path.join(data_dir, "pair.txt")):
        return PairClusterGenerator(data_dir, *args, **kwargs)
    raise FileNotFoundError("%s holds neither covisibility.json nor pair.txt" % data_dir)


# ------------------------------------------------------------------------------------------------
# training input pipeline (train.py:189-247: tf.data from_generator + parallel_interleave + prefetch)
# ------------------------------------------------------------------------------------------------

_WORKER_GENERATORS = {}                # in a worker process: constructor arguments -> (generator, {(session, ref view): cluster})


def _training_task(gen_args, ident, center):
    """Runs in a host-pool worker: `prepare_training` of the cluster `ident` = (session directory, reference view) of the
    generator described by `gen_args` (built once per worker; its cluster list is the parent's -- same directory listing,
    same seed -- but clusters are looked up by identity, not by position)."""
    hit = _WORKER_GENERATORS.get(gen_args)
    if hit is None:
        gen = ClusterGenerator(**dict(gen_args))
        hit = _WORKER_GENERATORS[gen_args] = (gen, {(c.session_dir, c.ref_index): c for c in gen.clusters})
    gen, by_ident = hit
    return gen.prepare_training(by_ident[ident], center=center)


class TrainingPrefetcher:
    """Training batches prepared AHEAD of the optimisation step, in order.

    The reference feeds its towers from `tf.data`: generators interleaved over parallel threads with prefetching
    (train.py:208-247), so decoding, resizing and cropping of the next clusters overlap the current step.  Here the clusters
    of `order` (indices into `gen.clusters`) are prepared by the host pool's worker PROCESSES (`host_pool.get_pool`; the
    thread that launches the kernels is the bottleneck of a step and shares no GIL with them), `ahead` clusters in flight,
    and handed out in `order`.  `workers` = 0: a pool of threads in this process; "inline": nothing ahead, every cluster is
    prepared on the calling thread when it is asked for (one generator, no prefetch).  A cluster that fails to load
    (OSError / ValueError / KeyError, cluster_generator.py:217-220) is skipped, or raised when `strict` (several ranks:
    a skipped step would leave the gradient all-reduce without its peer).  Iterating yields (position in `order`, batch) with
    batch = (images, cams, depth (H/4,W/4,1), full depth (H,W,1)); `center=False` leaves the images uint8 for the device."""

    def __init__(self, gen, gen_args, order, workers=None, ahead=None, center=True, strict=False):
        from . import host_pool
        self.gen, self.gen_args = gen, tuple(sorted(gen_args.items()))
        self.order = list(order)
        self.center, self.strict = bool(center), bool(strict)
        self.inline = workers == "inline"
        self.pool = None if self.inline else host_pool.get_pool(workers)
        self.threads = None
        if self.pool is None and not self.inline:
            from concurrent.futures import ThreadPoolExecutor
            self.threads = ThreadPoolExecutor(max_workers=4)
        n = self.pool.workers if self.pool is not None else 4
        self.ahead = max(1, int(ahead) if ahead is not None else 2 * n)       # prefetch_input_elements = 2 x generators (:243)

    def _submit(self, ci):
        c = self.gen.clusters[ci]
        if self.pool is not None:
            return self.pool.ex.submit(_training_task, self.gen_args, (c.session_dir, c.ref_index), self.center)
        return self.threads.submit(self.gen.prepare_training, c, self.center)

    def __iter__(self):
        import collections
        if self.inline:
            for pos, ci in enumerate(self.order):
                try:
                    batch = self.gen.prepare_training(self.gen.clusters[ci], self.center)
                except (OSError, ValueError, KeyError):
                    if self.strict:
                        raise
                    continue
                yield pos, batch
            return
        pending = collections.deque()
        it = iter(enumerate(self.order))
        for pos, ci in it:
            pending.append((pos, self._submit(ci)))
            if len(pending) >= self.ahead:
                break
        while pending:
            pos, fut = pending.popleft()
            for nxt, ci in it:
                pending.append((nxt, self._submit(ci)))
                break
            try:
                batch = fut.result()
            except (OSError, ValueError, KeyError):
                if self.strict:
                    raise
                continue
            yield pos, batch

    def close(self):
        if self.threads is not None:
            self.threads.shutdown(wait=False)
