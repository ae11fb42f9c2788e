"""CPU restatement (numpy) of the reference's train-time augmentation pipeline -- TEST INFRASTRUCTURE ONLY
(imported by tests/ and tools/bench_augment.py's CPU leg; the product never imports it).

Pipeline (configs/_base_/datasets/pascal_voc12.py:6-18, cityscapes_1024x1024.py:3-15):
    RandomResize(scale, ratio_range, keep_ratio=True) -> RandomCrop(crop_size, cat_max_ratio) ->
    RandomFlip(prob) -> PhotoMetricDistortion -> PackSegInputs

What is restated from files that ARE in /root/reference (pinned by construction: same statements, same order of
numpy.random calls):
  * RandomCrop                 mmseg/datasets/transforms/transforms.py:208-337
  * PhotoMetricDistortion      mmseg/datasets/transforms/transforms.py:583-750  (convert / brightness / contrast /
                               saturation / hue / transform)
  * RandomFlip._flip           mmseg/datasets/transforms/transforms.py:1013-1033
  * PackSegInputs              mmseg/datasets/transforms/formatting.py:50-107

PARITY UNPINNED (third-party code absent from /root/reference and from this image: mmcv>=2.0.0rc4,<2.2.0 and
opencv-python; restated from their published algorithms, no golden vector available here):
  * mmcv RandomResize._random_sample_ratio, Resize._resize_img/_resize_seg, imrescale / rescale_size / _scale_size
  * cv2.resize INTER_LINEAR on 8-bit images (fixed-point: 11-bit coefficients, the 8U vertical pass
    ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2)>>2), INTER_NEAREST (floor(dx*scale))
  * cv2.cvtColor BGR2HSV / HSV2BGR on 8-bit images (H in [0,180): 12-bit division tables forward, float sector
    formula backward)
  * mmcv RandomFlip._choose_direction (numpy.random.choice over [direction, None] with p=[prob, 1-prob])
"""
import numpy as np


# ----------------------------------------------------------------------------- mmcv geometry helpers
def rescale_size(old_wh, scale):
    """mmcv.image.geometric.rescale_size with a (long, short) tuple scale -> (new_w, new_h)"""
    w, h = old_wh
    if isinstance(scale, (int, float)):
        f = float(scale)
    else:
        f = min(max(scale) / max(h, w), min(scale) / min(h, w))
    return int(w * float(f) + 0.5), int(h * float(f) + 0.5)


def _cv_round_half_even(x):
    return np.rint(x)


def _linear_coeffs(src, dst):
    """cv::resize INTER_LINEAR tables for one axis: source index, fixed-point (alpha0, alpha1)"""
    scale = 1.0 / (float(dst) / float(src))
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo], s[lo] = 0.0, 0
    hi = s >= src - 1
    f[hi], s[hi] = 0.0, src - 1
    a1 = _cv_round_half_even(f * np.float32(2048.0)).astype(np.int64)
    a0 = _cv_round_half_even((np.float32(1.0) - f) * np.float32(2048.0)).astype(np.int64)
    s1 = np.minimum(s + 1, src - 1)
    return s, s1, a0, a1


def resize_bilinear_u8(img, new_wh):
    """cv2.resize(img, (w, h), interpolation=cv2.INTER_LINEAR) for uint8 HxWxC"""
    H, W = img.shape[:2]
    nw, nh = new_wh
    if (nw, nh) == (W, H):
        return img.copy()
    x0, x1, ax0, ax1 = _linear_coeffs(W, nw)
    y0, y1, by0, by1 = _linear_coeffs(H, nh)
    src = img.astype(np.int64)
    rows = src[:, x0] * ax0[None, :, None] + src[:, x1] * ax1[None, :, None]       # H x nw x C, scaled by 2048
    s0, s1 = rows[y0], rows[y1]
    out = (((by0[:, None, None] * (s0 >> 4)) >> 16) + ((by1[:, None, None] * (s1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def resize_nearest(seg, new_wh):
    """cv2.resize(..., interpolation=cv2.INTER_NEAREST): sx = min(floor(dx * scale), W - 1)"""
    H, W = seg.shape[:2]
    nw, nh = new_wh
    fx, fy = 1.0 / (float(nw) / W), 1.0 / (float(nh) / H)
    sx = np.minimum(np.floor(np.arange(nw) * fx).astype(np.int64), W - 1)
    sy = np.minimum(np.floor(np.arange(nh) * fy).astype(np.int64), H - 1)
    return seg[sy][:, sx]


# ----------------------------------------------------------------------------- cv2 8-bit HSV
_SDIV = np.zeros(256, np.int64)
_HDIV = np.zeros(256, np.int64)
_SDIV[1:] = np.rint((255 << 12) / (1.0 * np.arange(1, 256))).astype(np.int64)
_HDIV[1:] = np.rint((180 << 12) / (6.0 * np.arange(1, 256))).astype(np.int64)


def bgr2hsv_u8(img):
    b, g, r = (img[..., k].astype(np.int64) for k in range(3))
    v = np.maximum(np.maximum(b, g), r)
    vmin = np.minimum(np.minimum(b, g), r)
    diff = v - vmin
    vr, vg = v == r, v == g
    s = (diff * _SDIV[v] + (1 << 11)) >> 12
    h = np.where(vr, g - b, np.where(vg, b - r + 2 * diff, r - g + 4 * diff))
    h = (h * _HDIV[diff] + (1 << 11)) >> 12
    h = np.where(h < 0, h + 180, h)
    return np.stack([h, s, v], -1).astype(np.uint8)


_SECTOR = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])


def hsv2bgr_u8(hsv):
    f32 = np.float32
    h = hsv[..., 0].astype(f32)
    s = hsv[..., 1].astype(f32) * f32(1.0 / 255.0)
    v = hsv[..., 2].astype(f32) * f32(1.0 / 255.0)
    h = h * f32(6.0 / 180.0)
    # (h in [0, 255*6/180): one wrap at most)
    h = np.where(h >= f32(6.0), h - f32(6.0), h).astype(f32)
    sector = np.floor(h).astype(np.int64)
    h = (h - sector.astype(f32)).astype(f32)
    bad = (sector < 0) | (sector >= 6)
    sector = np.where(bad, 0, sector)
    h = np.where(bad, f32(0), h).astype(f32)
    one = f32(1.0)
    tab = np.stack([v, v * (one - s), v * (one - s * h), v * (one - s * (one - h))], -1).astype(f32)
    idx = _SECTOR[sector]                                        # ... x 3 (b, g, r)
    bgr = np.take_along_axis(tab, idx, axis=-1)
    bgr = np.where((hsv[..., 1] == 0)[..., None], v[..., None], bgr).astype(f32)
    return np.clip(np.rint(bgr * f32(255.0)), 0, 255).astype(np.uint8)


# ----------------------------------------------------------------------------- PhotoMetricDistortion
def convert(img, alpha=1, beta=0):
    """transforms.py:621-640"""
    img = img.astype(np.float32) * alpha + beta
    img = np.clip(img, 0, 255)
    return img.astype(np.uint8)


class PhotoMetricDistortion:
    """transforms.py:583-750 (numpy.random is the module-level `random` of that file)"""

    def __init__(self, brightness_delta=32, contrast_range=(0.5, 1.5), saturation_range=(0.5, 1.5), hue_delta=18):
        self.brightness_delta = brightness_delta
        self.contrast_lower, self.contrast_upper = contrast_range
        self.saturation_lower, self.saturation_upper = saturation_range
        self.hue_delta = hue_delta

    def brightness(self, img):
        if np.random.randint(2):
            return convert(img, beta=np.random.uniform(-self.brightness_delta, self.brightness_delta))
        return img

    def contrast(self, img):
        if np.random.randint(2):
            return convert(img, alpha=np.random.uniform(self.contrast_lower, self.contrast_upper))
        return img

    def saturation(self, img):
        if np.random.randint(2):
            img = bgr2hsv_u8(img)
            img[:, :, 1] = convert(img[:, :, 1], alpha=np.random.uniform(self.saturation_lower, self.saturation_upper))
            img = hsv2bgr_u8(img)
        return img

    def hue(self, img):
        if np.random.randint(2):
            img = bgr2hsv_u8(img)
            img[:, :, 0] = (img[:, :, 0].astype(int) + np.random.randint(-self.hue_delta, self.hue_delta)) % 180
            img = hsv2bgr_u8(img)
        return img

    def __call__(self, img):
        img = self.brightness(img)
        mode = np.random.randint(2)
        if mode == 1:
            img = self.contrast(img)
        img = self.saturation(img)
        img = self.hue(img)
        if mode == 0:
            img = self.contrast(img)
        return img


# ----------------------------------------------------------------------------- the geometric transforms
def random_resize(img, seg, scale, ratio_range):
    """mmcv RandomResize (ratio mode) + Resize(keep_ratio=True): bilinear image, nearest label"""
    lo, hi = ratio_range
    ratio = np.random.random_sample() * (hi - lo) + lo
    sc = (int(scale[0] * ratio), int(scale[1] * ratio))
    H, W = img.shape[:2]
    new_wh = rescale_size((W, H), sc)
    return resize_bilinear_u8(img, new_wh), resize_nearest(seg, new_wh)


def random_crop(img, seg, crop_size, cat_max_ratio=1.0, ignore_index=255):
    """transforms.py:248-333"""
    def gen():
        margin_h = max(img.shape[0] - crop_size[0], 0)
        margin_w = max(img.shape[1] - crop_size[1], 0)
        oh = np.random.randint(0, margin_h + 1)
        ow = np.random.randint(0, margin_w + 1)
        return oh, oh + crop_size[0], ow, ow + crop_size[1]

    box = gen()
    if cat_max_ratio < 1.0:
        for _ in range(10):
            t = seg[box[0]:box[1], box[2]:box[3]]
            labels, cnt = np.unique(t, return_counts=True)
            cnt = cnt[labels != ignore_index]
            if len(cnt) > 1 and np.max(cnt) / np.sum(cnt) < cat_max_ratio:
                break
            box = gen()
    y1, y2, x1, x2 = box
    return img[y1:y2, x1:x2], seg[y1:y2, x1:x2], box


def random_flip(img, seg, prob):
    """mmcv RandomFlip._choose_direction (direction='horizontal') + mmseg _flip"""
    cur = np.random.choice(np.array(['horizontal', None], dtype=object), p=[prob, 1 - prob])
    if cur is not None:
        return np.flip(img, axis=1), np.flip(seg, axis=1), True
    return img, seg, False


def train_pipeline(img, seg, scale, ratio_range, crop_size, cat_max_ratio=0.75, flip_prob=0.5, pmd=None):
    """one sample through the reference's train pipeline -> (uint8 CHW BGR image, int64 1xHxW label, meta)"""
    img, seg = random_resize(img, seg, scale, ratio_range)
    resized = img.shape[:2]
    img, seg, box = random_crop(img, seg, crop_size, cat_max_ratio)
    img, seg, flipped = random_flip(img, seg, flip_prob)
    img = (pmd or PhotoMetricDistortion())(np.ascontiguousarray(img))
    meta = dict(img_shape=img.shape[:2], resized=resized, crop=box, flip=flipped)
    return np.ascontiguousarray(img.transpose(2, 0, 1)), seg[None].astype(np.int64), meta
