"""Train-time augmentation pipeline on the GPU.

The reference runs ``RandomResize -> RandomCrop -> RandomFlip -> PhotoMetricDistortion -> PackSegInputs`` per sample
on dataloader CPU workers (configs/_base_/datasets/pascal_voc12.py:6-18, cityscapes_1024x1024.py:3-15).  Here the
transforms keep the reference's names, constructor arguments and -- call for call -- its ``numpy.random`` draws, but
each ``transform`` only RECORDS its decision in the sample's parameter block; all pixel work of the whole batch is
one launch of ``ledn_augment_batch`` (csrc/augment.hip), bit-identical to the CPU pipeline for the same draws.

    pipe = Compose(cfg['train_pipeline'])          # LoadImageFromFile / LoadAnnotations entries are skipped:
    out = pipe.batch([dict(img=uint8 HxWx3 device tensor (BGR, as cv2.imread decodes), gt_seg_map=uint8 HxW), ...])
    out['inputs']          # list of uint8 3 x h x w views into ONE N x 3 x crop_h x crop_w batch
    out['data_samples']    # SegDataSample(gt_sem_seg int64 1 x h x w, metainfo img_shape / scale_factor / flip ...)
    -> SegDataPreProcessor -> EncoderDecoder.loss

Only RandomCrop's ``cat_max_ratio`` test is data dependent: the class counts of the candidate crop come from
``ledn_aug_crop_hist`` (one small launch and a 1 KB device-to-host copy per candidate box; the reference does
np.unique on the CPU crop).  The draws stay sample by sample, as one dataloader worker makes them.

reference: mmseg/datasets/transforms/transforms.py:208-337 (RandomCrop), 583-750 (PhotoMetricDistortion),
956-1033 (RandomFlip), formatting.py:14-107 (PackSegInputs); mmcv (un-vendored): RandomResize, Resize, RandomFlip.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .ops import _run
from .segmentor import PixelData, SegDataSample

TRANSFORMS = {}


def register(cls):
    TRANSFORMS[cls.__name__] = cls
    return cls


def rescale_size(old_wh, scale):
    """mmcv.imrescale's size rule: (long edge, short edge) bound or a plain factor -> (new_w, new_h)"""
    w, h = old_wh
    if isinstance(scale, (int, float)):
        if scale <= 0:
            raise ValueError(f'Invalid scale {scale}, must be positive.')
        f = float(scale)
    elif isinstance(scale, (tuple, list)):
        f = min(max(scale) / max(h, w), min(scale) / min(h, w))
    else:
        raise TypeError(f'Scale must be a number or tuple of int, but got {type(scale)}')
    return int(w * float(f) + 0.5), int(h * float(f) + 0.5)


class _Params:
    """what the geometric / photometric transforms decided for one sample (-> ledn_aug_entry)"""

    def __init__(self, H, W):
        self.H, self.W = H, W
        self.RH, self.RW = H, W
        self.oy = self.ox = 0
        self.ch, self.cw = H, W
        self.flip = 0
        self.bright = self.contrast = self.sat = self.hue = None
        self.mode = 0


def _params(results):
    p = results.get('_aug')
    if p is None:
        img = results['img']
        p = results['_aug'] = _Params(int(img.shape[0]), int(img.shape[1]))
        results.setdefault('ori_shape', (p.H, p.W))
        results['img_shape'] = (p.H, p.W)
    return p


@register
class Resize:
    """mmcv Resize with keep_ratio (the test pipeline's fixed-scale resize and RandomResize's worker)"""

    def __init__(self, scale=None, scale_factor=None, keep_ratio=False, **kw):
        assert scale is not None or scale_factor is not None, '`scale` and`scale_factor` can not both be `None`'
        self.scale = scale if scale is None or not isinstance(scale, int) else (scale, scale)
        self.scale_factor = scale_factor
        self.keep_ratio = keep_ratio

    def transform(self, results):
        p = _params(results)
        assert (p.RH, p.RW) == (p.H, p.W) and (p.ch, p.cw) == (p.H, p.W), 'one resize, before the crop'
        scale = results.pop('scale', None) or self.scale
        if scale is None:
            f = self.scale_factor if isinstance(self.scale_factor, (tuple, list)) else (self.scale_factor,) * 2
            scale = (int(p.W * f[0] + 0.5), int(p.H * f[1] + 0.5))
        if self.keep_ratio:
            nw, nh = rescale_size((p.W, p.H), tuple(scale))
        else:
            nw, nh = int(scale[0]), int(scale[1])
        p.RH, p.RW = nh, nw
        p.ch, p.cw = nh, nw
        results['img_shape'] = (nh, nw)
        results['scale'] = (nw, nh)
        results['scale_factor'] = (nw / p.W, nh / p.H)
        results['keep_ratio'] = self.keep_ratio
        return results


@register
class RandomResize:
    """mmcv RandomResize, ratio mode: ratio = random_sample() * (max - min) + min; scale = int(base * ratio)"""

    def __init__(self, scale, ratio_range=None, resize_type='Resize', **resize_kwargs):
        self.scale, self.ratio_range = scale, ratio_range
        assert resize_type == 'Resize'
        self.resize = Resize(scale=0, **resize_kwargs)

    def _random_scale(self):
        if self.ratio_range is not None:
            lo, hi = self.ratio_range
            ratio = np.random.random_sample() * (hi - lo) + lo
            return int(self.scale[0] * ratio), int(self.scale[1] * ratio)
        if isinstance(self.scale[0], (tuple, list)):       # a (min, max) pair of scales: sampled edge by edge
            long_e = np.random.randint(min(max(s) for s in self.scale), max(max(s) for s in self.scale) + 1)
            short_e = np.random.randint(min(min(s) for s in self.scale), max(min(s) for s in self.scale) + 1)
            return long_e, short_e
        return tuple(self.scale)

    def transform(self, results):
        results['scale'] = self._random_scale()
        return self.resize.transform(results)


@register
class RandomCrop:
    """transforms.py:208-337.  The candidate box is drawn exactly as the reference draws it; with cat_max_ratio < 1
    the retry loop needs the class counts of the crop, which `Compose.batch` obtains on the device."""

    def __init__(self, crop_size, cat_max_ratio=1., ignore_index=255):
        assert isinstance(crop_size, int) or (isinstance(crop_size, (tuple, list)) and len(crop_size) == 2), \
            'The expected crop_size is an integer, or a tuple containing two intergers'
        if isinstance(crop_size, int):
            crop_size = (crop_size, crop_size)
        assert crop_size[0] > 0 and crop_size[1] > 0
        self.crop_size, self.cat_max_ratio, self.ignore_index = tuple(crop_size), cat_max_ratio, ignore_index

    def generate_crop_bbox(self, h, w):
        margin_h = max(h - self.crop_size[0], 0)
        margin_w = max(w - self.crop_size[1], 0)
        offset_h = np.random.randint(0, margin_h + 1)
        offset_w = np.random.randint(0, margin_w + 1)
        return offset_h, offset_h + self.crop_size[0], offset_w, offset_w + self.crop_size[1]

    def accept(self, counts):
        """the reference's test on np.unique(crop, return_counts=True) (counts: 256-bin histogram)"""
        cnt = np.asarray(counts).copy()
        cnt[self.ignore_index] = 0
        cnt = cnt[cnt > 0]
        return len(cnt) > 1 and np.max(cnt) / np.sum(cnt) < self.cat_max_ratio

    def set_box(self, results, box):
        p = _params(results)
        y1, y2, x1, x2 = box
        p.oy, p.ox = y1, x1
        p.ch, p.cw = min(y2, p.RH) - y1, min(x2, p.RW) - x1      # numpy slicing clips at the image border
        results['img_shape'] = (p.ch, p.cw)

    def transform(self, results, hist_fn=None):
        p = _params(results)
        box = self.generate_crop_bbox(p.RH, p.RW)
        self.set_box(results, box)
        if self.cat_max_ratio < 1. and results.get('gt_seg_map') is not None:
            assert hist_fn is not None, 'cat_max_ratio < 1 needs the crop histogram (use Compose.batch)'
            for _ in range(10):
                if self.accept(hist_fn(results)):
                    break
                box = self.generate_crop_bbox(p.RH, p.RW)
                self.set_box(results, box)
        return results


@register
class RandomFlip:
    """mmcv RandomFlip._choose_direction (one numpy.random.choice over [direction, None]) + mmseg's _flip"""

    def __init__(self, prob=None, direction='horizontal', swap_seg_labels=None):
        if isinstance(direction, (list, tuple)) or isinstance(prob, (list, tuple)) or direction != 'horizontal':
            raise NotImplementedError('the LED-Net configs flip horizontally with a scalar probability')
        if swap_seg_labels:
            raise NotImplementedError('swap_seg_labels')
        assert prob is None or 0 <= prob <= 1
        self.prob, self.direction = prob, direction

    def transform(self, results):
        p = _params(results)
        cur = None
        if self.prob is not None:
            cur = np.random.choice(np.array([self.direction, None], dtype=object), p=[self.prob, 1 - self.prob])
        if cur is None:
            results['flip'], results['flip_direction'] = False, None
        else:
            results['flip'], results['flip_direction'] = True, cur
            p.flip = 1
        return results


@register
class PhotoMetricDistortion:
    """transforms.py:583-750: the same numpy.random calls in the same order; values only recorded"""

    def __init__(self, brightness_delta=32, contrast_range=(0.5, 1.5), saturation_range=(0.5, 1.5), hue_delta=18):
        self.brightness_delta = brightness_delta
        self.contrast_lower, self.contrast_upper = contrast_range
        self.saturation_lower, self.saturation_upper = saturation_range
        self.hue_delta = hue_delta

    def _contrast(self, p):
        if np.random.randint(2):
            p.contrast = np.random.uniform(self.contrast_lower, self.contrast_upper)

    def transform(self, results):
        p = _params(results)
        if np.random.randint(2):
            p.bright = np.random.uniform(-self.brightness_delta, self.brightness_delta)
        p.mode = int(np.random.randint(2))
        if p.mode == 1:
            self._contrast(p)
        if np.random.randint(2):
            p.sat = np.random.uniform(self.saturation_lower, self.saturation_upper)
        if np.random.randint(2):
            p.hue = int(np.random.randint(-self.hue_delta, self.hue_delta))
        if p.mode == 0:
            self._contrast(p)
        return results


@register
class PackSegInputs:
    """formatting.py:14-107: which keys travel in the sample's metainfo"""

    def __init__(self, meta_keys=('img_path', 'seg_map_path', 'ori_shape', 'img_shape', 'pad_shape', 'scale_factor',
                                  'flip', 'flip_direction', 'reduce_zero_label')):
        self.meta_keys = meta_keys

    def transform(self, results):
        return results


_SKIPPED = ('LoadImageFromFile', 'LoadAnnotations')     # decoding stays on the host (out of scope)


def _entry(results):
    p = results['_aug']
    e = _lib.AugEntry()
    img, seg = results['img'], results.get('gt_seg_map')
    e.img, e.seg = img.data_ptr(), (seg.data_ptr() if seg is not None else None)
    e.H, e.W, e.RH, e.RW = p.H, p.W, p.RH, p.RW
    e.sx, e.sy = 1.0 / (float(p.RW) / p.W), 1.0 / (float(p.RH) / p.H)
    e.oy, e.ox, e.ch, e.cw, e.flip = p.oy, p.ox, p.ch, p.cw, p.flip
    e.bright_on, e.bright_beta = int(p.bright is not None), float(np.float32(p.bright or 0.0))
    e.contrast_mode = p.mode
    e.contrast_on, e.contrast_alpha = int(p.contrast is not None), float(np.float32(p.contrast or 1.0))
    e.sat_on, e.sat_alpha = int(p.sat is not None), float(np.float32(p.sat or 1.0))
    e.hue_on, e.hue_delta = int(p.hue is not None), int(p.hue or 0)
    return e


def _table(entries, device):
    host = (_lib.AugEntry * len(entries))(*entries)
    return torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(device)


class Compose:
    """The config's ``train_pipeline`` / ``test_pipeline`` list -> one GPU launch per batch."""

    def __init__(self, pipeline, pad_val=0, seg_pad_val=255):
        self.transforms = []
        for cfg in pipeline:
            cfg = dict(cfg)
            t = cfg.pop('type')
            if t in _SKIPPED:
                continue
            if t not in TRANSFORMS:
                raise KeyError(f'{t} is not in the GPU augmentation registry {sorted(TRANSFORMS)}')
            self.transforms.append(TRANSFORMS[t](**cfg))
        self.pad_val, self.seg_pad_val = pad_val, seg_pad_val
        self.pack = next((t for t in self.transforms if isinstance(t, PackSegInputs)), PackSegInputs())

    # -- RandomCrop's class-count test, for one sample (the reference's order of random draws is per sample) --------
    def _hist_fn(self, results):
        lib = _lib.get_lib()
        seg = results['gt_seg_map']
        tab = _table([_entry(results)], seg.device)
        hist = torch.zeros(256, dtype=torch.int32, device=seg.device)
        p = results['_aug']
        _run(lib, 'ledn_aug_crop_hist', seg, tab.data_ptr(), 1, p.ch * p.cw, hist.data_ptr())
        return hist.cpu().numpy()

    def batch(self, samples, out_hw=None):
        """samples: list of dict(img=uint8 H x W x 3 device tensor, gt_seg_map=uint8 H x W device tensor or None, ...)
        -> dict(inputs=[3 x h x w uint8 views], data_samples=[SegDataSample]); the views share one batch tensor
        (``out['batch']``, N x 3 x OH x OW, padded with pad_val / seg_pad_val); ``padded_samples`` = the samples as
        the data preprocessor would return them for ``batch`` (padded labels, img_shape / pad_shape / padding_size)."""
        lib = _lib.get_lib()
        dev = samples[0]['img'].device
        for r in samples:
            img, seg = r['img'], r.get('gt_seg_map')
            if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3 or not img.is_contiguous():
                raise _lib.LednError('augmentation: img must be a contiguous uint8 H x W x 3 tensor')
            if seg is not None and (seg.dtype != torch.uint8 or tuple(seg.shape) != tuple(img.shape[:2])
                                    or not seg.is_contiguous() or seg.device != img.device):
                raise _lib.LednError('augmentation: gt_seg_map must be a contiguous uint8 H x W tensor on the image\'s device')
            if img.device != dev or (lib.is_hip and not img.is_cuda):
                raise _lib.LednError('augmentation: all samples must live on the HIP device')
            r.pop('_aug', None)
            for t in self.transforms:               # per sample, in pipeline order: the reference's draw order
                if isinstance(t, RandomCrop):
                    t.transform(r, self._hist_fn)
                else:
                    t.transform(r)
            _params(r)
        if out_hw is None:
            out_hw = (max(r['_aug'].ch for r in samples), max(r['_aug'].cw for r in samples))
        OH, OW = out_hw
        if any(r['_aug'].ch > OH or r['_aug'].cw > OW for r in samples):
            raise _lib.LednError('augmentation: out_hw smaller than a sample')
        n = len(samples)
        has_seg = all(r.get('gt_seg_map') is not None for r in samples)
        batch = torch.empty((n, 3, OH, OW), dtype=torch.uint8, device=dev)
        labels = torch.empty((n, 1, OH, OW), dtype=torch.int64, device=dev) if has_seg else None
        tab = _table([_entry(r) for r in samples], dev)
        _run(lib, 'ledn_augment_batch', batch, tab.data_ptr(), n, batch.data_ptr(),
             labels.data_ptr() if has_seg else None, OH, OW, int(self.pad_val), int(self.seg_pad_val))
        inputs, data_samples, padded = [], [], []
        for i, r in enumerate(samples):
            p = r['_aug']
            inputs.append(batch[i, :, :p.ch, :p.cw])
            meta = {k: r[k] for k in self.pack.meta_keys if k in r}
            data_samples.append(SegDataSample(gt=labels[i, :, :p.ch, :p.cw] if has_seg else None, metainfo=meta))
            # the same sample as SegDataPreProcessor.forward(training=True) would hand it on (stack_batch's metainfo,
            # mmseg/utils/misc.py:77-117): the padded label and the padding extents, so that `batch` can go straight
            # into the model -- EncoderDecoder then makes the stem emit pad_val in the NORMALISED domain over the
            # padded area (normalise-then-pad, data_preprocessor.py:121-133) instead of normalising the raw fill
            padded.append(SegDataSample(gt=labels[i] if has_seg else None,
                                        metainfo=dict(meta, img_shape=(p.ch, p.cw), pad_shape=(OH, OW),
                                                      padding_size=(0, OW - p.cw, 0, OH - p.ch))))
        return dict(inputs=inputs, data_samples=data_samples, batch=batch, labels=labels, padded_samples=padded)
