"""EncoderDecoder-compatible segmentor (``type='EncoderDecoder'``) and its data preprocessor.

Reproduces the mode dispatch of mmseg/models/segmentors/base.py:60-101 and
encoder_decoder.py:117-143,187-239 for the LED-Net path: ``extract_feat``,
``encode_decode``, ``loss``, ``predict``, ``_forward``, and
``postprocess_result`` (base.py:127-200: un-pad, flip back, resize to
``ori_shape``, argmax).  ``SegDataPreProcessor`` (data_preprocessor.py:98-151 +
utils/misc.py:30-128 ``stack_batch``) pads and stacks the batch; its
normalisation, BGR->RGB flip and the value of the padded area (``pad_val``,
applied AFTER the normalisation as in the reference) are fused into the stem's
input transform kernel, which reads the raw uint8 planes.
"""
import math
import types

import torch
import torch.nn as nn

from . import ops
from .registry import MODELS


class PixelData(types.SimpleNamespace):
    pass


class SegDataSample(types.SimpleNamespace):
    """Just enough of mmseg.structures.SegDataSample: ``gt_sem_seg.data`` (1xHxW
    int64), ``metainfo``, and the prediction fields written by ``predict``."""

    def __init__(self, gt=None, metainfo=None):
        super().__init__()
        self.metainfo = metainfo or {}
        if gt is not None:
            self.gt_sem_seg = PixelData(data=gt)


def _pad_to(t, hw, value):
    """right/bottom padding of a [..., h, w] tensor to hw (stack_batch's padding mode, misc.py:79-93)"""
    h, w = t.shape[-2:]
    if (h, w) == tuple(hw):
        return t
    out = t.new_full(tuple(t.shape[:-2]) + tuple(hw), value)
    out[..., :h, :w] = t
    return out


@MODELS.register_module()
class SegDataPreProcessor(nn.Module):
    """mmseg/models/data_preprocessor.py:17-151 for the LED-Net path.

    ``forward(data, training)`` takes ``dict(inputs=[C x h x w tensors] | N x C x H x W tensor, data_samples=[...])``
    and returns ``dict(inputs=N x C x H x W RAW batch, data_samples=...)``: the images are padded (right/bottom) to
    ``size`` / a multiple of ``size_divisor`` (training) or ``test_cfg['size' | 'size_divisor']`` (testing) and
    stacked; the labels are padded with ``seg_pad_val``; ``metainfo`` gets ``img_shape`` / ``pad_shape`` /
    ``padding_size`` (training) or ``img_padding_size`` (testing), as ``stack_batch`` writes them.  The batch
    stays RAW (uint8 or unnormalised float): ``EncoderDecoder`` hands ``(1/std, -mean/std, channel order,
    valid extents, pad_val)`` to the stem's input kernel, which normalises on the fly and emits ``pad_val`` in
    the padded area -- numerically the reference's normalise-then-pad."""

    def __init__(self, mean=None, std=None, size=None, size_divisor=None, pad_val=0, seg_pad_val=255,
                 bgr_to_rgb=False, rgb_to_bgr=False, batch_augments=None, test_cfg=None):
        super().__init__()
        if batch_augments:
            raise NotImplementedError('batch_augments are not used by the LED-Net config')
        assert not (bgr_to_rgb and rgb_to_bgr), '`bgr2rgb` and `rgb2bgr` cannot be set to True at the same time'
        self.size, self.size_divisor = size, size_divisor
        self.pad_val, self.seg_pad_val = pad_val, seg_pad_val
        self.channel_conversion = bool(bgr_to_rgb or rgb_to_bgr)
        self.test_cfg = test_cfg
        if mean is not None:
            assert std is not None, 'To enable the normalization in preprocessing, please specify both `mean` and `std`.'
            m, sd = torch.tensor(mean, dtype=torch.float32), torch.tensor(std, dtype=torch.float32)
            self.register_buffer('scale', (1.0 / sd).contiguous(), persistent=False)
            self.register_buffer('shift', (-m / sd).contiguous(), persistent=False)
            order = [2, 1, 0] if self.channel_conversion else [0, 1, 2]
            self.register_buffer('chan_map', torch.tensor(order[:len(mean)], dtype=torch.int32), persistent=False)
        else:
            self.scale = self.shift = self.chan_map = None

    def forward(self, data, training=False):
        inputs = data['inputs']
        samples = data.get('data_samples')
        if torch.is_tensor(inputs):
            inputs = list(inputs)
        assert len({t.dim() for t in inputs}) == 1 and inputs[0].dim() == 3, 'inputs: C x H x W tensors expected'
        dev = self.scale.device if self.scale is not None else inputs[0].device
        sizes = [tuple(t.shape[-2:]) for t in inputs]
        if training:
            assert samples is not None, 'During training, `data_samples` must be define.'
            size, div = self.size, self.size_divisor
            assert (size is not None) ^ (div is not None), 'only one of size and size_divisor should be valid'
        else:
            assert all(sz == sizes[0] for sz in sizes), 'The image size in a batch should be the same.'
            size = (self.test_cfg or {}).get('size')
            div = (self.test_cfg or {}).get('size_divisor')
        mh, mw = max(sz[0] for sz in sizes), max(sz[1] for sz in sizes)
        if div is not None and div > 1:
            mh, mw = (mh + div - 1) // div * div, (mw + div - 1) // div * div
        out_hw = []
        for h, w in sizes:          # misc.py:79-89: per image max(size - shape, 0); all must agree to stack
            if size is not None:
                out_hw.append((max(size[-2], h), max(size[-1], w)))
            elif div is not None:
                out_hw.append((max(mh, h), max(mw, w)))
            else:
                out_hw.append((h, w))
        assert all(hw == out_hw[0] for hw in out_hw), 'padded images differ in size and cannot be stacked'
        batch = torch.stack([_pad_to(t.to(dev), out_hw[0], 0) for t in inputs])
        if samples is not None:
            for ds, (h, w) in zip(samples, sizes):
                pad = (0, out_hw[0][1] - w, 0, out_hw[0][0] - h)      # left, right, top, bottom
                if training:
                    if hasattr(ds, 'gt_sem_seg'):
                        ds.gt_sem_seg.data = _pad_to(ds.gt_sem_seg.data.to(dev), out_hw[0], self.seg_pad_val)
                    ds.metainfo.update(img_shape=(h, w), pad_shape=out_hw[0], padding_size=pad)
                elif size is not None or div is not None:
                    ds.metainfo.update(img_padding_size=pad)
        return dict(inputs=batch, data_samples=samples)


@MODELS.register_module()
class EncoderDecoder(nn.Module):
    def __init__(self, backbone, decode_head, data_preprocessor=None, train_cfg=None, test_cfg=None,
                 neck=None, auxiliary_head=None, pretrained=None, init_cfg=None):
        super().__init__()
        if neck is not None or auxiliary_head is not None:
            raise NotImplementedError('LED-Net uses neither a neck nor an auxiliary head')
        self.backbone = MODELS.build(backbone)
        self.decode_head = MODELS.build(decode_head)
        self.align_corners = self.decode_head.align_corners
        self.num_classes = self.decode_head.num_classes
        self.out_channels = self.decode_head.out_channels
        self.train_cfg, self.test_cfg = train_cfg, test_cfg or dict(mode='whole')
        dp = dict(data_preprocessor or {})
        dp.pop('type', None)
        self.data_preprocessor = SegDataPreProcessor(**dp)
        self.bgr_to_rgb = self.data_preprocessor.channel_conversion
        # Trainer.capture(): a resident int32 [N, 2] buffer of the per-image valid extents.  While set, _pre hands THIS
        # buffer to the stem kernel (and never builds one on the fly): the captured hipGraph reads the extents of the
        # batch being replayed, which Trainer.replay() writes here -- not the frozen extents of the capture batch.
        self._valid_static = None
        self._valid_for_ptr = None          # address of the Trainer's static input batch the buffer belongs to

    # the normalisation constants handed to the stem's input kernel (kept under their first-round names)
    @property
    def pre_scale(self):
        return self.data_preprocessor.scale

    @property
    def pre_shift(self):
        return self.data_preprocessor.shift

    @property
    def pre_map(self):
        return self.data_preprocessor.chan_map

    def set_act_dtype(self, dtype):
        self.backbone.act_dtype = dtype
        return self

    # ---- encoder_decoder.py:117-132
    @staticmethod
    def _padding(ds):
        """(left, right, top, bottom) batch padding of one sample (base.py:160-165)"""
        meta = getattr(ds, 'metainfo', None) or {}
        pad = meta.get('img_padding_size', meta.get('padding_size'))
        return tuple(int(v) for v in pad) if pad is not None else (0, 0, 0, 0)

    def _pre(self, inputs, data_samples=None):
        """RAW uint8 batches (and padded RAW float batches) are normalised by the stem's input kernel."""
        if (self._valid_static is not None and inputs.data_ptr() == self._valid_for_ptr
                and self.pre_scale is not None and inputs.dtype == torch.uint8):
            assert self._valid_static.shape[0] == inputs.shape[0]
            return (self.pre_scale, self.pre_shift, self.pre_map, self._valid_static, float(self.data_preprocessor.pad_val))
        padded = data_samples is not None and any(any(self._padding(ds)) for ds in data_samples)
        if inputs.dtype != torch.uint8 and not padded:
            return None                  # already normalised float input (the first-round / test contract)
        if inputs.dtype == torch.uint8 and self.pre_scale is None:
            raise ValueError('uint8 input needs data_preprocessor mean/std')
        if not padded:
            return (self.pre_scale, self.pre_shift, self.pre_map)
        valid = self.valid_extents(inputs.shape[2:], data_samples).to(inputs.device)
        return (self.pre_scale, self.pre_shift, self.pre_map, valid, float(self.data_preprocessor.pad_val))

    def valid_extents(self, hw, data_samples, n=None):
        """int32 [N, 2] (host): rows / columns of each image that hold data (the rest is batch padding)"""
        H, W = hw
        valid = []
        for ds in (data_samples or []):
            left, right, top, bottom = self._padding(ds)
            if left or top:
                raise NotImplementedError('stack_batch pads right/bottom only (misc.py:83,87)')
            valid.append((H - bottom, W - right))
        if not valid:
            valid = [(H, W)] * int(n)
        return torch.tensor(valid, dtype=torch.int32)

    def extract_feat(self, inputs, data_samples=None):
        return self.backbone(inputs, self._pre(inputs, data_samples))

    def encode_decode(self, inputs, batch_img_metas=None):
        return self.decode_head.predict(self.extract_feat(inputs), batch_img_metas, self.test_cfg)

    # ---- encoder_decoder.py:161-185
    def loss(self, inputs, data_samples):
        x = self.extract_feat(inputs, data_samples)
        out = self.decode_head.loss(x, data_samples, self.train_cfg)
        return {'decode.' + k: v for k, v in out.items()}

    # ---- encoder_decoder.py:187-222 (whole inference) + base.py:127-200
    def predict(self, inputs, data_samples=None):
        feats = self.extract_feat(inputs, data_samples)
        H, W = inputs.shape[2:]
        plain = data_samples is None or all(
            not any(self._padding(ds)) and not ds.metainfo.get('flip')
            and tuple(ds.metainfo.get('ori_shape', (H, W))[:2]) == (H, W) for ds in data_samples)
        if plain:       # nothing to crop / flip / resize: fused logits and first-max argmax in ONE pass
            logits, mask = self.decode_head.predict_with_mask(feats)
            if data_samples is None:
                data_samples = [SegDataSample() for _ in range(inputs.shape[0])]
            for i, ds in enumerate(data_samples):
                ds.seg_logits = PixelData(data=logits[i])
                ds.pred_sem_seg = PixelData(data=mask[i:i + 1])
            return data_samples
        return self.postprocess_result(self.decode_head.predict_nhwc(feats), data_samples)

    def postprocess_result(self, seg_logits, data_samples=None):
        """base.py:127-200 on the fused NHWC f32 logits: per image remove the batch padding, undo the test-time
        flip, resize to ``ori_shape`` (bilinear, align_corners=False) and take the first-max argmax -- resize and
        argmax in one kernel (ledn_bilinear(argmax)); ``seg_logits`` N x C x H x W tensors are accepted too."""
        if seg_logits.dim() == 4 and seg_logits.shape[1] == self.out_channels and seg_logits.shape[-1] != self.out_channels:
            seg_logits = seg_logits.permute(0, 2, 3, 1).contiguous()          # NCHW given: as the reference's
        N, H, W, Cc = seg_logits.shape
        if Cc == 1:
            raise NotImplementedError('binary (sigmoid / threshold) heads are not used by the LED-Net config')
        if data_samples is None:
            data_samples = [SegDataSample() for _ in range(N)]
        for i, ds in enumerate(data_samples):
            meta = ds.metainfo
            left, right, top, bottom = self._padding(ds)
            lg = seg_logits[i:i + 1, top:H - bottom, left:W - right, :]
            flip = meta.get('flip')
            if flip:
                direction = meta.get('flip_direction')
                assert direction in ('horizontal', 'vertical')
                lg = lg.flip(dims=(2,) if direction == 'horizontal' else (1,))
            size = tuple(meta.get('ori_shape', lg.shape[1:3])[:2])
            out, mask = ops.bilinear(lg.contiguous(), size, nchw=True, argmax=True)
            ds.seg_logits = PixelData(data=out[0])
            ds.pred_sem_seg = PixelData(data=mask)
        return data_samples

    def _forward(self, inputs, data_samples=None):
        return self.decode_head.forward(self.extract_feat(inputs))

    def forward(self, inputs, data_samples=None, mode='tensor'):
        if mode == 'loss':
            return self.loss(inputs, data_samples)
        if mode == 'predict':
            return self.predict(inputs, data_samples)
        if mode == 'tensor':
            return self._forward(inputs, data_samples)
        raise RuntimeError(f'Invalid mode "{mode}". Only supports loss, predict and tensor mode')
