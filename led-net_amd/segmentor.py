"""EncoderDecoder-compatible segmentor (``type='EncoderDecoder'``).

Reproduces the mode dispatch of mmseg/models/segmentors/base.py:60-101 and
encoder_decoder.py:117-143,187-239 for the LED-Net path: ``extract_feat``,
``encode_decode``, ``loss``, ``predict``, ``_forward``; the
``SegDataPreProcessor`` normalisation (data_preprocessor.py:98-133) is fused
into the stem's input transform.
"""
import types

import torch
import torch.nn as nn

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
        self.bgr_to_rgb = dp.get('bgr_to_rgb', False)
        mean, std = dp.get('mean'), dp.get('std')
        if mean is not None:
            m, s = torch.tensor(mean, dtype=torch.float32), torch.tensor(std, dtype=torch.float32)
            self.register_buffer('pre_scale', (1.0 / s).contiguous(), persistent=False)
            self.register_buffer('pre_shift', (-m / s).contiguous(), persistent=False)
            order = [2, 1, 0] if self.bgr_to_rgb else [0, 1, 2]
            self.register_buffer('pre_map', torch.tensor(order, dtype=torch.int32), persistent=False)
        else:
            self.pre_scale = self.pre_shift = self.pre_map = None

    def set_act_dtype(self, dtype):
        self.backbone.act_dtype = dtype
        return self

    # ---- encoder_decoder.py:117-132
    def _pre(self, inputs):
        if inputs.dtype == torch.uint8:
            if self.pre_scale is None:
                raise ValueError('uint8 input needs data_preprocessor mean/std')
            return (self.pre_scale, self.pre_shift, self.pre_map)
        return None

    def extract_feat(self, inputs):
        return self.backbone(inputs, self._pre(inputs))

    def encode_decode(self, inputs, batch_img_metas=None):
        return self.decode_head.predict(self.extract_feat(inputs), batch_img_metas, self.test_cfg)

    # ---- encoder_decoder.py:161-185
    def loss(self, inputs, data_samples):
        x = self.extract_feat(inputs)
        out = self.decode_head.loss(x, data_samples, self.train_cfg)
        return {'decode.' + k: v for k, v in out.items()}

    # ---- encoder_decoder.py:187-222 + base.py:127-200 (whole inference, no padding crop)
    def predict(self, inputs, data_samples=None):
        logits, mask = self.decode_head.predict_with_mask(self.extract_feat(inputs))
        if data_samples is None:
            data_samples = [SegDataSample() for _ in range(inputs.shape[0])]
        for i, ds in enumerate(data_samples):
            ds.seg_logits = PixelData(data=logits[i])
            ds.pred_sem_seg = PixelData(data=mask[i:i + 1])
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
