"""LEDHead (registered as ``type='LEDHead'``) and the multi-scale logit fusion.

Mirrors mmseg/models/decode_heads/led_head.py:15-146 and the LED-specific
``BaseDecodeHead.predict_by_feat`` (decode_heads/decode_head.py:341-379):
same constructor arguments, attribute names, state_dict keys
(``head.0.bn``, ``head.0.conv``, ``head.1``, ``conv_seg``, ``aux_cls_seg`` ...),
methods ``forward / loss / predict / loss_by_feat / predict_by_feat``.
"""
import math

import torch
import torch.nn as nn

from . import ops
from .blocks import Block, ConvModule, fold_bn, kaiming_init
from .lednet import to_nchw_view, to_nhwc
from .ops import ACT_RELU


class LEDHead(Block):
    def __init__(self, in_channels, channels, num_classes, norm_cfg=None, act_cfg=None,
                 dropout_ratio=0., align_corners=False, loss_decode=None, ignore_index=255,
                 out_channels=None, threshold=None, in_index=-1, input_transform=None,
                 sampler=None, conv_cfg=None, init_cfg=None):
        super().__init__()
        if dropout_ratio:
            raise NotImplementedError('LED-Net configs use dropout_ratio=0 (cfg :35)')
        if align_corners:
            raise NotImplementedError('align_corners=True is not used by any LED-Net config')
        if norm_cfg is not None and norm_cfg.get('type') not in ('BN', 'SyncBN'):
            raise ValueError(f'unsupported norm_cfg {norm_cfg}')
        self.in_channels, self.channels, self.num_classes = in_channels, channels, num_classes
        self.out_channels = out_channels or num_classes
        self.align_corners, self.ignore_index, self.threshold = align_corners, ignore_index, threshold
        self.sync_bn = bool(norm_cfg and norm_cfg.get('type') == 'SyncBN')
        self.head = self._make_base_head(in_channels, channels)
        self.aux_head = self._make_base_head(in_channels // 2, channels)
        # led_head.py:47-48 hard-codes (32, 2); identical for the shipped config
        self.head_x1 = self._make_base_head(32, 2)
        self.head_x2 = self._make_base_head(32, 2)
        self.conv_seg = nn.Conv2d(channels, self.out_channels, 1)
        self.aux_cls_seg = nn.Conv2d(channels, self.out_channels, 1)
        from .losses import build_loss
        if loss_decode is None:
            loss_decode = [dict(type='OhemCrossEntropy', thres=0.9, min_kept=131072, loss_weight=1.0),
                           dict(type='OhemCrossEntropy', thres=0.9, min_kept=131072, loss_weight=0.4)]
        if isinstance(loss_decode, dict):
            loss_decode = [loss_decode]
        self.loss_decode = nn.ModuleList(build_loss(c) for c in loss_decode)
        self.init_weights()

    @staticmethod
    def _make_base_head(cin, cout):
        """BN(in) -> ReLU -> conv3x3 -> BN -> ReLU  (led_head.py:84-99)."""
        return nn.Sequential(ConvModule(cin, cout, 3, 1, 1, act='relu', order=('norm', 'act', 'conv')),
                             nn.BatchNorm2d(cout), nn.ReLU())

    def init_weights(self):
        kaiming_init(self)

    # ------------------------------------------------------------------ #
    def _base_head_eval(self, seq, x, key, out_dtype=None):
        s, b = self.cached(key, lambda: fold_bn(seq[1]))
        return seq[0](x, post=(s, b, ACT_RELU), out_dtype=out_dtype)

    def forward_nhwc(self, inputs):
        """NHWC logits, f32.  eval: (x_c, head_x1, head_x2)  (led_head.py:76-81)."""
        if self.training:
            from .train import led_head_forward_train
            return led_head_forward_train(self, inputs)
        c5, x1, x2 = (to_nhwc(t) for t in inputs)
        f32 = torch.float32
        h1 = self._base_head_eval(self.head_x1, x1, 'h1', f32)
        h2 = self._base_head_eval(self.head_x2, x2, 'h2', f32)
        xc = self._base_head_eval(self.head, c5, 'h')
        xc = ops.conv2d(xc, self.conv_seg.weight, out_shift=self.conv_seg.bias, out_dtype=f32)
        return xc, h1, h2

    def forward(self, inputs):
        return tuple(to_nchw_view(t) for t in self.forward_nhwc(inputs))

    # ------------------------------------------------------------------ #
    @staticmethod
    def fuse_predict(xc, h1, h2, argmax=False):
        """decode_head.py:362-379 on NHWC f32 logits -> NCHW f32 logits (+ uint8 mask)."""
        size = (2 * h1.shape[1], 2 * h1.shape[2])
        r = ops.bilinear(xc, tuple(math.ceil(s / 4) for s in size), add=h2)
        r = ops.bilinear(r, tuple(math.ceil(s / 2) for s in size), add=h1)
        return ops.bilinear(r, size, nchw=True, argmax=argmax)

    def predict_by_feat(self, seg_logits, batch_img_metas=None):
        xc, h1, h2 = (to_nhwc(t, torch.float32) for t in seg_logits)
        return self.fuse_predict(xc, h1, h2)

    def predict(self, inputs, batch_img_metas=None, test_cfg=None):
        """decode_head.py:266-284 -> N x num_classes x H x W fused logits."""
        return self.fuse_predict(*self.forward_nhwc(inputs))

    def predict_nhwc(self, inputs):
        """fused logits in NHWC f32 (EncoderDecoder.postprocess_result crops / resizes them per image)"""
        xc, h1, h2 = self.forward_nhwc(inputs)
        size = (2 * h1.shape[1], 2 * h1.shape[2])
        r = ops.bilinear(xc, tuple(math.ceil(s / 4) for s in size), add=h2)
        r = ops.bilinear(r, tuple(math.ceil(s / 2) for s in size), add=h1)
        return ops.bilinear(r, size)

    def predict_with_mask(self, inputs):
        """fused logits and the first-max argmax mask (segmentors/base.py:188) in one pass."""
        return self.fuse_predict(*self.forward_nhwc(inputs), argmax=True)

    # ------------------------------------------------------------------ #
    def loss(self, inputs, batch_data_samples, train_cfg=None):
        """decode_head.py:248-264: forward, then loss_by_feat."""
        return self.loss_by_feat(self.forward(inputs), batch_data_samples)

    def loss_by_feat(self, seg_logits, batch_data_samples):
        """led_head.py:101-146: (context, spatial, head_x1, head_x2) logits (NCHW, as ``forward`` returns them in
        training) -> the two fused pyramids at the label size, OHEM-CE on each (loss_decode[0] / [1]) and the
        top-1 accuracy of the context logits; keys ``loss_context``, ``loss_spatial``, ``acc_seg``."""
        from .train import led_head_loss_by_feat
        return led_head_loss_by_feat(self, seg_logits, batch_data_samples)
