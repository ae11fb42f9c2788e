"""LEDNet backbone (registered as ``type='LEDNet'``).

The reference withholds ``mmseg/models/backbones/lednet.py`` (8-line notice), so
this is the reconstruction described in SURVEY.md section 8 row a2-R and
DESIGN.md: DDRNet stem and bilateral skeleton (backbones/ddrnet.py:53-224),
SESP/CESPB stages (nn_layers/eesp.py:15-118), two GETB blocks
(backbones/UNetFormer_GETB.py:209-226), MFAF cross-branch fusion
(classification/model_utils.py:356-429), SEAM edge gate
(tools/speed/ddrnet_speed.py:282-338,388-389).  Its I/O contract is pinned by
LEDHead (decode_heads/led_head.py:62-81): train -> (c3, c5, x1, x2), eval ->
(c5, x1, x2), NCHW shapes, with x1: C @1/2, x2: C @1/4, c3: 2C @1/8, c5: 4C @1/8.

Returned tensors are NCHW *views* of NHWC storage (channels-last), so LEDHead
consumes them without a copy and foreign callers still see reference shapes.
"""
import math

import torch
import torch.nn as nn

from . import ops
from .blocks import (CESPB, GETB, MFAF, PPM, SEAM, SESP, BasicBlock, Block, ConvModule, kaiming_init)
from .ops import ACT_RELU, RES_ADD


def to_nhwc(t, dtype=None):
    """NCHW-shaped tensor -> dense NHWC tensor; zero-copy for channels-last views."""
    v = t.permute(0, 2, 3, 1)
    if v.is_contiguous() and (dtype is None or v.dtype == dtype):
        return v
    return ops.nchw_to_nhwc(t.contiguous(), dtype or t.dtype)


def to_nchw_view(t):
    return t.permute(0, 3, 1, 2)


import os as _os
from ._env import knob_int as _knob_int  # noqa: E402
SEAM_SLOT_EVAL = _knob_int('LEDN_SEAM_SLOT_EVAL', 2)   # inference: SEAM edge map on its own stream
FUSED_STEM = _knob_int('LEDN_FUSED_STEM', 1)        # first stem conv as ONE kernel from the planar batch (ledn_stem_conv): on since its register-direct form (stem_conv_reg_kernel, r03: 103 us against 142 us for im2col + GEMM at 8 x 1024^2; the LDS-window form of round 2 measured 182 vs 161 and was off)


class LEDNet(Block):
    def __init__(self, in_channels=3, channels=32, ppm_channels=128, norm_cfg=None,
                 align_corners=False, act_cfg=None, init_cfg=None, num_heads=8, window_size=8,
                 seam_percentile=0.8, act_dtype=torch.float32, cespb_depth=2, context_tail='getb',
                 seam_mode='percentile', seam_threshold=0.1, getb_stage3=True):
        """The first seven arguments are the reference config's (cfg :24-30).  The rest are the choices of the
        reconstruction, each with the documented default:
          num_heads, window_size  GETB attention (prototype: dim 128, 8 heads, 8x8 windows, ddrnet_speed.py:81-83)
          cespb_depth    SESP blocks cascaded per CESPB: int or (spatial branch, context branch).  The paper gives no
                         count (PDF p.17); the survey's contract fixes the default at 2.  (2, 3) is the
                         setting that reproduces the PUBLISHED PARAMETER COUNT: 1.6636 M against 1.661 M (+0.16 %;
                         2 gives 1.534 M = 92.3 %, 3 gives 1.669 M: tests/count_complexity.py).
                         The context branch needs >= 2 (growth and stride cannot share a block)
          context_tail   'getb' (default: 1x1 16C -> ppm_channels + GETB; PDF p.18 reports pooling pyramids hurt),
                         'pappm' or 'dappm' (utils/ppm.py: the DDRNet slot `ppm_channels` names, ddrnet.py:118-119)
          seam_mode      'percentile' (default, per-image seam_percentile: PDF section 4.2 eq.1) or 'fixed'
                         (the prototype's constant seam_threshold, ddrnet_speed.py:290-338)
          getb_stage3    GETB after the first context stage (2 GETBs in total, PDF p.17)"""
        super().__init__()
        if norm_cfg is not None and norm_cfg.get('type') not in ('BN', 'SyncBN'):
            raise ValueError(f'unsupported norm_cfg {norm_cfg}')
        if align_corners:
            raise NotImplementedError('align_corners=True is not used by any LED-Net config')
        if context_tail not in ('getb', 'pappm', 'dappm'):
            raise ValueError(f"context_tail must be 'getb', 'pappm' or 'dappm', got {context_tail!r}")
        if seam_mode not in ('percentile', 'fixed'):
            raise ValueError(f"seam_mode must be 'percentile' or 'fixed', got {seam_mode!r}")
        ds, dc = (cespb_depth, cespb_depth) if isinstance(cespb_depth, int) else tuple(cespb_depth)
        if dc < 2:
            raise ValueError('context-branch CESPBs need depth >= 2: SESP(stride=2, Spatial=False) requires '
                             'nIn == nOut (eesp.py:110-111), so growth and stride cannot share one block')
        C = channels
        self.in_channels, self.channels, self.ppm_channels = in_channels, C, ppm_channels
        self.cespb_depth, self.context_tail, self.seam_mode, self.getb_stage3 = (ds, dc), context_tail, seam_mode, getb_stage3
        self.sync_bn = bool(norm_cfg and norm_cfg.get('type') == 'SyncBN')
        self.act_dtype = act_dtype
        # stem: ddrnet.py:121-149 (Sequential indices 0,1,2,(3=ReLU),4,(5=ReLU) kept)
        self.stem = nn.ModuleDict({
            '0': ConvModule(in_channels, C, 3, 2, 1),
            '1': ConvModule(C, C, 3, 2, 1),
            '2': nn.Sequential(BasicBlock(C, C), BasicBlock(C, C, act_out=False)),
            '4': nn.Sequential(BasicBlock(C, 2 * C, 2, downsample=True), BasicBlock(2 * C, 2 * C, act_out=False)),
        })
        # spatial branch (names layer3_/layer4_/layer5_: tools/feature_map_visual.py:147, dsnet.py:70-72)
        self.layer3_ = CESPB(2 * C, 2 * C, 1, True, ds)
        self.layer4_ = CESPB(2 * C, 2 * C, 1, True, ds)
        self.layer5_ = SESP(2 * C, 4 * C, 1, 4, 7, True)
        # context branch
        self.layer3 = CESPB(2 * C, 4 * C, 2, False, dc)
        self.layer4 = CESPB(4 * C, 8 * C, 2, False, dc)
        self.layer5 = CESPB(8 * C, 16 * C, 2, False, dc)
        if getb_stage3:
            self.getb1 = GETB(4 * C, num_heads, window_size)
        if context_tail == 'getb':
            self.spp = ConvModule(16 * C, ppm_channels, 1)
            self.getb2 = GETB(ppm_channels, num_heads, window_size)
            assert ppm_channels == 4 * C, 'context tail width must equal the spatial tail (4C)'
        else:       # DAPPM(channels*16, ppm_channels, channels*4, num_scales=5): ddrnet.py:118-119
            self.spp = PPM(16 * C, ppm_channels, 4 * C, context_tail)
        # bilateral fusion (ddrnet.py:68-105)
        self.compression_1 = ConvModule(4 * C, 2 * C, 1, act=None)
        self.compression_2 = ConvModule(8 * C, 2 * C, 1, act=None)
        self.down_1 = ConvModule(2 * C, 4 * C, 3, 2, 1, act=None)
        self.down_2 = nn.Sequential(ConvModule(2 * C, 4 * C, 3, 2, 1), ConvModule(4 * C, 8 * C, 3, 2, 1, act=None))
        self.aff1 = MFAF(2 * C)
        self.aff2 = MFAF(2 * C)
        self.seam = SEAM(2 * C, seam_percentile if seam_mode == 'percentile' else None, seam_threshold)
        self.init_weights()

    def init_weights(self):
        kaiming_init(self)

    def _stem0(self, xin, patches=None):
        """first stem conv (3x3 s2, 3 -> C).  bf16: im2col to 32 columns + K=32 GEMM on the
        MFMA path (the 3-channel direct conv is VALU-bound); f32: direct kernel."""
        m = self.stem['0']
        if patches is None and (xin.dtype != torch.bfloat16 or 9 * self.in_channels > 32 or self.channels % 32):
            return m(xin)
        from .blocks import fold_bn
        w_eff = self.cached('stem_w', lambda: ops.stem_weight_as_1x1(m.conv.weight.detach()))
        wp = self.cached('stem_wp', lambda: ops.pack_conv_weights(w_eff, 0))
        s, b = self.cached('stem_fold', lambda: fold_bn(m.bn))
        if patches is None:
            patches = ops.im2col_stem(xin)
        return ops.conv2d(patches, w_eff, out_scale=s, out_shift=b, act=ACT_RELU, w_bf16=wp)

    # ------------------------------------------------------------------ #
    def forward(self, x, pre=None):
        """x: [N,3,H,W] float (normalised) or, with pre=(scale, shift, map[, valid_hw, pad_val]), raw
        uint8/float planes normalised -- and, outside each image's valid extent, padded -- on the fly
        (SegDataPreProcessor fusion: data_preprocessor.py:98-151)."""
        if self.training:
            from .train import lednet_forward_train
            return lednet_forward_train(self, x, pre)
        N, _, H, W = x.shape
        out_size = (math.ceil(H / 8), math.ceil(W / 8))                      # ddrnet.py:185
        s, b, m, valid, pad_val = (tuple(pre) + (None, 0.0))[:5] if pre is not None else (None, None, None, None, 0.0)
        if (self.act_dtype == torch.bfloat16 and self.in_channels == 3 and self.channels == 32 and FUSED_STEM
                and x.dtype in (torch.uint8, torch.float32, torch.bfloat16)):
            # normalisation + batch padding + im2col + K=32 GEMM + folded BatchNorm + ReLU: one kernel (ledn_stem_conv)
            from .blocks import fold_bn
            m0 = self.stem['0']
            wp = self.cached('stem_wp', lambda: ops.pack_conv_weights(ops.stem_weight_as_1x1(m0.conv.weight.detach()), 0))
            fs, fb = self.cached('stem_fold', lambda: fold_bn(m0.bn))
            x1 = ops.stem_conv(x.contiguous(), wp, s, b, m, valid, pad_val, out_scale=fs, out_shift=fb, act=ACT_RELU)
        elif self.act_dtype == torch.bfloat16 and 9 * self.in_channels <= 32 and self.channels % 32 == 0:
            # planar batch -> im2col patches in one kernel (normalisation folded in), then a K=32 GEMM
            x1 = self._stem0(None, ops.im2col_stem_planar(x.contiguous(), s, b, m, valid, pad_val))
        else:
            x1 = self._stem0(ops.nchw_to_nhwc(x.contiguous(), self.act_dtype, s, b, m, valid, pad_val))    # C  @1/2
        x2 = self.stem['1'](x1)                                                # C  @1/4
        y = self.stem['2'][1](self.stem['2'][0](x2), final_relu=True)
        y = self.stem['4'][1](self.stem['4'][0](y), final_relu=True)           # 2C @1/8
        # The context branch (1/16 .. 1/64 resolution: small launch-bound kernels) and the SEAM edge map
        # run on auxiliary streams between the bilateral fusion points (ops.Fork).
        with ops.Fork(y, SEAM_SLOT_EVAL) as fe:
            edge = self.seam.edge(y)
        # stage 3
        with ops.Fork(y, 1) as f3:
            x_c = self.layer3(y)                                               # 4C @1/16
            if self.getb_stage3:
                x_c = self.getb1(x_c)
            comp = ops.bilinear(self.compression_1(x_c, in_act=ACT_RELU), out_size)
        x_s = self.layer3_(y)
        f3.join(x_c, comp)
        x_c_r = self.down_1(x_s, in_act=ACT_RELU, res=x_c, res_mode=RES_ADD, act_override=ACT_RELU)
        x_s_r = self.aff1(x_s, comp, out_relu=True)                             # relu(x_s) for stage 4
        # stage 4
        with ops.Fork(x_c_r, 1) as f4:
            x_c = self.layer4(x_c_r)                                           # 8C @1/32
            comp = ops.bilinear(self.compression_2(x_c, in_act=ACT_RELU), out_size)
        x_s = self.layer4_(x_s_r)
        d = self.down_2[0](x_s, in_act=ACT_RELU)
        f4.join(x_c, comp)
        x_c_r = self.down_2[1](d, res=x_c, res_mode=RES_ADD, act_override=ACT_RELU)
        x_s = self.aff2(x_s, comp)
        fe.join(edge)
        x_s = self.seam.gate(edge, x_s)
        # stage 5
        with ops.Fork(x_c_r, 1) as f5:
            x_c = self.layer5(x_c_r)                                           # 16C @1/64
            x_c = self.getb2(self.spp(x_c)) if self.context_tail == 'getb' else self.spp(x_c)
        x_s = self.layer5_(x_s, in_relu=True)                                  # 4C @1/8
        f5.join(x_c)
        c5 = ops.bilinear(x_c, out_size, add=x_s)
        return tuple(to_nchw_view(t) for t in (c5, x1, x2))
