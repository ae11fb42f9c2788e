"""LED-Net building blocks on the HIP kernels.

Parameter containers are stock ``torch.nn`` modules (``nn.Conv2d``,
``nn.BatchNorm2d``, ``nn.PReLU``) whose ``forward`` is never called: they only
give every block the reference's ``state_dict`` key names, so a checkpoint
written by the reference's blocks (SESP, GETBBlock, Muti_AFF, ConvModule,
BasicBlock) loads unchanged.  All compute goes through :mod:`ops`.

Activations are NHWC tensors.  In eval mode BatchNorm is folded into the
producing kernel's epilogue (or the consumer's prologue for norm-act-conv
order); in training mode batch statistics are reduced by the producing kernel
and applied by a separate affine pass (see train.py).
"""
import math

import torch
import torch.nn as nn

from . import ops
from .ops import ACT_NONE, ACT_PRELU, ACT_RELU, ACT_RELU6, RES_ADD, RES_GATE, RES_NONE

_ACT = {None: ACT_NONE, 'relu': ACT_RELU, 'relu6': ACT_RELU6}


class Block(nn.Module):
    """Base: a per-module cache of derived tensors (folded BN, packed depthwise
    filters, gathered attention bias) valid while the parameters do not change,
    i.e. in eval mode; cleared by train()/eval(), load_state_dict() and .to()."""

    def __init__(self):
        super().__init__()
        self._cache = {}
        self._register_load_state_dict_pre_hook(lambda *a, **k: self._cache.clear())

    def train(self, mode=True):
        self._cache.clear()
        return super().train(mode)

    def _apply(self, fn, *a, **k):
        self._cache.clear()
        return super()._apply(fn, *a, **k)

    def cached(self, key, make):
        if self.training:
            return make()
        v = self._cache.get(key)
        if v is None:
            with torch.no_grad():
                v = self._cache[key] = make()
        return v


def fold_bn(bn, conv_bias=None):
    """eval-mode BatchNorm as (scale, shift), an optional preceding bias folded in."""
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    shift = bn.bias - bn.running_mean * scale
    if conv_bias is not None:
        shift = shift + conv_bias * scale
    return scale.contiguous(), shift.contiguous()


def packed(block, w, x, mode=0, groups=1):
    """bf16 weight pack for the MFMA conv path (None when the layer runs on the direct
    kernel): cached per block in eval mode, rebuilt per call in training."""
    if x.dtype != torch.bfloat16 or not ops.mfma_weight_ok(w, groups):
        return None
    return block.cached(('wp', id(w), mode), lambda: ops.pack_conv_weights(w.detach(), mode, groups))


def pack_dw(weights):
    """list of [n,1,KH,KW] depthwise filters -> [KH,KW,sum n] f32."""
    return torch.cat([w[:, 0].permute(1, 2, 0) for w in weights], dim=2).contiguous()


# --------------------------------------------------------------------------- #
class ConvModule(Block):
    """mmcv ConvModule semantics (conv / bn / activate in `order`), child names
    `conv`, `bn` as in the reference's checkpoints.
    Call sites: basic_block.py:43-57, ddrnet.py:68-105,123-138, led_head.py:87-94."""

    def __init__(self, cin, cout, k, stride=1, padding=0, act='relu', with_norm=True,
                 order=('conv', 'norm', 'act'), bias='auto', groups=1):
        super().__init__()
        if bias == 'auto':
            bias = not with_norm
        self.conv = nn.Conv2d(cin, cout, k, stride, padding, groups=groups, bias=bias)
        self.groups = groups
        self.norm_first = order.index('norm') < order.index('conv')
        if with_norm:
            self.bn = nn.BatchNorm2d(cin if self.norm_first else cout)
        self.with_norm = with_norm
        self.act = act
        self.stride, self.padding = stride, padding

    def forward(self, x, *, in_act=ACT_NONE, res=None, res_mode=RES_NONE, act_override=None,
                out_dtype=None, xadd=None, post=None):
        """post=(scale, shift, act): an extra BatchNorm+activation that FOLLOWS a
        norm->act->conv module (LEDHead._make_base_head), fused as the epilogue."""
        assert not self.training, 'training path: see train.py'
        act = _ACT[self.act] if act_override is None else act_override
        w = self.conv.weight
        if self.norm_first:
            # norm -> act -> conv: BN+act become the conv's input prologue
            s, b = self.cached('fold', lambda: fold_bn(self.bn))
            ps, pb, pact = post if post is not None else (None, self.conv.bias, ACT_NONE)
            return ops.conv2d(x, w, stride=self.stride, pad=self.padding, groups=self.groups, in_scale=s, in_shift=b,
                              in_act=act, out_scale=ps, out_shift=pb, act=pact, out_dtype=out_dtype, res=res,
                              res_mode=res_mode, w_bf16=packed(self, w, x, 0, self.groups))
        if self.with_norm:
            s, b = self.cached('fold', lambda: fold_bn(self.bn, self.conv.bias))
        else:
            s, b = None, self.conv.bias
        return ops.conv2d(x, w, stride=self.stride, pad=self.padding, groups=self.groups, in_act=in_act, xadd=xadd,
                          out_scale=s, out_shift=b, act=act, res=res, res_mode=res_mode,
                          out_dtype=out_dtype, w_bf16=packed(self, w, x, 0, self.groups))


class BasicBlock(Block):
    """mmseg/models/utils/basic_block.py:13-75."""

    def __init__(self, cin, cout, stride=1, downsample=False, act_out=True):
        super().__init__()
        self.conv1 = ConvModule(cin, cout, 3, stride, 1, act='relu')
        self.conv2 = ConvModule(cout, cout, 3, 1, 1, act=None)
        if downsample:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False),
                                            nn.BatchNorm2d(cout))
        else:
            self.downsample = None
        self.act_out = act_out
        self.stride = stride

    def forward(self, x, final_relu=False):
        """final_relu: fuse the stage-level nn.ReLU that follows the last block
        (ddrnet.py:143-149) into this block's epilogue."""
        assert not self.training
        out = self.conv1(x)
        if self.downsample is not None:
            s, b = self.cached('ds', lambda: fold_bn(self.downsample[1]))
            wd = self.downsample[0].weight
            res = ops.conv2d(x, wd, stride=self.stride, out_scale=s, out_shift=b, w_bf16=packed(self, wd, x))
        else:
            res = x
        act = ACT_RELU if (self.act_out or final_relu) else ACT_NONE
        return self.conv2(out, res=res, res_mode=RES_ADD, act_override=act)


class Bottleneck(Block):
    """mmseg/models/utils/basic_block.py:156-221: 1x1 -> 3x3 (stride) -> 1x1 to channels * 2, shortcut (identity or a
    1x1 stride-s conv + BatchNorm, as ddrnet.py:186-206 builds it), optional output activation (default: none)."""
    expansion = 2

    def __init__(self, cin, channels, stride=1, downsample=False, act_out=False):
        super().__init__()
        self.conv1 = ConvModule(cin, channels, 1, 1, 0, act='relu')
        self.conv2 = ConvModule(channels, channels, 3, stride, 1, act='relu')
        self.conv3 = ConvModule(channels, channels * self.expansion, 1, 1, 0, act=None)
        if downsample:
            self.downsample = nn.Sequential(nn.Conv2d(cin, channels * self.expansion, 1, stride, bias=False),
                                            nn.BatchNorm2d(channels * self.expansion))
        else:
            self.downsample = None
        self.act_out = act_out
        self.stride = stride

    def forward(self, x, final_relu=False):
        assert not self.training
        out = self.conv2(self.conv1(x))
        if self.downsample is not None:
            s, b = self.cached('ds', lambda: fold_bn(self.downsample[1]))
            wd = self.downsample[0].weight
            res = ops.conv2d(x, wd, stride=self.stride, out_scale=s, out_shift=b, w_bf16=packed(self, wd, x))
        else:
            res = x
        act = ACT_RELU if (self.act_out or final_relu) else ACT_NONE
        return self.conv3(out, res=res, res_mode=RES_ADD, act_override=act)


# --------------------------------------------------------------------------- #
class _CBR(nn.Module):
    def __init__(self, cin, cout, groups, act=True):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 1, bias=False, groups=groups)
        self.bn = nn.BatchNorm2d(cout)
        if act:
            self.act = nn.PReLU(cout)


class _BR(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.bn = nn.BatchNorm2d(c)
        self.act = nn.PReLU(c)


class _CD(nn.Module):
    def __init__(self, n, stride, d):
        super().__init__()
        self.conv = nn.Conv2d(n, n, 3, stride=stride, padding=d, dilation=d, groups=n, bias=False)


def sesp_dilations(k=4, r_lim=7, spatial=True):
    """nn_layers/eesp.py:40-62."""
    if spatial:
        return [1] * k
    table = {3: 1, 5: 2, 7: 3, 9: 4, 11: 5, 13: 6, 15: 7, 17: 6, 19: 12, 21: 18, 23: 24}
    ks = sorted((3 + 2 * i) if (3 + 2 * i) <= r_lim else 3 for i in range(k))
    return [table[s] for s in ks]


class SESP(Block):
    """nn_layers/eesp.py:15-118 (SESPV2=True, down_method='esp', k=4).

    Kernel plan (eval): grouped 1x1 + BN + PReLU  ->  fused 4-branch dilated
    depthwise pyramid with HFF adds  ->  second depthwise pass (d+1) with the
    cat-BN + PReLU epilogue  ->  grouped 1x1 + BN + residual (+avg-pooled input
    for stride 2) + PReLU.  4 launches (5 when stride 2)."""

    def __init__(self, nIn, nOut, stride=1, k=4, r_lim=7, Spatial=True):
        super().__init__()
        assert k == 4 and nOut % k == 0
        n = nOut // k
        self.nIn, self.nOut, self.n, self.stride, self.spatial = nIn, nOut, n, stride, Spatial
        self.dil = sesp_dilations(k, r_lim, Spatial)
        self.proj_1x1 = _CBR(nIn, n, k)
        self.spp_dw = nn.ModuleList(_CD(n, stride, d) for d in self.dil)
        self.spp_dw_v2 = nn.ModuleList(_CD(n, 1, d + 1) for d in self.dil)
        self.conv_1x1_exp = _CBR(nOut, nOut, k, act=False)
        self.br_after_cat = _BR(nOut)
        self.module_act = nn.PReLU(nOut)

    def forward(self, x, in_relu=False):
        """in_relu: the block input is relu(x) (fused as the 1x1's prologue); only
        legal when no residual / avg-pool shortcut reads the input."""
        assert not self.training
        uses_input = (self.stride == 2 and not self.spatial) or (self.stride == 1 and self.nIn == self.nOut)
        assert not (in_relu and uses_input)
        s, b = self.cached('proj', lambda: fold_bn(self.proj_1x1.bn))
        wpj = self.proj_1x1.conv.weight
        o1 = ops.conv2d(x, wpj, groups=4, in_act=ACT_RELU if in_relu else ACT_NONE,
                        out_scale=s, out_shift=b, act=ACT_PRELU, slope=self.proj_1x1.act.weight,
                        w_bf16=packed(self, wpj, x, 0, 4))
        w1 = self.cached('dw1', lambda: torch.stack(
            [m.conv.weight[:, 0].permute(1, 2, 0) for m in self.spp_dw]).contiguous())
        p = ops.sesp_pyramid(o1, w1, self.dil, self.stride)
        w2 = self.cached('dw2', lambda: pack_dw([m.conv.weight for m in self.spp_dw_v2]))
        s, b = self.cached('cat', lambda: fold_bn(self.br_after_cat.bn))
        cat = ops.dwconv2d(p, w2, dil=[d + 1 for d in self.dil], group_size=self.n, out_scale=s,
                           out_shift=b, act=ACT_PRELU, slope=self.br_after_cat.act.weight)
        s, b = self.cached('exp', lambda: fold_bn(self.conv_1x1_exp.bn))
        w = self.conv_1x1_exp.conv.weight
        wp = packed(self, w, cat, 0, 4)
        if self.stride == 2 and not self.spatial:                       # eesp.py:110-111
            return ops.conv2d(cat, w, groups=4, out_scale=s, out_shift=b, res=ops.avgpool3x3s2(x),
                              res_mode=RES_ADD, w_bf16=wp)
        res = x if (self.stride == 1 and self.nIn == self.nOut) else None  # eesp.py:114-115
        return ops.conv2d(cat, w, groups=4, out_scale=s, out_shift=b, res=res, res_mode=RES_ADD,
                          act=ACT_PRELU, slope=self.module_act.weight, w_bf16=wp)


class CESPB(nn.Sequential):
    """Cascade of `depth` SESP blocks (PDF p.17 "stage-wise cascade of ESP blocks"; the count is not published:
    default 2).  Growth in the first (stride-1) block, the stride in the last one, because
    SESP(stride=2, Spatial=False) needs nIn == nOut (eesp.py:110-111)."""

    def __init__(self, nIn, nOut, stride, spatial, depth=2):
        r = 7 if spatial else 9
        if depth < 1:
            raise ValueError('CESPB depth must be >= 1')
        if depth == 1:
            if stride == 2 and not spatial and nIn != nOut:
                raise ValueError('a one-block CESPB cannot grow channels at stride 2: SESP(stride=2, Spatial=False) '
                                 'needs nIn == nOut (eesp.py:110-111); use depth >= 2 on the context branch')
            blocks = [SESP(nIn, nOut, stride, 4, r, spatial)]
        else:
            blocks = ([SESP(nIn, nOut, 1, 4, r, spatial)] + [SESP(nOut, nOut, 1, 4, r, spatial) for _ in range(depth - 2)]
                      + [SESP(nOut, nOut, stride, 4, r, spatial)])
        super().__init__(*blocks)


class PPM(Block):
    """DAPPM / PAPPM pooling pyramid (mmseg/models/utils/ppm.py:11-192) as the selectable context tail of LEDNet
    (the `ppm_channels` slot of the DDRNet skeleton, ddrnet.py:118-119).  Child names = the reference's
    (`scales.{i}[.1]`, `processes[.{i}]`, `compression`, `shortcut`, each a norm -> act -> conv ConvModule without
    bias), so a checkpoint of the reference modules loads unchanged."""
    NAC = ('norm', 'act', 'conv')

    def __init__(self, cin, branch, cout, kind='pappm', num_scales=5, kernel_sizes=(5, 9, 17), strides=(2, 4, 8),
                 paddings=(2, 4, 8)):
        super().__init__()
        if kind not in ('dappm', 'pappm'):
            raise ValueError(f'unknown pooling pyramid {kind!r}')
        self.kind, self.num_scales = kind, num_scales
        self.pools = list(zip(kernel_sizes, strides, paddings))[:num_scales - 2]
        cm = lambda ci, co, k, pad=0, g=1: ConvModule(ci, co, k, 1, pad, act='relu', order=self.NAC, bias=False, groups=g)  # noqa: E731
        scales = [cm(cin, branch, 1)]
        for k, st, pd in self.pools:
            scales.append(nn.Sequential(nn.AvgPool2d(k, st, pd), cm(cin, branch, 1)))
        scales.append(nn.Sequential(nn.AdaptiveAvgPool2d((1, 1)), cm(cin, branch, 1)))
        self.scales = nn.ModuleList(scales)
        if kind == 'dappm':
            self.processes = nn.ModuleList(cm(branch, branch, 3, 1) for _ in range(num_scales - 1))
        else:
            self.processes = cm(branch * (num_scales - 1), branch * (num_scales - 1), 3, 1, num_scales - 1)
        self.compression = cm(branch * num_scales, cout, 1)
        self.shortcut = cm(cin, cout, 1)

    def pooled(self, x, i):
        """input of scales[i] (i >= 1): the k/s/p average pool or, for the last scale, the global mean"""
        if i < self.num_scales - 1:
            return ops.avgpool2d(x, *self.pools[i - 1])
        g = ops.adaptive_avgpool(x, 1)
        return g if g.dtype == x.dtype else ops.affine_act(g, out_dtype=x.dtype)

    def forward(self, x):
        assert not self.training, 'training path: see train.ppm'
        hw = x.shape[1:3]
        x_ = self.scales[0](x)
        n = self.num_scales
        if self.kind == 'dappm':
            feats = [x_]
            for i in range(1, n):
                up = ops.bilinear(self.scales[i][1](self.pooled(x, i)), hw, add=feats[i - 1])
                feats.append(self.processes[i - 1](up))
            cat = torch.cat(feats, dim=-1)
        else:
            ups = [ops.bilinear(self.scales[i][1](self.pooled(x, i)), hw, add=x_) for i in range(1, n)]
            cat = torch.cat([x_, self.processes(torch.cat(ups, dim=-1))], dim=-1)
        # (torch.cat along the channels of <= 1/64-resolution maps: data movement of a few hundred KB)
        return self.compression(cat, res=self.shortcut(x), res_mode=RES_ADD)


# --------------------------------------------------------------------------- #
class _Seq(nn.Sequential):
    pass


class _GLA(Block):
    """GlobalLocalAttention parameters (UNetFormer_GETB.py:97-143)."""

    def __init__(self, dim, heads, ws):
        super().__init__()
        self.qkv = nn.Sequential(nn.Conv2d(dim, 3 * dim, 1, bias=False))
        self.proj = nn.Sequential(nn.Conv2d(dim, dim, ws, padding=(ws - 1) // 2, groups=dim, bias=False),
                                  nn.BatchNorm2d(dim), nn.Conv2d(dim, dim, 1, bias=False))
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) ** 2, heads))
        coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing='ij')).flatten(1)
        rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
        rel[:, :, 0] += ws - 1
        rel[:, :, 1] += ws - 1
        rel[:, :, 0] *= 2 * ws - 1
        self.register_buffer('relative_position_index', rel.sum(-1))
        nn.init.trunc_normal_(self.relative_position_bias_table, std=.02)

    def bias_t(self):
        """[heads][key j][query i] relative-position bias (:181-187)."""
        return ops.relpos_bias(self.relative_position_bias_table.detach(), self.relative_position_index)


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Conv2d(dim, hidden, 1)
        self.fc2 = nn.Conv2d(hidden, dim, 1)


class GETB(Block):
    """GETBBlock, UNetFormer_GETB.py:209-226 (drop_path = drop = 0)."""

    def __init__(self, dim=128, num_heads=8, window_size=8, mlp_ratio=4.):
        super().__init__()
        self.dim, self.heads, self.ws = dim, num_heads, window_size
        self.norm1 = nn.BatchNorm2d(dim)
        self.attn = _GLA(dim, num_heads, window_size)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))
        self.norm2 = nn.BatchNorm2d(dim)

    def forward(self, x):
        assert not self.training
        a = self.attn
        s1, b1 = self.cached('n1', lambda: fold_bn(self.norm1))
        n1 = ops.affine_act(x, s1, b1)                                   # norm1(x): also `local`
        qkv = ops.conv2d(n1, a.qkv[0].weight, w_bf16=packed(self, a.qkv[0].weight, n1))
        att = ops.window_attn(qkv, self.cached('bias', a.bias_t), self.heads, self.ws)
        mix = ops.getb_pool(att, n1, self.ws)
        sp, bp = self.cached('proj', lambda: fold_bn(a.proj[1]))
        wdw = self.cached('dw', lambda: pack_dw([a.proj[0].weight]))
        pj = ops.dwconv2d(mix, wdw, pad=(self.ws - 1) // 2, ext1=True, out_scale=sp, out_shift=bp)
        x1 = ops.conv2d(pj, a.proj[2].weight, res=x, res_mode=RES_ADD, w_bf16=packed(self, a.proj[2].weight, pj))
        s2, b2 = self.cached('n2', lambda: fold_bn(self.norm2))
        h = ops.conv2d(x1, self.mlp.fc1.weight, in_scale=s2, in_shift=b2, out_shift=self.mlp.fc1.bias,
                       act=ACT_RELU6, w_bf16=packed(self, self.mlp.fc1.weight, x1))
        return ops.conv2d(h, self.mlp.fc2.weight, out_shift=self.mlp.fc2.bias, res=x1, res_mode=RES_ADD,
                          w_bf16=packed(self, self.mlp.fc2.weight, h))


# --------------------------------------------------------------------------- #
def _aff_seq(c, inter, pool=None):
    layers = [] if pool is None else [nn.AdaptiveAvgPool2d(pool)]
    layers += [nn.Conv2d(c, inter, 1), nn.BatchNorm2d(inter), nn.ReLU(inplace=True),
               nn.Conv2d(inter, c, 1), nn.BatchNorm2d(c)]
    return nn.Sequential(*layers)


import os as _os
from ._env import knob_int as _knob_int  # noqa: E402
FUSE_MFAF_CTX = _knob_int('LEDN_FUSE_MFAF_CTX', 1)


class MFAF(Block):
    """Muti_AFF, classification/model_utils.py:356-429."""

    POOLS = (('context1', 4), ('context2', 8), ('context3', 16), ('global_att', 1))

    def __init__(self, channels=64, r=4):
        super().__init__()
        inter = channels // r
        self.local_att = _aff_seq(channels, inter)
        self.context1 = _aff_seq(channels, inter, (4, 4))
        self.context2 = _aff_seq(channels, inter, (8, 8))
        self.context3 = _aff_seq(channels, inter, (16, 16))
        self.global_att = _aff_seq(channels, inter, 1)

    def _mlp(self, seq, off, x, xadd=None, out_dtype=None):
        """conv(+bias)+BN+ReLU -> conv(+bias); the trailing BN is returned as an
        affine pair for the gate kernel."""
        c0, bn0, c1, bn1 = seq[off], seq[off + 1], seq[off + 3], seq[off + 4]
        key = f'{id(seq)}'
        s0, b0 = self.cached(key + 'a', lambda: fold_bn(bn0, c0.bias))
        mid = ops.conv2d(x, c0.weight, xadd=xadd, out_scale=s0, out_shift=b0, act=ACT_RELU)
        out = ops.conv2d(mid, c1.weight, out_shift=c1.bias, out_dtype=out_dtype)
        return out, self.cached(key + 'b', lambda: fold_bn(bn1))

    def forward(self, x, r, out_relu=False):
        assert not self.training
        # pooled-context chains (pool + two tiny convs each) on auxiliary streams, the local branch on the
        # main one (see train.mfaf)
        forks, ctx, affs_ctx = [], [], []
        seqs = [getattr(self, name) for name, _ in self.POOLS]
        if (FUSE_MFAF_CTX and x.shape[-1] == 64 and all(sq[1].out_channels == 16 and sq[4].out_channels == 64 for sq in seqs)):
            # the four pooled-context MLPs (two tiny convs each, BatchNorm on the running statistics) in one launch
            # sequence on ONE auxiliary stream (ledn_mfaf_ctx_fwd) instead of 8 launches on four streams
            from . import ops_train as T
            f = ops.Fork(x, 3, r)
            with f:
                pooled = list(ops.multi_pool(x, [S for _, S in self.POOLS], xadd=r))
                ctx, _ = T.mfaf_ctx_fwd(pooled, [(sq[1], sq[2], sq[4]) for sq in seqs], False)
            affs_ctx = [self.cached(f'{id(sq)}b', lambda sq=sq: fold_bn(sq[5])) for sq in seqs]
            forks = [(f, tuple(ctx))]
        else:
            for idx, (name, S) in enumerate(self.POOLS):
                f = ops.Fork(x, 3 + idx, r)
                with f:
                    pooled = ops.adaptive_avgpool(x, S, xadd=r)
                    c, aff = self._mlp(getattr(self, name), 1, pooled)
                forks.append((f, c))
                ctx.append(c)
                affs_ctx.append(aff)
        xl, aff_l = self._mlp(self.local_att, 0, x, xadd=r)
        for f, c in forks:
            f.join(*(c if isinstance(c, tuple) else (c,)))
        affs = [aff_l] + affs_ctx
        return ops.mfaf_gate(x, r, xl, ctx, affs, act=ACT_RELU if out_relu else ACT_NONE)


class SEAM(Block):
    """Edge-attention gate (prototype tools/speed/ddrnet_speed.py:88-113,282-338,
    388-389; percentile rule PDF section 4.2)."""

    def __init__(self, channels=64, percentile=0.8, fixed_threshold=0.1):
        super().__init__()
        self.conv_1 = ConvModule(channels, 1, 3, 1, 1, act=None)
        self.conv_2 = ConvModule(1, channels, 3, 1, 1, act=None)
        self.percentile = percentile
        self.fixed_threshold = fixed_threshold

    def edge(self, feat):
        seg = self.conv_1(feat, out_dtype=torch.float32)
        return ops.seam_edge(seg, self.percentile, self.fixed_threshold, 0.1)

    def gate(self, edge, x_s):
        """x_s = conv_2(edge) * x_s + x_s."""
        return self.conv_2(edge, res=x_s, res_mode=RES_GATE, out_dtype=x_s.dtype)


def kaiming_init(module):
    """LEDHead.init_weights (led_head.py:53-60): Kaiming-normal fan_out/relu for
    every Conv2d, BN gamma=1 beta=0."""
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)
