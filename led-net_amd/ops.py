"""Tensor-level wrappers over the C ABI (include/ledn.h).

All activations are dense NHWC torch tensors (float32 or bfloat16); torch is
used for device memory and the current HIP stream only.  Every function checks
shapes/dtypes/devices before handing raw pointers to the library.
"""
import ctypes as C
import math

import torch

from . import _lib
from ._lib import (ACT_NONE, ACT_PRELU, ACT_RELU, ACT_RELU6, ACT_SIGMOID, BF16, F32,  # noqa: F401
                   RES_ADD, RES_GATE, RES_NONE, LednError)

_DT = {torch.float32: F32, torch.bfloat16: BF16}


def _dt(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise LednError(f'unsupported activation dtype {t.dtype}')


def _check(lib, *tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_contiguous():
            raise LednError('non-contiguous tensor passed to a ledn op')
        if lib.is_hip and not t.is_cuda:
            raise LednError('ledn ops run on the HIP device only (tensor is on %s); '
                            'there is no CPU fallback' % t.device)
        if not lib.is_hip and t.is_cuda:
            raise LednError('emulation library bound but tensor is on the GPU')


def _p(t):
    return None if t is None else t.data_ptr()


def _f32(t, n=None):
    if t is None:
        return None
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise LednError('parameter tensors must be contiguous float32')
    if n is not None and t.numel() != n:
        raise LednError(f'parameter length {t.numel()} != {n}')
    return t


def _stream(lib, ref):
    if lib.is_hip:
        return torch.cuda.current_stream(ref.device).cuda_stream
    return None


# ---- optional per-launch timing with HIP events (bench.py roofline leg) -------
_TIMING = None


def start_timing():
    """Record a pair of HIP events (on the launch stream) around every kernel launch."""
    global _TIMING
    _TIMING = []


def stop_timing():
    """-> list of dicts {entry, sig, bytes, flops, kernel, ms}; call after a device sync."""
    global _TIMING
    rec, _TIMING = _TIMING or [], None
    return [dict(entry=n, sig=w[0], bytes=w[1], flops=w[2], kernel=w[3] if len(w) > 3 else n[5:] + '_kernel',
                 ms=e0.elapsed_time(e1)) for n, w, e0, e1 in rec]


# ---- zero arena: every small zero-initialised f32 scratch of a training step (BatchNorm statistics,
# reduction targets) is a slice of ONE buffer cleared by ONE fill at the start of the step, instead
# of a torch.zeros (= one fill kernel) each: ~550 fills per step otherwise.
class ZeroArena:
    def __init__(self, device, nfloats=4 << 20):
        self.buf = torch.empty(nfloats, dtype=torch.float32, device=device)
        self.off = nfloats      # nothing available until reset()

    def reset(self):
        self.buf.zero_()
        self.off = 0

    def take(self, shape):
        n = 1
        for k in shape:
            n *= int(k)
        end = self.off + ((n + 3) & ~3)     # 16-byte granules
        if n > (1 << 16) or end > self.buf.numel():
            return None
        t = self.buf[self.off:self.off + n].view(shape)
        self.off = end
        return t


_ARENA = None


def set_zero_arena(arena):
    """install (or remove, arena=None) the zero arena used by zeros_f32"""
    global _ARENA
    _ARENA = arena


def zeros_f32(shape, device):
    """zero-initialised f32 scratch: an arena slice when a training step installed one"""
    if isinstance(shape, int):
        shape = (shape,)
    if _ARENA is not None and _ARENA.buf.device == device:
        t = _ARENA.take(tuple(shape))
        if t is not None:
            return t
    return torch.zeros(tuple(shape), dtype=torch.float32, device=device)


def _nb(*ts):
    return sum(t.numel() * t.element_size() for t in ts if t is not None)


# ---- auxiliary HIP streams: LED-Net's two branches (and the SEAM edge map) are independent between
# the bilateral fusion points; the context branch is a chain of small launch/latency-bound kernels
# that leaves most CUs idle, so it runs concurrently with the HBM-bound spatial branch.  Inside a
# hipGraph capture the fork/join pairs become parallel branches of the graph; autograd replays every
# backward kernel on the stream of its forward, so the backward overlaps the same way.
_AUX = {}            # (device, slot) -> torch.cuda.Stream
MULTI_STREAM = True


def _aux_stream(device, slot):
    st = _AUX.get((device, slot))
    if st is None:
        st = _AUX[(device, slot)] = torch.cuda.Stream(device=device)
    return st


def _slot(ref):
    """workspace slot of the current stream of ref's device (0 = any stream that is not auxiliary)"""
    if _AUX:
        sid = torch.cuda.current_stream(ref.device).cuda_stream
        for (dev, slot), st in _AUX.items():
            if dev == ref.device and st.cuda_stream == sid:
                return slot
    return 0


class Fork:
    """``with Fork(ref, slot, *inputs) as f: ...`` runs the enclosed ops on auxiliary stream `slot`
    (after everything already queued on the current stream); ``f.join(*outputs)`` makes the
    current stream wait for it.  A no-op for host tensors (emulator) and when MULTI_STREAM is off."""

    def __init__(self, ref, slot=1, *inputs):
        self.on = bool(ref.is_cuda and MULTI_STREAM and slot > 0)     # slot 0 = stay on the current stream
        if self.on:
            self.main = torch.cuda.current_stream(ref.device)
            self.side = _aux_stream(ref.device, slot)
            self.inputs = (ref,) + inputs
            self.joined = False

    def __enter__(self):
        if self.on:
            self.side.wait_stream(self.main)
            for t in self.inputs:            # caching allocator: these blocks are also in use on `side`
                t.record_stream(self.side)
            self.ctx = torch.cuda.stream(self.side)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.ctx.__exit__(*exc)
        return False

    def join(self, *outputs):
        if self.on and not self.joined:
            self.main.wait_stream(self.side)
            for t in outputs:
                if t is not None:
                    t.record_stream(self.main)
            self.joined = True


def _run(lib, name, ref, *args, work=None):
    stream = _stream(lib, ref)
    lib.ensure_workspace(ref.device, _slot(ref) if lib.is_hip else 0, stream=stream)
    if _TIMING is not None and lib.is_hip:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.call(name, *args, stream)
        e1.record()
        _TIMING.append((name, work or ('', 0, 0), e0, e1))
    else:
        lib.call(name, *args, stream)


import os as _os
from ._env import knob_int as _knob_int  # noqa: E402
DEFER_STATS = _knob_int('LEDN_DEFER_STATS', 3)   # 0 off, 1 convolutions only, 3 all producers


def _run_stats(lib, name, ref, stats, defer, *args, work=None):
    """_run for a statistics producer; defer: its (sum, sumsq) rows stay in the workspace for the
    bn_finalize that follows immediately (PendingRows)"""
    if not (defer and stats is not None and (DEFER_STATS >= 3 or (DEFER_STATS >= 1 and name == 'ledn_conv2d'))):
        return _run(lib, name, ref, *args, work=work)
    lib.call('ledn_stats_defer_begin')
    try:
        _run(lib, name, ref, *args, work=work)
    finally:
        part, rows = C.c_void_p(), C.c_int(0)
        lib.call('ledn_stats_defer_end', C.byref(part), C.byref(rows))
    if rows.value > 0:
        PendingRows.put(stats[0], part.value, rows.value, lib)


def conv_out_size(h, k, stride, pad, dil):
    return (h + 2 * pad - ((k - 1) * dil + 1)) // stride + 1


def conv2d(x, w, *, stride=1, pad=0, dil=1, groups=1, xadd=None, in_scale=None, in_shift=None,
           in_act=ACT_NONE, in_slope=None, out_scale=None, out_shift=None, act=ACT_NONE, slope=None, res=None,
           res_mode=RES_NONE, stats=None, out_dtype=None, transposed=False, out_hw=None, w_bf16=None,
           defer_stats=False, _query=False):
    """Dense/grouped convolution.  x: [N,H,W,Cin]; w: OIHW f32 [Cout_f, Cin_f/groups, KH, KW].
    transposed=True: x is dz [N,Ho_f,Wo_f,Cout_f]; returns dx [N,*out_hw,Cin_f].
    w_bf16: pack_conv_weights(w, mode=int(transposed)) -- enables the MFMA path for bf16."""
    lib = _lib.get_lib()
    N, H, W, Cin = x.shape
    cof, cigf, KH, KW = w.shape
    _f32(w)
    d = _lib.ConvDesc()
    if not transposed:
        if Cin != cigf * groups:
            raise LednError(f'conv2d: Cin {Cin} != {cigf}*{groups}')
        Cout = cof
        Ho, Wo = conv_out_size(H, KH, stride, pad, dil), conv_out_size(W, KW, stride, pad, dil)
        d.ws_co, d.ws_ci, d.ws_tap = cigf * KH * KW, KH * KW, 1
    else:
        if Cin != cof:
            raise LednError(f'conv2d(transposed): dz channels {Cin} != {cof}')
        Cout = cigf * groups
        Ho, Wo = out_hw
        d.ws_co, d.ws_ci, d.ws_tap = KH * KW, cigf * KH * KW, 1
    y = torch.empty((N, Ho, Wo, Cout), dtype=out_dtype or x.dtype, device=x.device)
    if res is not None and (res.shape != y.shape or res.dtype != y.dtype):
        raise LednError('conv2d: residual shape/dtype mismatch')
    if xadd is not None and (xadd.shape != x.shape or xadd.dtype != x.dtype):
        raise LednError('conv2d: xadd shape/dtype mismatch')
    _check(lib, x, w, y, res, xadd, in_scale, in_shift, in_slope, out_scale, out_shift, slope, w_bf16)
    if in_act == ACT_PRELU and in_slope is None:
        raise LednError('conv2d: in_act PRELU needs in_slope')
    if w_bf16 is not None and (w_bf16.dtype != torch.bfloat16 or w_bf16.numel() != w.numel() * groups):
        raise LednError('conv2d: w_bf16 must be the bfloat16 pack of w')
    d.x, d.xadd, d.w, d.y, d.res, d.w_bf16 = _p(x), _p(xadd), _p(w), _p(y), _p(res), _p(w_bf16)
    d.in_scale, d.in_shift, d.in_slope = _p(_f32(in_scale, Cin)), _p(_f32(in_shift, Cin)), _p(_f32(in_slope, Cin))
    d.out_scale, d.out_shift = _p(_f32(out_scale, Cout)), _p(_f32(out_shift, Cout))
    d.slope = _p(_f32(slope, Cout))
    if stats is not None:
        _f32(stats[0], Cout), _f32(stats[1], Cout)
        _check(lib, stats[0], stats[1])
        d.stat_sum, d.stat_sqsum = _p(stats[0]), _p(stats[1])
    d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = N, H, W, Cin, Ho, Wo, Cout
    d.KH, d.KW, d.stride, d.pad, d.dil, d.groups = KH, KW, stride, pad, dil, groups
    d.in_act, d.act_out, d.res_mode = in_act, act, res_mode if res is not None else RES_NONE
    d.dtype_x, d.dtype_y, d.transposed = _dt(x), _dt(y), int(transposed)
    if _query:
        return int(lib.cdll.ledn_conv2d_uses_mfma(d))
    _run_stats(lib, 'ledn_conv2d', x, stats, defer_stats, d, work=_TIMING is not None and (
        f'conv{KH}x{KW}{"T" if transposed else ""} {Cin}->{Cout} g{groups} s{stride} {N}x{H}x{W} {str(x.dtype)[6:]}',
        _nb(x, xadd, y, res, w), 2 * N * Ho * Wo * Cout * (Cin // groups) * KH * KW,
        ('conv_direct_kernel', 'conv_mfma_kernel', 'conv1x1_mfma_kernel', 'conv3x3_reg_kernel', 'conv3x3_narrowin_mfma_kernel', 'conv_f32_mfma_kernel', 'head_fwd_kernel')[lib.cdll.ledn_conv2d_uses_mfma(d)]))
    return y


def conv2d_kernel_id(x, w, **kw):
    """which kernel ledn_conv2d runs these arguments on (no launch): 0 conv_direct_kernel (VALU), 1 conv_mfma_kernel,
    2 conv1x1_mfma_kernel, 3 conv3x3_reg_kernel, 4 conv3x3_narrowin_mfma_kernel, 5 conv_f32_mfma_kernel, 6 head_fwd_kernel"""
    return conv2d(x, w, _query=True, **kw)


def im2col_stem(x):
    """x: [N,H,W,3] bf16 -> [N,Ho,Wo,32] bf16 patches of the 3x3/s2/p1 stem conv."""
    lib = _lib.get_lib()
    N, H, W, Cc = x.shape
    if x.dtype != torch.bfloat16 or 9 * Cc > 32:
        raise LednError('im2col_stem: bf16 input with <= 3 channels required')
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    p = torch.empty((N, Ho, Wo, 32), dtype=torch.bfloat16, device=x.device)
    _check(lib, x, p)
    _run(lib, 'ledn_im2col_stem', x, _p(x), _p(p), N, H, W, Cc, Ho, Wo,
         work=_TIMING is not None and (f'im2col_stem {N}x{H}x{W}', _nb(x, p), 0))
    return p


def _valid_hw(valid, N, ref):
    if valid is None:
        return None
    if valid.dtype != torch.int32 or tuple(valid.shape) != (N, 2) or valid.device != ref.device:
        raise LednError('valid_hw must be an int32 [N, 2] tensor on the input\'s device')
    return valid


def im2col_stem_planar(x, scale=None, shift=None, chan_map=None, valid=None, pad_val=0.0):
    """x: [N,C,H,W] uint8/float32/bfloat16 planar batch -> [N,Ho,Wo,32] bf16 patches of the 3x3/s2/p1 stem
    conv with y = x[map[c]]*scale[c]+shift[c] applied (= im2col_stem(nchw_to_nhwc(x, bf16, ...))).
    valid: int32 [N,2] (rows, columns) holding image data; the rest is batch padding = pad_val."""
    lib = _lib.get_lib()
    N, Cc, H, W = x.shape
    dtx = {torch.float32: F32, torch.bfloat16: BF16, torch.uint8: _lib.U8}.get(x.dtype)
    if dtx is None or 9 * Cc > 32:
        raise LednError('im2col_stem_planar: uint8/f32/bf16 input with <= 3 channels required')
    if chan_map is not None and (chan_map.dtype != torch.int32 or chan_map.numel() != Cc):
        raise LednError('im2col_stem_planar: chan_map must be int32[C]')
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    p = torch.empty((N, Ho, Wo, 32), dtype=torch.bfloat16, device=x.device)
    _check(lib, x, p, scale, shift, chan_map, _valid_hw(valid, N, x))
    _run(lib, 'ledn_im2col_stem_planar', x, _p(x), dtx, _p(p), N, H, W, Cc, Ho, Wo, _p(_f32(scale, Cc)),
         _p(_f32(shift, Cc)), _p(chan_map), _p(valid), float(pad_val),
         work=_TIMING is not None and (f'im2col_stem_planar {N}x{H}x{W}', _nb(x, p), 0, 'im2col_stem_planar_kernel'))
    return p


def stem_conv(x, w_pack, scale=None, shift=None, chan_map=None, valid=None, pad_val=0.0, out_scale=None, out_shift=None,
              act=ACT_NONE, stats=None, defer_stats=False):
    """First stem convolution (3x3 s2, 3 -> 32) straight from the planar batch x [N,3,H,W] (uint8 / f32 / bf16):
    normalisation, channel map, batch padding, im2col and the K = 32 GEMM in one kernel (ledn_stem_conv).
    w_pack: pack_conv_weights(stem_weight_as_1x1(w), 0).  -> y [N,Ho,Wo,32] bf16."""
    lib = _lib.get_lib()
    N, Cc, H, W = x.shape
    if Cc != 3 or x.dtype not in (torch.uint8, torch.float32, torch.bfloat16):
        raise LednError('stem_conv: uint8/f32/bf16 input with 3 channels required')
    if chan_map is not None and (chan_map.dtype != torch.int32 or chan_map.numel() != Cc):
        raise LednError('stem_conv: chan_map must be int32[C]')
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if x.data_ptr() % 4:      # (a slice of a uint8 batch with an odd image size: the kernel reads 4-byte aligned windows)
        x = x.clone()
    y = torch.empty((N, Ho, Wo, 32), dtype=torch.bfloat16, device=x.device)
    _check(lib, x, w_pack, y, scale, shift, chan_map, valid, out_scale, out_shift)
    dtx = {torch.uint8: _lib.U8, torch.float32: F32, torch.bfloat16: BF16}[x.dtype]
    valid = _valid_hw(valid, N, x)
    s0 = s1 = None
    if stats is not None:
        _check(lib, stats[0], stats[1])
        s0, s1 = _p(_f32(stats[0], 32)), _p(_f32(stats[1], 32))
    _run_stats(lib, 'ledn_stem_conv', x, stats, defer_stats, _p(x), dtx, _p(w_pack), _p(y), N, H, W, Cc, Ho, Wo, 32,
               _p(_f32(scale, Cc)), _p(_f32(shift, Cc)), _p(chan_map), _p(valid), float(pad_val),
               _p(_f32(out_scale, 32)), _p(_f32(out_shift, 32)), int(act), s0, s1,
               work=_TIMING is not None and (f'stem_conv {N}x{H}x{W}', _nb(x, y), 2 * y.numel() * 27, 'stem_conv_reg_kernel'))
    return y


def stem_conv_wgrad(x, dz, dw, scale=None, shift=None, chan_map=None, valid=None, pad_val=0.0, bn_desc=None):
    """dw [32,3,3,3] f32 += weight gradient of stem_conv from the planar batch x [N,3,H,W] and dz [N,Ho,Wo,32] bf16
    (ledn_stem_conv_wgrad: no patch matrix).  -> dw
    bn_desc: a prepared _lib.BnBwdDesc whose reduce half has run -- dz is then the gradient of act(BN(z)) and the apply half
    happens inside the kernel (ledn_stem_conv_wgrad_bn)."""
    lib = _lib.get_lib()
    N, Cc, H, W = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if Cc != 3 or x.dtype not in (torch.uint8, torch.float32, torch.bfloat16):
        raise LednError('stem_conv_wgrad: uint8/f32/bf16 input with 3 channels required')
    if dz.dtype != torch.bfloat16 or tuple(dz.shape) != (N, Ho, Wo, 32) or not dz.is_contiguous():
        raise LednError('stem_conv_wgrad: dz must be contiguous bf16 [N,Ho,Wo,32]')
    if dw.dtype != torch.float32 or dw.numel() != 32 * 27 or not dw.is_contiguous():
        raise LednError('stem_conv_wgrad: dw must be contiguous f32 [32,3,3,3]')
    if x.data_ptr() % 4:
        x = x.clone()
    _check(lib, x, dz, dw, scale, shift, chan_map, valid)
    dtx = {torch.uint8: _lib.U8, torch.float32: F32, torch.bfloat16: BF16}[x.dtype]
    valid = _valid_hw(valid, N, x)
    if bn_desc is not None:
        _run(lib, 'ledn_stem_conv_wgrad_bn', x, _p(x), dtx, bn_desc, _p(dw), N, H, W, Cc, Ho, Wo, 32,
             _p(_f32(scale, Cc)), _p(_f32(shift, Cc)), _p(chan_map), _p(valid), float(pad_val),
             work=_TIMING is not None and (f'stem_conv_wgrad_bn {N}x{H}x{W}', _nb(x, dz, dz), 2 * dz.numel() * 27, 'stem_wgrad_reg_kernel'))
        return dw
    _run(lib, 'ledn_stem_conv_wgrad', x, _p(x), dtx, _p(dz), _p(dw), N, H, W, Cc, Ho, Wo, 32,
         _p(_f32(scale, Cc)), _p(_f32(shift, Cc)), _p(chan_map), _p(valid), float(pad_val),
         work=_TIMING is not None and (f'stem_conv_wgrad {N}x{H}x{W}', _nb(x, dz), 2 * dz.numel() * 27, 'stem_wgrad_reg_kernel'))
    return dw


def stem_weight_as_1x1(w):
    """[Cout][3][3][3] OIHW -> [Cout][32][1][1] matching im2col_stem's column order (differentiable)."""
    co, ci, kh, kw = w.shape
    w2 = w.permute(0, 2, 3, 1).reshape(co, kh * kw * ci)
    return torch.nn.functional.pad(w2, (0, 32 - kh * kw * ci)).reshape(co, 32, 1, 1).contiguous()


def mfma_weight_ok(w, groups=1):
    """shape gate of the MFMA conv path (csrc/conv_mfma.hip: conv_mfma_supported)."""
    co, cig, kh, kw = w.shape
    if groups != 1 and kh != 1:
        return False
    return (co % 16 == 0 or co <= 8) and (cig * groups) % 16 == 0 and kh == kw and kh in (1, 3)


def pack_conv_weights(w, mode=0, groups=1):
    """bf16 weight pack for the MFMA path: mode 0 forward [tap][co][ci], mode 1 dgrad;
    dense over the full input width (a grouped 1x1 gets zeros outside its group)."""
    lib = _lib.get_lib()
    Cout, cig, KH, KW = w.shape
    Cin = cig * groups
    _f32(w)
    out = torch.empty(Cout * Cin * KH * KW, dtype=torch.bfloat16, device=w.device)
    _check(lib, w, out)
    _run(lib, 'ledn_pack_conv_weights', w, _p(w), _p(out), Cout, Cin, KH, KW, mode, groups,
         work=_TIMING is not None and (f'packw {tuple(w.shape)} m{mode}', _nb(w, out), 0))
    return out


class WgradDefer:
    """Deferred reduction of the convolution weight gradients (ledn_conv2d_wgrad_partial / _finish_multi).  While
    `active`, conv2d_wgrad calls whose gradient goes into a persistent sink (`dw_out`: the Trainer's flat gradient
    buffer) only run the main MFMA kernel -- partial tiles into a buffer owned by that convolution -- and record a
    table entry; finish() sums all of them into their sinks with ONE launch (instead of one small launch per
    convolution on the critical stream: ~55 per step).  The buffers and the device table persist from step to step
    (the same convolutions arrive in the same order), so a captured hipGraph replays them unchanged."""
    active = False
    bufs = {}          # (dw.data_ptr(), floats) -> partial-tile buffer (persistent: nbx * pairs * KK * 1024 floats per convolution)
    pending = []       # [(key, WgradFinishEntry, buffer)] of the current step
    tables = {}        # keys -> (device table, entries, total chunks, reference tensor, pinned host staging buffer)

    @staticmethod
    def reset():
        WgradDefer.active, WgradDefer.bufs, WgradDefer.pending, WgradDefer.tables = False, {}, [], {}

    @staticmethod
    def finish():
        """sum every recorded convolution's partial tiles into its gradient sink: one launch.  A step may call this
        more than once with different sets (the overlapped gradient exchange flushes the non-stem convolutions first,
        the stem's follow at the end of the backward): every set has its own table, built once in the eager warm-up
        steps and kept -- with its host staging buffer -- for as long as the buffers live, so a captured graph never
        sees a host-to-device copy, nor a table freed under a recorded launch."""
        pend = WgradDefer.pending
        if not pend:
            return
        WgradDefer.pending = []
        lib = _lib.get_lib()
        keys = tuple(p[0] for p in pend)
        tab = WgradDefer.tables.get(keys)
        if tab is None:
            dev = pend[0][2].device
            if dev.type == 'cuda' and torch.cuda.is_current_stream_capturing():
                raise LednError('WgradDefer.finish: a new set of weight gradients arrived inside a graph capture '
                                '(run the same step eagerly first: Trainer.capture does)')
            arr = (_lib.WgradFinishEntry * len(pend))()
            chunk = 0
            for i, (_, e, _buf) in enumerate(pend):
                arr[i] = e
                arr[i].chunk0 = chunk
                chunk += e.pairs * e.KK * 16
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            if dev.type == 'cuda':
                host = host.pin_memory()
            tab = WgradDefer.tables[keys] = (host.to(dev), len(pend), chunk, pend[0][2], host)
        _run(lib, 'ledn_conv2d_wgrad_finish_multi', tab[3], tab[0].data_ptr(), tab[1], tab[2],
             work=_TIMING is not None and (f'wgrad_finish_multi x{tab[1]}', 0, 0, 'conv_wgrad_finish_multi_kernel'))


def conv2d_wgrad(x, dz, w_shape, *, stride=1, pad=0, dil=1, groups=1, xadd=None, in_scale=None,
                 in_shift=None, in_act=ACT_NONE, in_slope=None, bias=False, dw_out=None, db_out=None, _query=False):
    """Returns (dw [OIHW f32], db or None) of conv2d(pre(x), w).  dw_out / db_out: contiguous f32
    tensors the gradients are ACCUMULATED into (the trainer's flat gradient buffer) instead of
    fresh zeroed ones.  _query: no launch -> ledn_conv2d_wgrad_uses_mfma (0 VALU kernels, 1 conv_wgrad_mfma_kernel,
    2 conv3x3_wgrad_narrow_kernel, 3 conv1x1_wgrad_reg_kernel)."""
    lib = _lib.get_lib()
    N, H, W, Cin = x.shape
    cof, cigf, KH, KW = w_shape
    _, Ho, Wo, Cout = dz.shape
    if Cout != cof or Cin != cigf * groups:
        raise LednError('conv2d_wgrad: channel mismatch')
    dw = dw_out if dw_out is not None else torch.zeros(w_shape, dtype=torch.float32, device=x.device)
    db = None
    if bias:
        db = db_out if db_out is not None else zeros_f32((Cout,), x.device)
    if tuple(dw.shape) != tuple(w_shape) or dw.dtype != torch.float32 or (db is not None and db.numel() != Cout):
        raise LednError('conv2d_wgrad: gradient buffer shape/dtype mismatch')
    _check(lib, x, dz, xadd, in_scale, in_shift, in_slope, dw, db)
    if in_act == ACT_PRELU and in_slope is None:
        raise LednError('conv2d_wgrad: in_act PRELU needs in_slope')
    d = _lib.WgradDesc()
    d.x, d.xadd, d.dz, d.dw, d.db = _p(x), _p(xadd), _p(dz), _p(dw), _p(db)
    d.in_scale, d.in_shift, d.in_slope = _p(_f32(in_scale, Cin)), _p(_f32(in_shift, Cin)), _p(_f32(in_slope, Cin))
    d.ws_co, d.ws_ci, d.ws_tap = cigf * KH * KW, KH * KW, 1
    d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = N, H, W, Cin, Ho, Wo, Cout
    d.KH, d.KW, d.stride, d.pad, d.dil, d.groups = KH, KW, stride, pad, dil, groups
    d.in_act, d.dtype_x, d.dtype_dz = in_act, _dt(x), _dt(dz)
    flops = 2 * N * Ho * Wo * Cout * (Cin // groups) * KH * KW
    if _query:
        return int(lib.cdll.ledn_conv2d_wgrad_uses_mfma(d))
    if WgradDefer.active and dw_out is not None:
        nfl = int(lib.cdll.ledn_conv2d_wgrad_partial_floats(d))
        if nfl > 0:
            key = (dw.data_ptr(), nfl)
            if any(p[0] == key for p in WgradDefer.pending):
                # the same weight applied twice before its gradients were summed (a shared module): one partial buffer
                # and one table entry per weight -- the second use takes the immediate path, which accumulates
                nfl = 0
        if nfl > 0:
            buf = WgradDefer.bufs.get(key)
            if buf is None:
                buf = WgradDefer.bufs[key] = torch.empty(nfl, dtype=torch.float32, device=x.device)
            e = _lib.WgradFinishEntry()
            _run(lib, 'ledn_conv2d_wgrad_partial', x, d, buf.data_ptr(), nfl, C.byref(e),
                 work=_TIMING is not None and (f'wgrad{KH}x{KW} {Cin}->{Cout} g{groups} s{stride} {N}x{H}x{W}',
                                               _nb(x, xadd, dz, dw), flops,
                                               'conv1x1_wgrad_reg_kernel' if lib.cdll.ledn_conv2d_wgrad_uses_mfma(d) == 3 else 'conv_wgrad_mfma_kernel'))
            WgradDefer.pending.append((key, e, buf))
            return dw, db
    _run(lib, 'ledn_conv2d_wgrad', x, d,
         work=_TIMING is not None and (f'wgrad{KH}x{KW} {Cin}->{Cout} g{groups} s{stride} {N}x{H}x{W}', _nb(x, xadd, dz, dw), flops,
                                       ('conv_wgrad_direct', 'conv_wgrad_mfma_kernel', 'conv3x3_wgrad_narrow_kernel', 'conv1x1_wgrad_reg_kernel', 'conv_wgrad_f32_mfma_kernel')[lib.cdll.ledn_conv2d_wgrad_uses_mfma(d)]))
    return dw, db


def dwconv2d(x, w_khwc, *, stride=1, pad=-1, dil=(1, 1, 1, 1), group_size=None, out_scale=None,
             out_shift=None, act=ACT_NONE, slope=None, stats=None, ext1=False, out_dtype=None, defer_stats=False):
    """Depthwise conv.  x: [N,H,W,C]; w_khwc: [KH,KW,C] f32."""
    lib = _lib.get_lib()
    N, H, W, Cc = x.shape
    KH, KW, Cw = w_khwc.shape
    if Cw != Cc:
        raise LednError('dwconv2d: channel mismatch')
    group_size = group_size or Cc
    Hx, Wx = H + int(ext1), W + int(ext1)
    d0 = dil[0]
    ph = pad if pad >= 0 else d0 * (KH - 1) // 2
    pw = pad if pad >= 0 else d0 * (KW - 1) // 2
    Ho, Wo = conv_out_size(Hx, KH, stride, ph, d0), conv_out_size(Wx, KW, stride, pw, d0)
    y = torch.empty((N, Ho, Wo, Cc), dtype=out_dtype or x.dtype, device=x.device)
    _check(lib, x, w_khwc, y, out_scale, out_shift, slope)
    d = _lib.DwDesc()
    d.x, d.w, d.y = _p(x), _p(_f32(w_khwc)), _p(y)
    d.out_scale, d.out_shift, d.slope = _p(_f32(out_scale, Cc)), _p(_f32(out_shift, Cc)), _p(_f32(slope, Cc))
    if stats is not None:
        _check(lib, stats[0], stats[1])
        d.stat_sum, d.stat_sqsum = _p(_f32(stats[0], Cc)), _p(_f32(stats[1], Cc))
    d.N, d.H, d.W, d.C, d.Ho, d.Wo = N, H, W, Cc, Ho, Wo
    d.KH, d.KW, d.stride, d.pad = KH, KW, stride, pad
    for i in range(4):
        d.dil[i] = dil[i] if i < len(dil) else dil[-1]
    d.group_size, d.act_out, d.ext1 = group_size, act, int(ext1)
    d.dtype_x, d.dtype_y = _dt(x), _dt(y)
    _run_stats(lib, 'ledn_dwconv2d', x, stats, defer_stats, d, work=_TIMING is not None and (f'dw{KH}x{KW} C{Cc} s{stride} {N}x{H}x{W}', _nb(x, y, w_khwc), 2 * y.numel() * KH * KW))
    return y


def sesp_pyramid(x, w_b33n, dil, stride):
    """x: [N,H,W,n]; w: [4,3,3,n] f32 -> [N,Ho,Wo,4n] (cat layout, HFF sums applied)."""
    lib = _lib.get_lib()
    N, H, W, n = x.shape
    if tuple(w_b33n.shape) != (4, 3, 3, n):
        raise LednError('sesp_pyramid: weight shape')
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    y = torch.empty((N, Ho, Wo, 4 * n), dtype=x.dtype, device=x.device)
    _check(lib, x, w_b33n, y)
    d = _lib.PyrDesc()
    d.x, d.w, d.y = _p(x), _p(_f32(w_b33n)), _p(y)
    d.N, d.H, d.W, d.n, d.Ho, d.Wo, d.stride = N, H, W, n, Ho, Wo, stride
    for i in range(4):
        d.dil[i] = dil[i]
    d.dtype_x = d.dtype_y = _dt(x)
    _run(lib, 'ledn_sesp_pyramid', x, d, work=_TIMING is not None and (f'pyr n{n} s{stride} {N}x{H}x{W}', _nb(x, y, w_b33n), 2 * y.numel() * 9))
    return y


def channel_stats(x, xadd=None, stats=None, defer_stats=False):
    """x: [..., C] -> (sum[C], sqsum[C]) f32 (accumulated into `stats` when given)."""
    lib = _lib.get_lib()
    Cc = x.shape[-1]
    P = x.numel() // Cc
    if stats is None:
        stats = (zeros_f32(Cc, x.device), zeros_f32(Cc, x.device))
    _check(lib, x, xadd, stats[0], stats[1])
    _run_stats(lib, 'ledn_channel_stats', x, stats, defer_stats, _p(x), _p(xadd), P, Cc, _dt(x), _p(_f32(stats[0], Cc)),
               _p(_f32(stats[1], Cc)), work=_TIMING is not None and (f'stats C{Cc} P{P}', _nb(x, xadd), 3 * x.numel()))
    return stats


class PendingRows:
    """Hand-off of deferred convolution statistics (conv2d(defer_stats=True) -> the bn_finalize that
    follows it): keyed by the statistics buffer; at most one entry is ever pending."""
    entry = None

    @staticmethod
    def put(sum_tensor, part_ptr, rows, lib):
        if PendingRows.entry is not None:
            raise LednError('deferred statistics of the previous convolution were never finalized')
        PendingRows.entry = (sum_tensor.data_ptr(), part_ptr, rows, lib)

    @staticmethod
    def take(sum_tensor, lib):
        e = PendingRows.entry
        if e is None:
            return None
        if e[0] != sum_tensor.data_ptr() or e[3] is not lib:
            raise LednError('deferred statistics pending for another buffer: a ledn call came between the '
                            'convolution and its bn_finalize')
        PendingRows.entry = None
        return e[1], e[2]


def bn_finalize(stats, count, gamma, beta, running_mean=None, running_var=None, momentum=0.1, eps=1e-5):
    """-> (scale, shift, mean, invstd); updates the running statistics in place."""
    lib = _lib.get_lib()
    Cc = stats[0].numel()
    dev = stats[0].device
    scale, shift, mean, invstd = (torch.empty(Cc, dtype=torch.float32, device=dev) for _ in range(4))
    _check(lib, stats[0], stats[1], gamma, beta, running_mean, running_var)
    pend = PendingRows.take(stats[0], lib)
    if pend is not None:
        _run(lib, 'ledn_bn_finalize_rows', stats[0], pend[0], pend[1], float(count), _p(_f32(gamma, Cc)),
             _p(_f32(beta, Cc)), _p(_f32(running_mean, Cc)), _p(_f32(running_var, Cc)), momentum, eps,
             _p(scale), _p(shift), _p(mean), _p(invstd), _p(stats[0]), _p(stats[1]), Cc,
             work=_TIMING is not None and (f'bnfin_rows C{Cc}', 0, 0, 'bn_finalize_rows_kernel'))
        return scale, shift, mean, invstd
    _run(lib, 'ledn_bn_finalize', stats[0], _p(stats[0]), _p(stats[1]), float(count), _p(_f32(gamma, Cc)),
         _p(_f32(beta, Cc)), _p(_f32(running_mean, Cc)), _p(_f32(running_var, Cc)), momentum, eps,
         _p(scale), _p(shift), _p(mean), _p(invstd), Cc, work=_TIMING is not None and (f'bnfin C{Cc}', 0, 0))
    return scale, shift, mean, invstd


def affine_stats_ok(x, act, out_dtype=None):
    """ledn_affine_act can also deliver the per-channel sums of its output (the statistics of a BatchNorm that follows)"""
    Cc = x.shape[-1]
    return (x.dtype == torch.bfloat16 and out_dtype in (None, torch.bfloat16) and 8 <= Cc <= 512 and Cc & (Cc - 1) == 0
            and x.numel() // 8 >= 4096 and act in (ACT_NONE, ACT_RELU, ACT_PRELU))


def affine_act(x, scale=None, shift=None, *, act=ACT_NONE, slope=None, res=None, res_mode=RES_NONE,
               xadd=None, out_dtype=None, stats=None):
    """stats = (sum, sqsum) f32 [C]: ACCUMULATE the per-channel sums of the output into them (affine_stats_ok decides)"""
    lib = _lib.get_lib()
    Cc = x.shape[-1]
    y = torch.empty(x.shape, dtype=out_dtype or x.dtype, device=x.device)
    if res is not None and (res.shape != y.shape or res.dtype != y.dtype):
        raise LednError('affine_act: residual shape/dtype mismatch')
    _check(lib, x, y, xadd, res, scale, shift, slope)
    d = _lib.AffineDesc()
    d.x, d.xadd, d.y, d.res = _p(x), _p(xadd), _p(y), _p(res)
    d.scale, d.shift, d.slope = _p(_f32(scale, Cc)), _p(_f32(shift, Cc)), _p(_f32(slope, Cc))
    d.P, d.C, d.act = x.numel() // Cc, Cc, act
    d.res_mode = res_mode if res is not None else RES_NONE
    d.dtype_x, d.dtype_y = _dt(x), _dt(y)
    if stats is not None:
        _check(lib, stats[0], stats[1])
        d.stat_sum, d.stat_sqsum = _p(_f32(stats[0], Cc)), _p(_f32(stats[1], Cc))
        try:
            _run(lib, 'ledn_affine_act', x, d, work=_TIMING is not None and (f'affine+stats C{Cc} P{x.numel() // Cc}', _nb(x, y), 4 * x.numel(), 'affine_stats_fast_kernel'))
            return y
        except LednError:       # outside the streaming kernel's gate (include/ledn.h): the two-launch form
            d.stat_sum = d.stat_sqsum = None
    _run(lib, 'ledn_affine_act', x, d, work=_TIMING is not None and (f'affine C{Cc} P{x.numel() // Cc}', _nb(x, xadd, y, res), 2 * x.numel()))
    if stats is not None:
        channel_stats(y, stats=stats)
    return y


def nchw_to_nhwc(x, out_dtype, scale=None, shift=None, chan_map=None, valid=None, pad_val=0.0):
    """x: [N,C,H,W] uint8/float32/bfloat16 -> [N,H,W,C] out_dtype, y = x[map[c]]*scale[c]+shift[c];
    valid / pad_val: batch padding as in im2col_stem_planar."""
    lib = _lib.get_lib()
    N, Cc, H, W = x.shape
    dtx = {torch.float32: F32, torch.bfloat16: BF16, torch.uint8: _lib.U8}.get(x.dtype)
    if dtx is None:
        raise LednError(f'nchw_to_nhwc: unsupported input dtype {x.dtype}')
    y = torch.empty((N, H, W, Cc), dtype=out_dtype, device=x.device)
    if chan_map is not None and (chan_map.dtype != torch.int32 or chan_map.numel() != Cc):
        raise LednError('nchw_to_nhwc: chan_map must be int32[C]')
    _check(lib, x, y, scale, shift, chan_map, _valid_hw(valid, N, x))
    _run(lib, 'ledn_nchw_to_nhwc', x, _p(x), dtx, _p(y), _DT[out_dtype], N, Cc, H, W, _p(_f32(scale, Cc)),
         _p(_f32(shift, Cc)), _p(chan_map), _p(valid), float(pad_val), work=_TIMING is not None and (f'nchw2nhwc {N}x{Cc}x{H}x{W}', _nb(x, y), 2 * x.numel()))
    return y


def bilinear(x, size, *, add=None, out_dtype=None, nchw=False, argmax=False):
    """F.interpolate(bilinear, align_corners=False) of NHWC x to `size` (+ add)."""
    lib = _lib.get_lib()
    N, H, W, Cc = x.shape
    Ho, Wo = int(size[0]), int(size[1])
    odt = torch.float32 if nchw else (out_dtype or x.dtype)
    y = torch.empty((N, Cc, Ho, Wo) if nchw else (N, Ho, Wo, Cc), dtype=odt, device=x.device)
    am = torch.empty((N, Ho, Wo), dtype=torch.uint8, device=x.device) if argmax else None
    if add is not None and (tuple(add.shape) != (N, Ho, Wo, Cc) or add.dtype != odt):
        raise LednError('bilinear: add shape/dtype mismatch')
    _check(lib, x, y, add, am)
    d = _lib.ResizeDesc()
    d.x, d.add, d.y, d.argmax = _p(x), _p(add), _p(y), _p(am)
    d.N, d.H, d.W, d.C, d.Ho, d.Wo = N, H, W, Cc, Ho, Wo
    d.out_nchw, d.dtype_x, d.dtype_y = int(nchw), _dt(x), _DT[odt]
    _run(lib, 'ledn_bilinear', x, d, work=_TIMING is not None and (f'bilinear C{Cc} {H}x{W}->{Ho}x{Wo} N{N}', _nb(x, y, add, am), 8 * y.numel()))
    return (y, am) if argmax else y


def adaptive_avgpool(x, S, xadd=None):
    lib = _lib.get_lib()
    N, H, W, Cc = x.shape
    y = torch.empty((N, S, S, Cc), dtype=torch.float32, device=x.device)
    _check(lib, x, xadd, y)
    _run(lib, 'ledn_adaptive_avgpool', x, _p(x), _p(xadd), _p(y), N, H, W, Cc, S, _dt(x),
         work=_TIMING is not None and (f'apool S{S} C{Cc} {N}x{H}x{W}', _nb(x, xadd, y), x.numel()))
    return y


POOL_PYRAMID = _knob_int('LEDN_POOL_PYRAMID', 1)    # the coarse context pools from the finest one (one pass over the map instead of four)


def multi_pool(x, sizes, xadd=None):
    """adaptive average pools of x (+ xadd) [N,H,W,C] to S x S for S in sizes (f32).  When every window grid is a
    refinement of the finest one (H, W multiples of the largest S, which the others divide) the coarser pools are averages of
    equal numbers of the finest pool's cells: the map is read ONCE (the 16x16 pool) and the others come from that small map
    -- Muti_AFF's four pools (classification/model_utils.py:402-423) cost four passes over the map otherwise."""
    big = max(sizes)
    H, W = x.shape[1], x.shape[2]
    if not (POOL_PYRAMID and H % big == 0 and W % big == 0 and all(big % S == 0 for S in sizes)):
        return tuple(adaptive_avgpool(x, S, xadd=xadd) for S in sizes)
    fine = adaptive_avgpool(x, big, xadd=xadd)
    return tuple(fine if S == big else adaptive_avgpool(fine, S) for S in sizes)


def avgpool2d(x, k, stride, pad):
    """nn.AvgPool2d(k, stride, pad) (zero padding counted in the divisor) on NHWC x."""
    lib = _lib.get_lib()
    N, H, W, Cc = x.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    if Ho <= 0 or Wo <= 0:
        raise LednError(f'avgpool2d: a {k}x{k} window (pad {pad}) does not fit a {H}x{W} map')
    y = torch.empty((N, Ho, Wo, Cc), dtype=x.dtype, device=x.device)
    _check(lib, x, y)
    _run(lib, 'ledn_avgpool2d', x, _p(x), _p(y), N, H, W, Cc, Ho, Wo, k, stride, pad, _dt(x),
         work=_TIMING is not None and (f'avgpool{k}s{stride} C{Cc} {N}x{H}x{W}', _nb(x, y), k * k * y.numel()))
    return y


def avgpool2d_bwd(dy, in_hw, k, stride, pad):
    lib = _lib.get_lib()
    N, Ho, Wo, Cc = dy.shape
    H, W = in_hw
    dx = torch.empty((N, H, W, Cc), dtype=dy.dtype, device=dy.device)
    _check(lib, dy, dx)
    _run(lib, 'ledn_avgpool2d_bwd', dy, _p(dy), _p(dx), N, H, W, Cc, Ho, Wo, k, stride, pad, _dt(dy),
         work=_TIMING is not None and (f'avgpool{k}s{stride}_bwd C{Cc} {N}x{H}x{W}', _nb(dy, dx), k * k * dx.numel()))
    return dx


def avgpool3x3s2(x):
    lib = _lib.get_lib()
    N, H, W, Cc = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty((N, Ho, Wo, Cc), dtype=x.dtype, device=x.device)
    _check(lib, x, y)
    _run(lib, 'ledn_avgpool3x3s2', x, _p(x), _p(y), N, H, W, Cc, Ho, Wo, _dt(x),
         work=_TIMING is not None and (f'avgpool3s2 C{Cc} {N}x{H}x{W}', _nb(x, y), 9 * y.numel()))
    return y


def window_attn(qkv, biasT, heads, ws=8):
    lib = _lib.get_lib()
    N, H, W, C3 = qkv.shape
    Cc = C3 // 3
    if tuple(biasT.shape) != (heads, ws * ws, ws * ws):
        raise LednError('window_attn: biasT shape')
    out = torch.empty((N, H, W, Cc), dtype=qkv.dtype, device=qkv.device)
    _check(lib, qkv, biasT, out)
    nwin = N * ((H + ws - 1) // ws) * ((W + ws - 1) // ws)
    _run(lib, 'ledn_window_attn', qkv, _p(qkv), _p(_f32(biasT)), _p(out), N, H, W, Cc, heads, ws, _dt(qkv),
         work=_TIMING is not None and (f'wattn C{Cc} h{heads} {N}x{H}x{W}', _nb(qkv, out, biasT), 4 * nwin * (ws * ws) ** 2 * Cc))
    return out


def relpos_bias(table, index):
    """[heads][key j][query i] relative-position bias from the (2ws-1)^2 x heads table and the module's index buffer."""
    lib = _lib.get_lib()
    R, heads = table.shape
    T = index.shape[0]
    if index.dtype != torch.int64 or tuple(index.shape) != (T, T):
        raise LednError('relpos_bias: index must be the int64 [T,T] relative_position_index buffer')
    out = torch.empty((heads, T, T), dtype=torch.float32, device=table.device)
    _check(lib, table, index, out)
    _run(lib, 'ledn_relpos_bias', table, _p(_f32(table)), _p(index), _p(out), R, heads, T,
         work=_TIMING is not None and (f'relpos_bias h{heads} T{T}', _nb(out), 0))
    return out


def relpos_bias_bwd(dbias, index, dtable):
    """dtable [R,heads] (f32, accumulated into) += adjoint of relpos_bias"""
    lib = _lib.get_lib()
    R, heads = dtable.shape
    T = index.shape[0]
    _check(lib, dbias, index, dtable)
    _run(lib, 'ledn_relpos_bias_bwd', dbias, _p(_f32(dbias, heads * T * T)), _p(index), _p(_f32(dtable)), R, heads, T,
         work=_TIMING is not None and (f'relpos_bias_bwd h{heads} T{T}', _nb(dbias), 0))
    return dtable


def getb_pool(a, local, ws=8):
    lib = _lib.get_lib()
    N, H, W, Cc = a.shape
    if local.shape != a.shape or local.dtype != a.dtype:
        raise LednError('getb_pool: shape/dtype mismatch')
    out = torch.empty_like(a)
    _check(lib, a, local, out)
    _run(lib, 'ledn_getb_pool', a, _p(a), _p(local), _p(out), N, H, W, Cc, ws, _dt(a),
         work=_TIMING is not None and (f'getbpool C{Cc} {N}x{H}x{W}', _nb(a, local, out), 17 * a.numel()))
    return out


def mfaf_gate(x, r, xl, ctx, affines, act=ACT_NONE):
    """ctx: [c1 (N,4,4,C), c2 (N,8,8,C), c3 (N,16,16,C), xg (N,1,1,C)] f32;
    affines: 5 (scale, shift) pairs for xl, c1, c2, c3, xg."""
    lib = _lib.get_lib()
    N, H, W, Cc = x.shape
    if r.shape != x.shape or xl.shape != x.shape or r.dtype != x.dtype or xl.dtype != x.dtype:
        raise LednError('mfaf_gate: shape/dtype mismatch')
    out = torch.empty_like(x)
    d = _lib.MfafDesc()
    d.x, d.r, d.xl, d.out = _p(x), _p(r), _p(xl), _p(out)
    keep = [x, r, xl, out]
    for k, c in enumerate(ctx):
        if c.shape[0] != N or c.shape[1] != c.shape[2] or c.shape[3] != Cc:
            raise LednError('mfaf_gate: context map shape')
        d.ctx[k] = _p(_f32(c))
        d.ctx_size[k] = c.shape[1]
        keep.append(c)
    for k, (s, b) in enumerate(affines):
        d.scale[k], d.shift[k] = _p(_f32(s, Cc)), _p(_f32(b, Cc))
        keep += [s, b]
    _check(lib, *keep)
    d.N, d.H, d.W, d.C, d.dtype, d.act = N, H, W, Cc, _dt(x), act
    _run(lib, 'ledn_mfaf_gate', x, d, work=_TIMING is not None and (f'mfafgate C{Cc} {N}x{H}x{W}', _nb(x, r, xl, out), 20 * x.numel()))
    return out


def seam_edge(seg, percentile=0.8, thr=0.1, final_thr=0.1):
    """seg: [N,h,w,1] f32 -> edge [N,h,w,1] f32 in {0,1}.  percentile=None: fixed `thr`."""
    lib = _lib.get_lib()
    N, h, w, one = seg.shape
    if one != 1 or seg.dtype != torch.float32:
        raise LednError('seam_edge: seg must be [N,h,w,1] float32')
    edge = torch.empty_like(seg)
    scratch = torch.empty((N, 3, h, w), dtype=torch.float32, device=seg.device)
    kth = 0 if percentile is None else max(1, math.ceil(percentile * h * w))
    _check(lib, seg, edge, scratch)
    _run(lib, 'ledn_seam_edge', seg, _p(seg), _p(edge), _p(scratch), N, h, w, kth, thr, final_thr,
         work=_TIMING is not None and (f'seam {N}x{h}x{w}', _nb(seg, edge), 60 * seg.numel()))
    return edge
