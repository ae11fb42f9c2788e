"""Tensor-level wrappers for the backward / loss / optimizer entry points of the
C ABI (second half of include/ledn.h).  Same conventions as ops.py."""
import ctypes as C

import torch

from . import _lib
from . import ops as _ops
from .ops import (ACT_NONE, RES_NONE, _DT, LednError, _check, _dt, _f32, _nb, _p, _run,  # noqa: F401
                  conv_out_size)


class _BnBwd:
    """State of one BatchNorm(+activation, +residual) backward between its two kernels."""
    __slots__ = ('lib', 'd', 'z', 'bn', 'Cc', 'P', 'local', 'sunk', 'dz', 'dres', 'dslope', 'slope_sunk', 'keep', 'reduce_pending', 'hd')


from ._env import knob_int as _knob_int  # noqa: E402
BN_ROWS = _knob_int('LEDN_BN_ROWS', 0)   # measured r3k: 13.85 vs 13.85 ms -- the 69 summing launches it removes were not on the critical path; off: keeps the reduction order fixed


def bn_act_bwd_reduce(z, dy, *, scale=None, shift=None, mean=None, invstd=None, act=ACT_NONE, slope=None,
                      res=None, res_mode=RES_NONE, count=None, want_dres=False, sinks=None, sync=False,
                      dz_add=None, dres_add=None, launch=True, head=None, no_dz=False):
    """First half of bn_act_bwd: the per-channel sums (sum g*xhat, sum g) of THIS rank's shard, reduced
    into the parameter-gradient sinks when given (they ARE d_gamma, d_beta of the local shard -- under
    SyncBN too: torch.nn.SyncBatchNorm keeps grad_weight / grad_bias local, DDP averages them later),
    else into zeroed scratch.  -> state for bn_act_bwd_sync / bn_act_bwd_apply."""
    lib = _lib.get_lib()
    st = _BnBwd()
    Cc = z.shape[-1]
    P = z.numel() // Cc
    bn = mean is not None
    d = _lib.BnBwdDesc()
    sk_g, sk_b, sk_s = sinks if sinks is not None else (None, None, None)
    sunk = bool(bn and sk_g is not None and sk_b is not None)
    if sunk and sync and sk_b.data_ptr() != sk_g.data_ptr() + 4 * Cc:
        sunk = False                       # (never the case for an nn.BatchNorm: weight, bias are adjacent)
    if sunk:
        _f32(sk_g, Cc), _f32(sk_b, Cc)
        sum_gx, sum_g = sk_g, sk_b
        local = torch.as_strided(sk_g, (2, Cc), (Cc, 1)) if sync else None
    else:
        local = _ops.zeros_f32((2, Cc), z.device)           # [sum g*xhat (d_gamma), sum g (d_beta)]
        sum_gx, sum_g = local[0], local[1]
    dslope, slope_sunk = None, False
    if slope is not None:
        if sk_s is not None:
            dslope, slope_sunk = _f32(sk_s, Cc), True
        else:
            dslope = _ops.zeros_f32(Cc, z.device)
    dz = torch.empty_like(z) if not no_dz else None      # (no_dz: the apply half belongs to the consumer -- ledn_stem_conv_wgrad_bn)
    dres = torch.empty_like(dy) if (want_dres and res_mode != RES_NONE) else None
    if head is not None and (dy is not None or res is not None or want_dres):
        raise LednError('bn_act_bwd: the two-class head form takes no dy / residual')
    if dz_add is not None and (dz_add.shape != z.shape or dz_add.dtype != z.dtype):
        raise LednError('bn_act_bwd: dz_add must match z')
    if dres_add is not None and (dres is None or dres_add.shape != dres.shape or dres_add.dtype != dres.dtype):
        raise LednError('bn_act_bwd: dres_add needs want_dres and must match the residual gradient')
    _check(lib, z, dy, res, scale, shift, slope, mean, invstd, sum_g, sum_gx, dslope, dz_add, dres_add)
    d.z, d.res, d.dy = _p(z), _p(res), _p(dy)
    d.dz_add, d.dres_add = _p(dz_add), _p(dres_add)
    d.scale, d.shift, d.slope = _p(_f32(scale, Cc)), _p(_f32(shift, Cc)), _p(_f32(slope, Cc))
    d.mean, d.invstd = _p(_f32(mean, Cc)), _p(_f32(invstd, Cc))
    d.sum_g, d.sum_gx = sum_g.data_ptr(), sum_gx.data_ptr()
    d.dslope, d.dz, d.dres = _p(dslope), _p(dz), _p(dres)
    d.count = float(count if count is not None else P)
    d.P, d.C, d.act = P, Cc, act
    d.res_mode = res_mode if res is not None else RES_NONE
    d.bn_mode, d.dtype_z, d.dtype_y = int(bn), _dt(z), _dt(dy if dy is not None else z)
    rows = None
    st.hd = None
    if head is not None:
        # LEDHead's two-class heads: dy = conv_transpose3x3(head_dz, w) is recomputed inside both passes
        # (ledn_head_bwd_reduce / _apply)
        hdz, hw, hdw, hdb = (tuple(head) + (None, None))[:4]
        _check(lib, hdz, hw, hdw, hdb)
        hd = st.hd = _lib.HeadBwdDesc()
        hd.bn = d
        hd.head_dz, hd.w, hd.dw, hd.db = _p(hdz), _p(_f32(hw)), _p(_f32(hdw)), _p(_f32(hdb))
        hd.N, hd.H, hd.W, hd.Co, hd.dtype_dz = z.shape[0], z.shape[1], z.shape[2], hw.shape[0], _dt(hdz)
        if hdw is not None and (tuple(hdw.shape) != tuple(hw.shape) or (hdb is not None and hdb.numel() != hw.shape[0])):
            raise LednError('bn_act_bwd: the head weight / bias gradient sinks do not match the filter')
        if tuple(hdz.shape) != (z.shape[0], z.shape[1], z.shape[2], hw.shape[0]) or tuple(hw.shape) != (hw.shape[0], Cc, 3, 3) \
                or not lib.cdll.ledn_head_bwd_supported(hd):
            raise LednError('bn_act_bwd: shape outside the two-class head kernels (head_bwd_ok decides)')
        st.reduce_pending = False
        _run(lib, 'ledn_head_bwd_reduce', z, hd, work=_ops._TIMING is not None and (
            f'headbwd_reduce C{Cc} P{P}', _nb(z, hdz), 10 * z.numel(), 'head_bwd_reduce_kernel'))
        st.lib, st.d, st.z, st.bn, st.Cc, st.P, st.local, st.sunk = lib, d, z, bn, Cc, P, local, sunk
        st.dz, st.dres, st.dslope, st.slope_sunk = dz, None, dslope, slope_sunk
        st.keep = (hdz, None, scale, shift, slope, mean, invstd, sum_g, sum_gx, sk_g, sk_b, dz_add, None, None, hw, hdw, hdb)
        return st
    if BN_ROWS and not sync and not no_dz and (bn or slope is not None) and z.dtype == torch.bfloat16 and dy.dtype == torch.bfloat16:
        # the reduce pass leaves per-row sums (float atomics into 32 zeroed rows), the apply pass adds them up: no
        # summing launch between the two (ledn.h: ledn_bnbwd_desc.rows).  Not under SyncBN: the all-reduce needs totals.
        rows = _ops.zeros_f32((_lib.BNBWD_ROWS, 3, Cc), z.device)
        d.rows = _p(rows)
    st.reduce_pending = bool(bn or slope is not None)
    if st.reduce_pending and launch:
        st.reduce_pending = False
        _run(lib, 'ledn_bn_act_bwd_reduce', z, d, work=_ops._TIMING is not None and (f'bnbwd_reduce C{Cc} P{P}', _nb(z, dy, res), 6 * z.numel()))
    st.lib, st.d, st.z, st.bn, st.Cc, st.P, st.local, st.sunk = lib, d, z, bn, Cc, P, local, sunk
    st.dz, st.dres, st.dslope, st.slope_sunk = dz, dres, dslope, slope_sunk
    st.keep = (dy, res, scale, shift, slope, mean, invstd, sum_g, sum_gx, sk_g, sk_b, dz_add, dres_add, rows)
    return st


def bn_act_bwd_sync(st, sync):
    """SyncBN: the apply kernel needs the GLOBAL sums; all-reduce the local ones OUT OF PLACE into scratch
    (the local sums stay what they are: this rank's parameter gradient)."""
    if st.bn and sync is not None:
        glob = _ops.zeros_f32((2, st.Cc), st.z.device)
        sync.all_reduce(st.local, glob)
        st.d.sum_gx, st.d.sum_g = glob[0].data_ptr(), glob[1].data_ptr()
        st.keep += (glob,)


BN_FUSED = _knob_int('LEDN_BN_FUSED', 0)      # the persistent one-pass BatchNorm backward (csrc/stream_fast.hip bn_bwd_fused_kernel)


def set_bn_fused(flag):
    """switch the one-pass BatchNorm backward on / off (library option + this module's dispatch)"""
    global BN_FUSED
    BN_FUSED = int(bool(flag))
    _lib.get_lib().set_option(_lib.OPT_BN_FUSED, BN_FUSED)


def _bn_bwd_fused(st):
    """try ledn_bn_act_bwd_fused on the prepared descriptor: True = launched (both halves done), False = not applicable"""
    lib, z = st.lib, st.z
    if not lib.is_hip:
        return False
    if not getattr(lib, '_bn_fused_on', False):
        lib.set_option(_lib.OPT_BN_FUSED, 1)
        lib._bn_fused_on = True
    stream = _ops._stream(lib, z)
    lib.ensure_workspace(z.device, 0, stream=stream)
    timing = _ops._TIMING is not None
    if timing:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = lib.cdll.ledn_bn_act_bwd_fused(st.d, stream)
    if rc == _lib.ESKIP:
        return False
    if rc != _lib.OK:
        raise LednError(f'ledn_bn_act_bwd_fused failed: {rc}')
    if timing:
        e1.record()
        _ops._TIMING.append(('ledn_bn_act_bwd_fused', (f'bnbwd_fused C{st.Cc} P{st.P}', _nb(z, st.keep[0], st.keep[1], st.dz, st.dres),
                                                       14 * z.numel(), 'bn_bwd_fused_kernel'), e0, e1))
    return True


def _bn_bwd_result(st):
    give = st.bn and not st.sunk
    if give and st.keep[9] is not None and st.keep[10] is not None:
        st.keep[9].add_(st.local[0])
        st.keep[10].add_(st.local[1])
        give = False
    return (st.dz, st.dres, (st.local[0] if give else None), (st.local[1] if give else None),
            None if st.slope_sunk else st.dslope)


def bn_act_bwd_apply(st):
    """-> (dz, dres, dgamma, dbeta, dslope); a gradient that went into its sink is returned as None."""
    z, d = st.z, st.d
    if getattr(st, 'hd', None) is not None:
        st.hd.bn = d                       # (the SyncBN exchange re-pointed the sums)
        _run(st.lib, 'ledn_head_bwd_apply', z, st.hd, work=_ops._TIMING is not None and (
            f'headbwd_apply C{st.Cc} P{st.P}', _nb(z, st.keep[0], st.keep[11], st.dz), 8 * z.numel(), 'head_bwd_apply_kernel'))
    else:
        _run(st.lib, 'ledn_bn_act_bwd_apply', z, d, work=_ops._TIMING is not None and (
            f'bnbwd_apply C{st.Cc} P{st.P}', _nb(z, st.keep[0], st.keep[1], st.dz, st.dres), 8 * z.numel()))
    give = st.bn and not st.sunk
    if give and st.keep[9] is not None and st.keep[10] is not None:
        st.keep[9].add_(st.local[0])       # sinks exist but were not adjacent (see reduce): add the local sums
        st.keep[10].add_(st.local[1])
        give = False
    return (st.dz, st.dres, (st.local[0] if give else None), (st.local[1] if give else None),
            None if st.slope_sunk else st.dslope)


def bn_act_bwd(z, dy, *, scale=None, shift=None, mean=None, invstd=None, act=ACT_NONE, slope=None,
               res=None, res_mode=RES_NONE, count=None, want_dres=False, sync=None, sinks=None, dz_add=None,
               dres_add=None, head=None):
    """Backward of y = act(res_mode(z*scale+shift, res)).
    head = (head_dz [N,H,W,2] bf16, w [2,32,3,3][, dw sink, db sink or None]): dy is NOT given but recomputed from the logits'
    gradient of a two-class 3x3 head on y (LEDHead's norm -> act -> conv; see head_bwd_ok); with a dw sink the reduce pass also ACCUMULATES the head's weight / bias gradient into the sinks.
    BN mode (mean/invstd given): returns (dz, dres, dgamma, dbeta, dslope).
    Plain mode: returns (dz, dres, None, None, dslope).
    sync: optional collective object (train._Collective) all-reducing the [2,C] (sum_gx, sum_g) sums (SyncBN);
    the returned / sunk dgamma, dbeta are always the LOCAL sums (DDP semantics).
    dz_add / dres_add: partial gradients of z / res from another consumer of the same tensor, added in the apply pass.
    sinks: optional (dgamma, dbeta, dslope) ZEROED f32 [C] buffers (any may be None) the kernels
    reduce straight into (the trainer's gradient views); the matching return value is then None."""
    fused = BN_FUSED and sync is None and z.is_cuda and z.dtype == torch.bfloat16 and _ops._slot(z) == 0 and head is None
    st = bn_act_bwd_reduce(z, dy, scale=scale, shift=shift, mean=mean, invstd=invstd, act=act, slope=slope,
                           res=res, res_mode=res_mode, count=count, want_dres=want_dres, sinks=sinks,
                           sync=sync is not None, dz_add=dz_add, dres_add=dres_add, launch=not fused, head=head)
    if fused:
        # one persistent launch (reduce + grid barrier + apply; z held on chip): only from the step's MAIN stream -- its
        # workgroups wait for each other, two such kernels on concurrent streams could starve each other of compute units
        if _bn_bwd_fused(st):
            return _bn_bwd_result(st)
        if st.reduce_pending:
            st.reduce_pending = False
            _run(st.lib, 'ledn_bn_act_bwd_reduce', z, st.d, work=_ops._TIMING is not None and (
                f'bnbwd_reduce C{st.Cc} P{st.P}', _nb(z, dy, res), 6 * z.numel()))
    bn_act_bwd_sync(st, sync)
    return bn_act_bwd_apply(st)


HEAD_BWD = _knob_int('LEDN_HEAD_BWD', 1)      # the two-pass backward of LEDHead's two-class heads (csrc/head_bwd.hip)


def head_bwd_ok(x, dz, w, stride, pad, groups, act):
    """the norm -> act -> 3x3 conv backward can take the two-pass form of csrc/head_bwd.hip (include/ledn.h
    ledn_head_bwd_supported states the same gate)"""
    return bool(HEAD_BWD and x.dtype == torch.bfloat16 and dz.dtype == torch.bfloat16 and tuple(w.shape[2:]) == (3, 3)
                and w.shape[0] == 2 and w.shape[1] == 32 and x.shape[-1] == 32 and stride == 1 and pad == 1
                and groups == 1 and act in (ACT_NONE, _ops.ACT_RELU, _ops.ACT_PRELU) and x.numel() // x.shape[-1] >= 16384
                and x.numel() < (1 << 31) and dz.shape[1:3] == x.shape[1:3])


# Depthwise / pyramid WEIGHT gradients of the main stream on an auxiliary stream (experimental knob, 0 = off): they have no
# consumer inside the step (they reduce into the filter banks' gradient sinks, read by the flush at the end of the backward),
# run at ~1-2 TB/s (latency-bound stencil reductions) and so fill the tails of the data-gradient -> BatchNorm-backward chain
# without competing with it for HBM the way the convolution weight gradients did (LEDN_WGRAD_SLOT).
DW_WGRAD_SLOT = _knob_int('LEDN_DW_WGRAD_SLOT', 0)


def _on_side_stream(ref, sink, *keep):
    """context manager: the auxiliary stream for a sink-bound weight gradient launched from the main stream, else a no-op"""
    import contextlib
    if not (DW_WGRAD_SLOT and _ops.MULTI_STREAM and ref.is_cuda and sink is not None and _ops._slot(ref) == 0):
        return contextlib.nullcontext()
    cur = torch.cuda.current_stream(ref.device)
    side = _ops._aux_stream(ref.device, DW_WGRAD_SLOT)
    side.wait_stream(cur)
    for t in keep:
        if isinstance(t, torch.Tensor):
            t.record_stream(side)
    return torch.cuda.stream(side)


def _dw_desc(x_shape, dz, w_khwc, stride, pad, dil, group_size, ext1, dtype):
    N, H, W, Cc = x_shape
    KH, KW, _ = w_khwc.shape
    d = _lib.DwBwdDesc()
    d.N, d.H, d.W, d.C, d.Ho, d.Wo = N, H, W, Cc, dz.shape[1], dz.shape[2]
    d.KH, d.KW, d.stride, d.pad = KH, KW, stride, pad
    for i in range(4):
        d.dil[i] = dil[i] if i < len(dil) else dil[-1]
    d.group_size, d.ext1, d.dtype = group_size or Cc, int(ext1), dtype
    return d


def dwconv2d_bwd(x, dz, w_khwc, *, stride=1, pad=-1, dil=(1, 1, 1, 1), group_size=None, ext1=False,
                 add=None, need_dx=True, need_dw=True, dw_out=None):
    """-> (dx or None, dw [KH,KW,C] or None)."""
    lib = _lib.get_lib()
    if dz.dtype != x.dtype:
        raise LednError('dwconv2d_bwd: dz dtype must match x')
    dx = dw = None
    d = _dw_desc(x.shape, dz, w_khwc, stride, pad, dil, group_size, ext1, _dt(x))
    _check(lib, x, dz, w_khwc, add)
    d.x, d.dz, d.w, d.add = _p(x), _p(dz), _p(_f32(w_khwc)), _p(add)
    KH, KW, Cc = w_khwc.shape
    if need_dx:
        dx = torch.empty_like(x)
        d.dx = _p(dx)
        _run(lib, 'ledn_dwconv2d_bwd_data', x, d, work=_ops._TIMING is not None and (f'dwbwd_data{KH}x{KW} C{Cc} {tuple(x.shape)}', _nb(dz, dx, add), 2 * dz.numel() * KH * KW))
    if need_dw:
        # dw_out: an f32 buffer of the bank's shape the gradient is ACCUMULATED into (the trainer's bank-gradient sink)
        dw = dw_out if dw_out is not None else _ops.zeros_f32(tuple(w_khwc.shape), x.device)
        if tuple(dw.shape) != tuple(w_khwc.shape) or dw.dtype != torch.float32:
            raise LednError('dwconv2d_bwd: dw_out shape/dtype mismatch')
        _check(lib, dw)
        d.dw = _p(dw)
        with _on_side_stream(x, dw_out, x, dz, dw):
            _run(lib, 'ledn_dwconv2d_bwd_weight', x, d, work=_ops._TIMING is not None and (f'dwbwd_w{KH}x{KW} C{Cc} {tuple(x.shape)}', _nb(x, dz) , 2 * dz.numel() * KH * KW))
    return dx, dw


def _dwpack_desc(lib, weights, stacked, sinks=None):
    d = _lib.DwPackDesc()
    if not 0 < len(weights) <= 8:
        raise LednError('dw pack: 1..8 filters')
    KH, KW = weights[0].shape[2:]
    for k, w in enumerate(weights):
        if w.dim() != 4 or w.shape[1] != 1 or tuple(w.shape[2:]) != (KH, KW) or w.dtype != torch.float32:
            raise LednError('dw pack: depthwise filters [n,1,KH,KW] f32 of one kernel size expected')
        if stacked and w.shape[0] != weights[0].shape[0]:
            raise LednError('dw pack: stacked banks need equal channel counts')
        _check(lib, w)
        d.w[k], d.n[k] = w.data_ptr(), w.shape[0]
        if sinks is not None:
            g = sinks[k]
            if g.shape != w.shape or g.dtype != torch.float32:
                raise LednError('dw pack: gradient sink shape/dtype mismatch')
            _check(lib, g)
            d.dw[k] = g.data_ptr()
    d.nsrc, d.taps, d.stacked = len(weights), KH * KW, int(stacked)
    return d, KH, KW


def dw_pack(weights, stacked):
    """[n_k,1,KH,KW] depthwise filters -> [KH,KW,sum n] (stacked=False) or [K,KH,KW,n] (stacked=True)."""
    lib = _lib.get_lib()
    d, KH, KW = _dwpack_desc(lib, weights, stacked)
    n0 = weights[0].shape[0]
    shape = (len(weights), KH, KW, n0) if stacked else (KH, KW, sum(w.shape[0] for w in weights))
    out = torch.empty(shape, dtype=torch.float32, device=weights[0].device)
    _run(lib, 'ledn_dw_pack', out, d, _p(out), work=_ops._TIMING is not None and (f'dwpack {shape}', 8 * out.numel(), 0))
    return out


def dw_unpack_grad(weights, sinks, dpacked, stacked):
    """sinks[k] ([n_k,1,KH,KW] f32) += the gradient of the packed bank, in PyTorch's layout."""
    lib = _lib.get_lib()
    d, KH, KW = _dwpack_desc(lib, weights, stacked, sinks)
    if dpacked.dtype != torch.float32 or dpacked.numel() != sum(w.numel() for w in weights):
        raise LednError('dw_unpack_grad: packed gradient size/dtype mismatch')
    _check(lib, dpacked)
    _run(lib, 'ledn_dw_unpack_grad', dpacked, d, _p(dpacked),
         work=_ops._TIMING is not None and (f'dwunpack {tuple(dpacked.shape)}', 12 * dpacked.numel(), 0))


def sesp_pyramid_bwd(x, dy, w_b33n, dil, stride, dw_out=None):
    """-> (dx [N,H,W,n], dw [4,3,3,n]); dw_out: f32 buffer the weight gradient is accumulated into."""
    lib = _lib.get_lib()
    N, H, W, n = x.shape
    if dy.dtype != x.dtype:
        raise LednError('sesp_pyramid_bwd: dtype mismatch')
    d = _lib.PyrBwdDesc()
    gsum = torch.empty_like(dy)
    dx = torch.empty_like(x)
    dw = dw_out if dw_out is not None else _ops.zeros_f32(tuple(w_b33n.shape), x.device)
    if tuple(dw.shape) != tuple(w_b33n.shape) or dw.dtype != torch.float32:
        raise LednError('sesp_pyramid_bwd: dw_out shape/dtype mismatch')
    _check(lib, x, dy, w_b33n, dw)
    d.x, d.dy, d.w, d.gsum, d.dx, d.dw = _p(x), _p(dy), _p(_f32(w_b33n)), _p(gsum), _p(dx), _p(dw)
    d.N, d.H, d.W, d.n, d.Ho, d.Wo, d.stride = N, H, W, n, dy.shape[1], dy.shape[2], stride
    for i in range(4):
        d.dil[i] = dil[i]
    d.dtype = _dt(x)
    _run(lib, 'ledn_sesp_pyramid_bwd_data', x, d, work=_ops._TIMING is not None and (f'pyrbwd_data n{n} {tuple(x.shape)}', _nb(dy, gsum, gsum, dx), 2 * dy.numel() * 9))
    with _on_side_stream(x, dw_out, x, dy, gsum, dw):
        _run(lib, 'ledn_sesp_pyramid_bwd_weight', x, d, work=_ops._TIMING is not None and (f'pyrbwd_w n{n} {tuple(x.shape)}', _nb(x, gsum), 2 * dy.numel() * 9))
    return dx, dw


def bilinear_bwd(dy, in_hw, out_dtype=None):
    lib = _lib.get_lib()
    N, Ho, Wo, Cc = dy.shape
    H, W = in_hw
    dx = torch.empty((N, H, W, Cc), dtype=out_dtype or dy.dtype, device=dy.device)
    _check(lib, dy, dx)
    _run(lib, 'ledn_bilinear_bwd', dy, _p(dy), _p(dx), N, H, W, Cc, Ho, Wo, _dt(dy), _dt(dx),
         work=_ops._TIMING is not None and (f'bilinear_bwd C{Cc} {Ho}x{Wo}->{H}x{W} N{N}', _nb(dy, dx), 8 * dy.numel()))
    return dx


def avgpool3x3s2_bwd(dy, in_hw, add=None):
    lib = _lib.get_lib()
    N, Ho, Wo, Cc = dy.shape
    H, W = in_hw
    dx = torch.empty((N, H, W, Cc), dtype=dy.dtype, device=dy.device)
    if add is not None and (add.shape != dx.shape or add.dtype != dx.dtype):
        raise LednError('avgpool3x3s2_bwd: add mismatch')
    _check(lib, dy, dx, add)
    _run(lib, 'ledn_avgpool3x3s2_bwd', dy, _p(dy), _p(add), _p(dx), N, H, W, Cc, Ho, Wo, _dt(dy),
         work=_ops._TIMING is not None and (f'avgpool_bwd C{Cc} {N}x{H}x{W}', _nb(dy, dx, add), 3 * dx.numel()))
    return dx


def window_attn_bwd(qkv, biasT, dout, heads, ws=8):
    """-> (dqkv in qkv.dtype, dbiasT f32 [heads, ws^2, ws^2])."""
    lib = _lib.get_lib()
    N, H, W, C3 = qkv.shape
    Cc = C3 // 3
    padded = (H % ws != 0) or (W % ws != 0)
    dq32 = (torch.zeros if padded else torch.empty)((N, H, W, C3), dtype=torch.float32, device=qkv.device)
    dbias = _ops.zeros_f32(tuple(biasT.shape), qkv.device)
    _check(lib, qkv, biasT, dout)
    nwin = N * ((H + ws - 1) // ws) * ((W + ws - 1) // ws)
    _run(lib, 'ledn_window_attn_bwd', qkv, _p(qkv), _p(_f32(biasT)), _p(dout), _p(dq32), _p(dbias), N, H, W,
         Cc, heads, ws, _dt(qkv), work=_ops._TIMING is not None and (f'wattn_bwd C{Cc} h{heads} {N}x{H}x{W}', _nb(qkv, dout, dq32), 10 * nwin * (ws * ws) ** 2 * Cc))
    if qkv.dtype != torch.float32:
        from .ops import affine_act
        dq32 = affine_act(dq32, out_dtype=qkv.dtype)
    return dq32, dbias


def getb_pool_bwd(dout, ws=8):
    lib = _lib.get_lib()
    N, H, W, Cc = dout.shape
    da = torch.empty_like(dout)
    _check(lib, dout, da)
    _run(lib, 'ledn_getb_pool_bwd', dout, _p(dout), _p(da), N, H, W, Cc, ws, _dt(dout),
         work=_ops._TIMING is not None and (f'getbpool_bwd C{Cc} {N}x{H}x{W}', _nb(dout, da), 18 * da.numel()))
    return da


def mfaf_gate_bwd(x, r, xl, ctx, affines, dout, act=ACT_NONE):
    """-> (dx_b, dr_b, ds, [dctx_k f32])."""
    lib = _lib.get_lib()
    N, H, W, Cc = x.shape
    d = _lib.MfafBwdDesc()
    dx, dr, ds = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    dctx = [_ops.zeros_f32(tuple(c.shape), c.device) if c.dtype == torch.float32 else torch.zeros_like(c) for c in ctx]
    keep = [x, r, xl, dout]
    d.x, d.r, d.xl, d.dout, d.dx, d.dr, d.ds = _p(x), _p(r), _p(xl), _p(dout), _p(dx), _p(dr), _p(ds)
    for k, c in enumerate(ctx):
        d.ctx[k], d.ctx_size[k], d.dctx[k] = _p(_f32(c)), c.shape[1], _p(dctx[k])
        keep.append(c)
    for k, (s, b) in enumerate(affines):
        d.scale[k], d.shift[k] = _p(_f32(s, Cc)), _p(_f32(b, Cc))
        keep += [s, b]
    _check(lib, *keep)
    d.N, d.H, d.W, d.C, d.dtype, d.act = N, H, W, Cc, _dt(x), act
    _run(lib, 'ledn_mfaf_gate_bwd', x, d, work=_ops._TIMING is not None and (f'mfafgate_bwd C{Cc} {N}x{H}x{W}', _nb(x, r, xl, dout, dx, dr, ds), 30 * x.numel()))
    return dx, dr, ds, dctx


def mfaf_bwd_combine(dx, dr, dxl, dpools):
    """in place: dx += dxa, dr += dxa, dxa = dxl + sum_k adjoint-adaptive-pool(dpools[k])."""
    lib = _lib.get_lib()
    N, H, W, Cc = dx.shape
    n = len(dpools)
    ptrs = (C.c_void_p * max(n, 1))(*[p.data_ptr() for p in dpools])
    sizes = (C.c_int * max(n, 1))(*[p.shape[1] for p in dpools])
    _check(lib, dx, dr, dxl, *dpools)
    _run(lib, 'ledn_mfaf_bwd_combine', dx, _p(dx), _p(dr), _p(dxl), ptrs, sizes, n, N, H, W, Cc, _dt(dx),
         work=_ops._TIMING is not None and (f'mfaf_combine C{Cc} {N}x{H}x{W}', _nb(dx, dr, dxl) * 2, 8 * dx.numel()))
    return dx, dr


def ohem_ce_fwd(logits, target, thres, min_kept, loss_weight, ignore_label=255):
    """logits [N,H,W,C] f32, target [N,H,W] int64 -> (out[4] = loss, acc, thr, nsel ; work)."""
    lib = _lib.get_lib()
    if logits.dtype != torch.float32 or target.dtype != torch.int64:
        raise LednError('ohem_ce: logits f32 NHWC and int64 target required')
    Cc = logits.shape[-1]
    P = logits.numel() // Cc
    if target.numel() != P:
        raise LednError('ohem_ce: target shape')
    work = torch.empty(lib.cdll.ledn_ohem_work_floats(P), dtype=torch.float32, device=logits.device)
    out = torch.empty(4, dtype=torch.float32, device=logits.device)
    _check(lib, logits, target, work, out)
    _run(lib, 'ledn_ohem_ce_fwd', logits, _p(logits), _p(target), P, Cc, thres, int(min_kept), loss_weight,
         ignore_label, _p(work), _p(out), work=_ops._TIMING is not None and (f'ohem_fwd P{P} C{Cc}', _nb(logits, target) + 5 * 8 * P, 30 * P))
    return out, work


def ohem_ce_bwd(logits, target, work, out, dloss, loss_weight, ignore_label=255):
    lib = _lib.get_lib()
    Cc = logits.shape[-1]
    P = logits.numel() // Cc
    dl = torch.empty_like(logits)
    dloss = dloss.reshape(1).to(torch.float32).contiguous()
    _check(lib, logits, target, work, out, dloss, dl)
    _run(lib, 'ledn_ohem_ce_bwd', logits, _p(logits), _p(target), P, Cc, ignore_label, _p(work), _p(out),
         _p(dloss), loss_weight, _p(dl), work=_ops._TIMING is not None and (f'ohem_bwd P{P} C{Cc}', _nb(logits, target, dl) + 4 * P, 20 * P))
    return dl


def ohem_ce_up_fwd(src, target, thres, min_kept, loss_weight, ignore_label=255):
    """OHEM-CE on bilinear(src -> target's H x W) without materialising the resized logits.
    src [N,Hs,Ws,2] f32, target [N,H,W] int64 -> (out[4], work)."""
    lib = _lib.get_lib()
    if src.dtype != torch.float32 or target.dtype != torch.int64 or src.dim() != 4 or src.shape[-1] != 2:
        raise LednError('ohem_ce_up: src f32 [N,Hs,Ws,2] and int64 target required')
    N, Hs, Ws, _ = src.shape
    if target.dim() != 3 or target.shape[0] != N:
        raise LednError('ohem_ce_up: target [N,H,W] required')
    H, W = int(target.shape[1]), int(target.shape[2])
    P = N * H * W
    work = torch.empty(lib.cdll.ledn_ohem_work_floats(P), dtype=torch.float32, device=src.device)
    out = torch.empty(4, dtype=torch.float32, device=src.device)
    _check(lib, src, target, work, out)
    _run(lib, 'ledn_ohem_ce_up_fwd', src, _p(src), N, Hs, Ws, H, W, _p(target), thres, int(min_kept), loss_weight,
         ignore_label, _p(work), _p(out),
         work=_ops._TIMING is not None and (f'ohem_up_fwd P{P}', _nb(src, target) + 5 * 8 * P, 40 * P))
    return out, work


def ohem_ce_up_bwd(src, target, work, out, dloss, loss_weight, ignore_label=255):
    """-> dsrc [N,Hs,Ws,2]: the adjoint of the exact 2x resize applied to the loss gradient (never materialised)."""
    lib = _lib.get_lib()
    N, Hs, Ws, _ = src.shape
    H, W = int(target.shape[1]), int(target.shape[2])
    if H != 2 * Hs or W != 2 * Ws:
        raise LednError('ohem_ce_up_bwd: exact 2x resize only')
    dsrc = torch.empty_like(src)
    dloss = dloss.reshape(1).to(torch.float32).contiguous()
    _check(lib, src, target, work, out, dloss, dsrc)
    _run(lib, 'ledn_ohem_ce_up_bwd', src, _p(src), N, Hs, Ws, H, W, _p(target), ignore_label, _p(work), _p(out),
         _p(dloss), loss_weight, _p(dsrc),
         work=_ops._TIMING is not None and (f'ohem_up_bwd P{N * H * W}', _nb(src, target, dsrc) + 4 * N * H * W, 30 * N * H * W))
    return dsrc


def ohem2_up_fwd(src0, src1, target, cfg0, cfg1, ignore_label=255):
    """Both OHEM-CE losses of LEDHead.loss_by_feat in one launch set (ledn_ohem2_up_fwd).  src0 / src1 [N,Hs,Ws,2]
    f32, target [N,H,W] int64, cfg_k = (thres, min_kept, loss_weight) -> (out [2,4], work)."""
    lib = _lib.get_lib()
    for src in (src0, src1):
        if src.dtype != torch.float32 or src.dim() != 4 or src.shape[-1] != 2 or src.shape != src0.shape:
            raise LednError('ohem2_up: two f32 [N,Hs,Ws,2] sources of one shape required')
    N, Hs, Ws, _ = src0.shape
    if target.dtype != torch.int64 or target.dim() != 3 or target.shape[0] != N:
        raise LednError('ohem2_up: int64 target [N,H,W] required')
    H, W = int(target.shape[1]), int(target.shape[2])
    if W % 4 or not 0 <= ignore_label <= 255:
        raise LednError('ohem2_up: W % 4 == 0 and ignore_label in [0, 255] required')
    P = N * H * W
    work = torch.empty(lib.cdll.ledn_ohem2_work_floats(P), dtype=torch.float32, device=src0.device)
    out = torch.empty((2, 4), dtype=torch.float32, device=src0.device)
    _check(lib, src0, src1, target, work, out)
    _run(lib, 'ledn_ohem2_up_fwd', src0, _p(src0), _p(src1), N, Hs, Ws, H, W, _p(target), cfg0[0], int(cfg0[1]), cfg0[2],
         cfg1[0], int(cfg1[1]), cfg1[2], ignore_label, _p(work), _p(out),
         work=_ops._TIMING is not None and (f'ohem2_up_fwd P{P}', _nb(src0, src1, target) + (1 + 5 * 8) * P, 80 * P,
                                            'ohem2_prob_kernel'))
    return out, work


def ohem2_up_bwd(src0, src1, hw, work, out, dloss0, dloss1, lw0, lw1, ignore_label=255):
    """-> (dsrc0, dsrc1): the adjoint of the exact 2x resize applied to both loss gradients (never materialised)."""
    lib = _lib.get_lib()
    N, Hs, Ws, _ = src0.shape
    H, W = hw
    if H != 2 * Hs or W != 2 * Ws:
        raise LednError('ohem2_up_bwd: exact 2x resize only')
    d0, d1 = torch.empty_like(src0), torch.empty_like(src1)
    g0 = dloss0.reshape(1).to(torch.float32).contiguous()
    g1 = dloss1.reshape(1).to(torch.float32).contiguous()
    _check(lib, src0, src1, work, out, g0, g1, d0, d1)
    _run(lib, 'ledn_ohem2_up_bwd', src0, _p(src0), _p(src1), N, Hs, Ws, H, W, ignore_label, _p(work), _p(out), _p(g0), _p(g1),
         lw0, lw1, _p(d0), _p(d1),
         work=_ops._TIMING is not None and (f'ohem2_up_bwd P{N * H * W}', _nb(src0, src1, d0, d1) + 9 * N * H * W,
                                            60 * N * H * W, 'ohem2_bwd_up2_kernel'))
    return d0, d1


def mfaf_ctx_fwd(pooled, seqs, training, stats1=None, momentum=0.1, tails=None, sync=None, world=1):
    """The four pooled-context MLPs of Muti_AFF in one launch sequence (ledn_mfaf_ctx_fwd).
    pooled: 4 f32 [N,S,S,C] maps; seqs: 4 x (conv1, bn1, conv2) modules.  -> (z2 list [N,S,S,C] f32, saved dict)
    tails (training): the 4 trailing BatchNorm modules -- their batch statistics are then formed here too:
    saved['bn2'][k] = [scale | shift | mean | invstd][C], running statistics updated.
    sync (training, data-parallel): object with all_reduce(src, dst); the sequence then runs in its phases with ONE
    all-reduce of the [4,2,Ci] (and [4,2,C]) statistics between them -- SyncBN over `world` ranks."""
    lib = _lib.get_lib()
    d = _lib.MfafCtxDesc()
    Cc = pooled[0].shape[-1]
    Ci = seqs[0][0].out_channels
    z1s, z2s, bn1s, keep = [], [], [], []
    for k, (pz, (c1, bn, c2)) in enumerate(zip(pooled, seqs)):
        if pz.dtype != torch.float32 or not pz.is_contiguous() or pz.shape[-1] != Cc:
            raise LednError('mfaf_ctx: contiguous f32 [N,S,S,C] pooled maps required')
        P = pz.numel() // Cc
        z1 = torch.empty((P, Ci), dtype=torch.float32, device=pz.device)
        z2 = torch.empty(pz.shape, dtype=torch.float32, device=pz.device)
        bn1 = torch.empty((4, Ci), dtype=torch.float32, device=pz.device)
        w1, w2 = _f32(c1.weight.detach()), _f32(c2.weight.detach())
        b1 = _f32(c1.bias.detach()) if c1.bias is not None else None
        b2 = _f32(c2.bias.detach()) if c2.bias is not None else None
        _check(lib, pz, z1, z2, bn1, w1, w2, b1, b2, bn.weight, bn.bias, bn.running_mean, bn.running_var)
        d.pooled[k], d.z1[k], d.z2[k], d.bn1[k] = _p(pz), _p(z1), _p(z2), _p(bn1)
        d.w1[k], d.b1[k], d.w2[k], d.b2[k] = _p(w1), _p(b1), _p(w2), _p(b2)
        d.gamma[k], d.beta[k] = _p(_f32(bn.weight.detach())), _p(_f32(bn.bias.detach()))
        d.running_mean[k], d.running_var[k] = _p(bn.running_mean), _p(bn.running_var)
        d.P[k] = P
        z1s.append(z1); z2s.append(z2); bn1s.append(bn1); keep += [w1, w2, b1, b2]
        eps, mom = bn.eps, (bn.momentum if bn.momentum is not None else momentum)
    d.C, d.Ci, d.momentum, d.eps = Cc, Ci, float(mom), float(eps)
    bn2s = None
    if training:
        if stats1 is None:
            stats1 = _ops.zeros_f32((4, 2, Ci), pooled[0].device)
        d.stats1 = _p(stats1)
        if tails is not None:
            bn2s = []
            stats2 = _ops.zeros_f32((4, 2, Cc), pooled[0].device)
            for k, bn in enumerate(tails):
                bn2 = torch.empty((4, Cc), dtype=torch.float32, device=pooled[0].device)
                _check(lib, bn2, bn.weight, bn.bias, bn.running_mean, bn.running_var)
                d.gamma2[k], d.beta2[k] = _p(_f32(bn.weight.detach())), _p(_f32(bn.bias.detach()))
                d.running_mean2[k], d.running_var2[k], d.bn2[k] = _p(bn.running_mean), _p(bn.running_var), _p(bn2)
                bn2s.append(bn2)
            d.stats2 = _p(stats2)
            keep.append(stats2)
    work = _ops._TIMING is not None and ('mfaf_ctx_fwd', 0, 0, 'mfaf_ctx_fwd2_kernel')
    if sync is None or not training:
        _run(lib, 'ledn_mfaf_ctx_fwd', pooled[0], C.byref(d), int(bool(training)), work=work)
    else:
        d.count_scale = float(world)
        for phase, st in ((1, stats1), (2, stats2 if tails is not None else None), (4, None)):
            if phase == 4 and tails is None:
                break
            d.phase = phase
            _run(lib, 'ledn_mfaf_ctx_fwd', pooled[0], C.byref(d), 1, work=work if phase == 2 else False)
            if st is not None:
                sync.all_reduce(st, st)
    return z2s, dict(z1=z1s, bn1=bn1s, bn2=bn2s)


def mfaf_ctx_bwd(pooled, saved, dz2, seqs, sinks, tails=None, sinks2=None, sync=None, world=1):
    """(sync / world: as mfaf_ctx_fwd -- the BatchNorm backward sums are all-reduced between the phases, the
    parameter gradients stay this rank's own, as torch.nn.SyncBatchNorm's)
    -> (dpooled list, grads list of 4 x [dw1, db1, dgamma, dbeta, dw2, db2 (, dgamma2, dbeta2)] -- None where a
    sink took it).  sinks: 4 x 6 f32 buffers (or None) the parameter gradients are accumulated into.
    tails: the forward ran the trailing BatchNorms too (saved['z2'] = its outputs, saved['bn2']) -- dz2 is then the
    gradient with respect to THEIR output; sinks2: 4 x (dgamma2, dbeta2) sinks."""
    lib = _lib.get_lib()
    d = _lib.MfafCtxBwdDesc()
    dev = pooled[0].device
    Cc = pooled[0].shape[-1]
    Ci = seqs[0][0].out_channels
    sums = _ops.zeros_f32((4, 2, Ci), dev)
    dps, grads, keep = [], [], []
    for k, (pz, (c1, bn, c2)) in enumerate(zip(pooled, seqs)):
        P = pz.numel() // Cc
        g = torch.empty((P, Ci), dtype=torch.float32, device=dev)
        dp = torch.empty_like(pz)
        dz = dz2[k]
        if dz.dtype != torch.float32 or not dz.is_contiguous() or dz.numel() != pz.numel():
            raise LednError('mfaf_ctx_bwd: contiguous f32 dz2 of the pooled shape required')
        outs = []
        for j, prm in enumerate((c1.weight, c1.bias, bn.weight, bn.bias, c2.weight, c2.bias)):
            if prm is None:
                outs.append(None)
                continue
            sk = sinks[k][j] if sinks is not None else None
            outs.append(sk if sk is not None else _ops.zeros_f32(tuple(prm.shape), dev))
        w1, w2 = _f32(c1.weight.detach()), _f32(c2.weight.detach())
        _check(lib, pz, dz, g, dp, w1, w2, saved['z1'][k], saved['bn1'][k], *[t for t in outs if t is not None])
        d.pooled[k], d.z1[k], d.dz2[k], d.w1[k], d.w2[k] = _p(pz), _p(saved['z1'][k]), _p(dz), _p(w1), _p(w2)
        d.bn1[k], d.g[k], d.dpooled[k] = _p(saved['bn1'][k]), _p(g), _p(dp)
        d.dw1[k], d.db1[k], d.dgamma[k], d.dbeta[k], d.dw2[k], d.db2[k] = (_p(t) for t in outs)
        d.P[k] = P
        dps.append(dp)
        grads.append([None if (sinks is not None and sinks[k][j] is not None) else outs[j] for j in range(6)])
        keep += [g, w1, w2]
    d.sums, d.C, d.Ci = _p(sums), Cc, Ci
    if tails is not None:
        sums2 = _ops.zeros_f32((4, 2, Cc), dev)
        for k, bn in enumerate(tails):
            scr = torch.empty_like(pooled[k])
            outs2 = []
            for j, prm in enumerate((bn.weight, bn.bias)):
                sk = sinks2[k][j] if sinks2 is not None else None
                outs2.append(sk if sk is not None else _ops.zeros_f32(tuple(prm.shape), dev))
            _check(lib, saved['z2'][k], saved['bn2'][k], scr, *outs2)
            d.z2[k], d.bn2[k], d.dz2s[k] = _p(saved['z2'][k]), _p(saved['bn2'][k]), _p(scr)
            d.dgamma2[k], d.dbeta2[k] = _p(outs2[0]), _p(outs2[1])
            grads[k] += [None if (sinks2 is not None and sinks2[k][j] is not None) else outs2[j] for j in range(2)]
            keep += [scr]
        d.sums2 = _p(sums2)
        keep.append(sums2)
    work = _ops._TIMING is not None and ('mfaf_ctx_bwd', 0, 0, 'mfaf_ctx_bwd1_kernel')
    if sync is None:
        _run(lib, 'ledn_mfaf_ctx_bwd', pooled[0], C.byref(d), work=work)
    else:
        d.count_scale = float(world)
        sums_l = _ops.zeros_f32((4, 2, Ci), dev)
        d.sums_local = _p(sums_l)
        sums2_l = None
        if tails is not None:
            sums2_l = _ops.zeros_f32((4, 2, Cc), dev)
            d.sums2_local = _p(sums2_l)
        for phase, (src, dst) in ((1, (sums2_l, sums2 if tails is not None else None)), (2, (sums_l, sums)), (4, (None, None))):
            if phase == 1 and tails is None:
                continue
            d.phase = phase
            _run(lib, 'ledn_mfaf_ctx_bwd', pooled[0], C.byref(d), work=work if phase == 4 else False)
            if src is not None:
                sync.all_reduce(src, dst)
    return dps, grads


class SgdTable:
    """Device table of (param, grad, momentum) pointers for ledn_sgd_step."""

    def __init__(self, params, grads, moms):
        lib = _lib.get_lib()
        n = len(params)
        host = (_lib.SgdEntry * n)()
        self.max_n = 0
        for i, (p, g, m) in enumerate(zip(params, grads, moms)):
            for t in (p, g, m):
                if t.dtype != torch.float32 or not t.is_contiguous():
                    raise LednError('SGD tensors must be contiguous float32')
            _check(lib, p, g, m)
            host[i].p, host[i].g, host[i].m, host[i].n = p.data_ptr(), g.data_ptr(), m.data_ptr(), p.numel()
            self.max_n = max(self.max_n, p.numel())
        raw = bytes(host)
        t = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
        self.table = t.to(params[0].device)
        self.n = n
        self.ref = params[0]
        self.keep = (params, grads, moms)

    def step(self, lr, momentum, weight_decay, grad_scale=1.0, lr_dev=None):
        lib = _lib.get_lib()
        _run(lib, 'ledn_sgd_step', self.ref, self.table.data_ptr(), self.n, self.max_n, float(lr), _p(lr_dev),
             momentum, weight_decay, grad_scale, work=_ops._TIMING is not None and (f'sgd {self.n} tensors', 16 * sum(p.numel() for p in self.keep[0]), 0))


class PackTable:
    """Device table for ledn_pack_conv_weights_multi: all bf16 weight packs of the model
    (forward and data-gradient variants) refreshed by ONE launch per training step."""

    def __init__(self, entries):
        """entries: list of (w_param [Cout, Cin/g, KH, KW] f32, out bf16 buffer, mode, groups)"""
        lib = _lib.get_lib()
        n = len(entries)
        host = (_lib.PackEntry * n)()
        self.max_elems = 0
        for i, (w, out, mode, groups) in enumerate(entries):
            co, cig, kh, kw = w.shape
            _check(lib, w, out)
            host[i].w, host[i].out = w.data_ptr(), out.data_ptr()
            host[i].Cout, host[i].Cin, host[i].KK, host[i].mode, host[i].groups = co, cig * groups, kh * kw, mode, groups
            self.max_elems = max(self.max_elems, out.numel())
        self.table = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(entries[0][0].device)
        self.n, self.ref, self.keep = n, entries[0][0], entries

    def run(self):
        lib = _lib.get_lib()
        _run(lib, 'ledn_pack_conv_weights_multi', self.ref, self.table.data_ptr(), self.n, self.max_elems,
             work=_ops._TIMING is not None and (f'packw_multi {self.n} tensors', 0, 0))


class DwBankTable:
    """Device table for ledn_dw_repack_multi: every depthwise filter bank of the model (SESP pyramids and second
    passes, GETB 8x8) packed from the PyTorch-layout filters by ONE launch per step, and every bank gradient
    unpacked into the parameters' gradient views (and re-zeroed) by ONE launch at the end of the backward."""

    def __init__(self, banks):
        """banks: list of (weights [n_k,1,KH,KW] f32 params, gradient sinks (same shapes), stacked, packed, dpacked)"""
        lib = _lib.get_lib()
        host = (_lib.DwPackEntry * len(banks))()
        self.max_elems = 0
        for i, (weights, sinks, stacked, packed, dpacked) in enumerate(banks):
            d, _, _ = _dwpack_desc(lib, [w.detach() for w in weights], stacked, sinks)
            host[i].d = d
            _check(lib, packed, dpacked)
            if packed.dtype != torch.float32 or dpacked.shape != packed.shape or dpacked.dtype != torch.float32:
                raise LednError('DwBankTable: packed / dpacked must be float32 of one shape')
            host[i].packed, host[i].dpacked = packed.data_ptr(), dpacked.data_ptr()
            self.max_elems = max(self.max_elems, max(w.numel() for w in weights))
        self.table = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(banks[0][3].device)
        self.n, self.ref, self.keep = len(banks), banks[0][3], banks

    def run(self, direction):
        lib = _lib.get_lib()
        _run(lib, 'ledn_dw_repack_multi', self.ref, self.table.data_ptr(), self.n, self.max_elems, int(direction),
             work=_ops._TIMING is not None and (f'dw_repack_multi {self.n} banks dir{direction}', 0, 0))
