"""Training path: forward with batch-statistics BatchNorm, hand-written backward
kernels wired through ``torch.autograd.Function`` (torch only routes tensors and
owns the tape), OHEM-CE loss, SGD + PolyLR, data-parallel gradient exchange.

Reference call stack: tools/train.py:99-106 -> mmengine IterBasedTrainLoop ->
model.train_step -> EncoderDecoder.loss (segmentors/encoder_decoder.py:161-185)
-> LEDHead.loss (decode_heads/decode_head.py:248-264, led_head.py:101-146) ->
OptimWrapper.update_params (backward + SGD step); SyncBN / DDP collectives are
implicit there and explicit here (RCCL via
torch.distributed on the [2,C] statistic buffers and the flat gradient buffer).
"""
import math
import os as _os

import torch
import torch.nn as nn
from torch.autograd import Function

from . import ops, ops_train as T
from .ops import ACT_NONE, ACT_PRELU, ACT_RELU, ACT_RELU6, RES_ADD, RES_GATE, RES_NONE

from ._env import knob_int as _knob_int  # noqa: E402
BN_MOMENTUM = 0.1


class _Env:
    """Process-wide training environment (SyncBN collective, world size)."""
    sync_bn = None      # _Collective (all_reduce(src, dst), group(ref)) or None
    world = 1
    head_acc = {}       # gradient fan-in accumulators (_Acc) of the stem tensors LEDHead reads: {'x1': .., 'x2': ..}
    grad_ready = None   # callable(tag) fired from the backward by GradReadyFn, or None
    out_stats = {}      # y.data_ptr() -> [2, C] per-channel sums of y left by the pass that wrote it (BNActFn out_stats -> conv_module norm_first)
    lazy_bn = {}        # dy.data_ptr() -> ops_train._BnBwd whose apply half the consumer of dz performs itself (BNActFn lazy_dz -> StemConvFn)
    ctx_fin = {}        # z2.data_ptr() -> (scale, shift, mean, invstd) of its trailing BatchNorm (MfafCtxFn -> MfafTailFn)


class _Collective:
    """Sum all-reduce of small f32 buffers across the data-parallel ranks, on the CURRENT launch stream.
    commset: rccl.CommSet (one communicator per launch stream: plain stream operations, capturable in the
    step's hipGraph); else torch.distributed (`dist`: RCCL through ProcessGroupNCCL on the GPU, gloo in the
    CPU tests)."""

    log = None      # tests / bench: a list that receives (slot, 'all_reduce' | 'group_begin' | 'group_end', numel)
                    # for every collective issued, in issue order (the per-communicator operation sequence that
                    # must be identical on every rank)

    def __init__(self, commset=None, dist=None):
        assert (commset is None) != (dist is None)
        self.commset, self.dist = commset, dist

    @staticmethod
    def _slot_of(ref, slot):
        return slot if slot is not None else (ops._slot(ref) if ref.is_cuda else 0)

    def _comm(self, ref, slot=None):
        return self.commset.get(ops._slot(ref) if slot is None else slot)

    def all_reduce(self, src, dst, slot=None):
        """dst = sum over ranks of src (dst may be src)."""
        if _Collective.log is not None:
            _Collective.log.append((self._slot_of(src, slot), 'all_reduce', int(src.numel())))
        if self.commset is not None:
            return self._comm(src, slot).all_reduce(src, dst)
        if dst is not src:
            dst.copy_(src)
        self.dist.all_reduce(dst)
        return dst

    def group(self, ref, slot=None):
        """context manager: the all-reduces inside are ONE launch (ncclGroupStart/End); BatchNorms whose
        statistics are ready together (MFAF's five context BNs) share it."""
        import contextlib
        inner = self._comm(ref, slot).group() if self.commset is not None else contextlib.nullcontext()
        if _Collective.log is None:
            return inner
        key = self._slot_of(ref, slot)

        @contextlib.contextmanager
        def logged():
            _Collective.log.append((key, 'group_begin', 0))
            with inner:
                yield
            _Collective.log.append((key, 'group_end', 0))
        return logged()


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


class _Acc:
    """Gradient fan-in without elementwise adds.  A forward tensor with k consumers is handed out as k aliases
    (FanoutFn).  In the backward each consumer's kernel that can take an addend (data gradient of a convolution:
    epilogue `res`; BatchNorm backward apply: dz_add / dres_add; average-pool adjoint: add) adds the partial
    gradient left by the consumers that ran before it and leaves the new partial sum here; FanoutFn.backward
    returns the last one.  Consumers on another stream, or whose kernel has no addend, simply return their own
    gradient and are summed the old way.  The _Acc object is handed to the consumers explicitly (`acc=`)."""

    counters = {'chained': 0, 'added': 0}     # diagnostics: fan-ins folded into a kernel / summed by an elementwise add

    def __init__(self):
        self.buf, self.stream, self.included, self.keep = None, None, set(), []

    @staticmethod
    def _stream(t):
        return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else 0

    def take(self, like):
        """the partial sum so far if the chain can continue on the current stream, else None"""
        if self.buf is None or self.stream != _Acc._stream(like) or self.buf.shape != like.shape \
                or self.buf.dtype != like.dtype:
            return None
        return self.buf

    def put(self, t, chained):
        """t: this consumer's output (with the previous partial folded in when `chained`)"""
        if chained or self.buf is None:
            self.buf, self.stream = t, _Acc._stream(t)
            self.included.add(t.data_ptr())
            self.keep.append(t)      # (alive until FanoutFn.backward: its address must not be reused meanwhile)


def _take(acc, like):
    return acc.take(like) if acc is not None else None


class FanoutFn(Function):
    """x -> k aliases of x, one per consumer (see _Acc)."""

    @staticmethod
    def forward(ctx, x, k, acc):
        ctx.acc = acc
        return tuple(x.view_as(x) for _ in range(k))

    @staticmethod
    def backward(ctx, *gs):
        acc = ctx.acc
        total = acc.buf
        for g in gs:
            if g is None:
                continue
            if total is not None and g.data_ptr() in acc.included:
                _Acc.counters['chained'] += g.data_ptr() != total.data_ptr()
                continue
            if total is not None:
                _Acc.counters['added'] += 1
            total = g if total is None else total + g        # (consumer outside the chain: other stream / no addend)
        acc.buf, acc.keep = None, []
        return total, None, None


FANIN_CHAIN = bool(_knob_int('LEDN_FANIN_CHAIN', 1))


def fanout(x, k):
    """-> (k aliases of x, the _Acc their consumers share -- or None: plain autograd fan-in)"""
    if not (FANIN_CHAIN and x.requires_grad and k > 1):
        return (x,) * k, None
    acc = _Acc()
    return FanoutFn.apply(x, k, acc), acc


class _Packs:
    """bf16 weight packs of the MFMA conv path.  First (eager) step: packed per call and
    recorded; afterwards the Trainer refreshes all of them with one table-driven launch at
    the start of every step and the conv Functions just pick up their buffer."""
    entries = {}     # (id(param), mode, groups) -> (param, buffer)
    table = None
    frozen = False

    @staticmethod
    def reset():
        _Packs.entries, _Packs.table, _Packs.frozen = {}, None, False


class _Sinks:
    """Gradient sinks: from its second step on the Trainer maps every live parameter to its view of
    the flat gradient buffer (zeroed by the SGD kernel); the backward kernels of conv / BatchNorm /
    PReLU then reduce straight into those views and hand autograd None, instead of returning a fresh
    tensor that AccumulateGrad adds in (one fill + one add kernel per parameter per step)."""
    map = {}

    @staticmethod
    def get(p):
        return _Sinks.map.get(id(p)) if isinstance(p, nn.Parameter) else None


class _DwBanks:
    """Depthwise filter banks (DwPackFn).  First (eager) step: packed per call and recorded.  Afterwards the Trainer
    refreshes ALL banks with one table-driven launch at the start of a step (DwPackFn.forward hands out the
    persistent packed buffer), the depthwise / pyramid weight-gradient kernels accumulate into the bank's
    persistent gradient buffer (found by the packed buffer's address), and one launch at the end of the backward
    adds every bank gradient into the filters' gradient views and re-zeroes it: 2 launches per step instead of
    ~70 (48 packs + 24 unpacks of a few KB each, ~5 us apiece on the critical path)."""
    entries = {}       # (ids of the filter params, stacked) -> dict(weights, stacked, packed, dpacked)
    by_ptr = {}        # packed.data_ptr() -> dpacked
    table = None
    frozen = False
    pending = False    # bank gradients accumulated and not yet unpacked

    @staticmethod
    def reset():
        _DwBanks.entries, _DwBanks.by_ptr, _DwBanks.table, _DwBanks.frozen, _DwBanks.pending = {}, {}, None, False, False

    @staticmethod
    def grad_sink(packed):
        """the persistent gradient buffer of the bank `packed` (or None: return the gradient to autograd)"""
        if not _DwBanks.frozen:
            return None
        d = _DwBanks.by_ptr.get(packed.data_ptr())
        if d is not None:
            _DwBanks.pending = True
        return d

    @staticmethod
    def flush():
        if _DwBanks.pending and _DwBanks.table is not None:
            _DwBanks.table.run(1)
            _DwBanks.pending = False


def get_pack(w, mode, groups):
    if not isinstance(w, nn.Parameter):
        return ops.pack_conv_weights(w, mode, groups)      # derived weights (stem im2col form)
    key = (id(w), mode, groups)
    if _Packs.frozen:
        e = _Packs.entries.get(key)
        if e is not None:
            return e[1]
        return ops.pack_conv_weights(w, mode, groups)
    buf = ops.pack_conv_weights(w.detach(), mode, groups)
    _Packs.entries[key] = (w, buf)
    return buf


# --------------------------------------------------------------------------- #
# autograd Functions (forward/backward = C-ABI kernels)
# --------------------------------------------------------------------------- #
class ConvFn(Function):
    """z = conv2d(x [+ xadd], w) + b, with per-channel sum/sumsq of z accumulated
    into `stats` ([2,Cout] f32, zeroed by the caller) by the kernel's epilogue."""

    @staticmethod
    def forward(ctx, x, w, b, xadd, stride, pad, groups, stats, out_dtype, defer=False, acc=None):
        st = (stats[0], stats[1]) if stats is not None else None
        wp = None
        if (xadd is None and x.dtype == torch.bfloat16 and ops.mfma_weight_ok(w, groups)
                and out_dtype in (None, torch.bfloat16)):
            wp = get_pack(w, 0, groups)
        # defer: the BatchNorm finalize is the very next ledn call and sums the statistic rows itself
        z = ops.conv2d(x, w, stride=stride, pad=pad, groups=groups, xadd=xadd, out_shift=b, stats=st,
                       out_dtype=out_dtype, w_bf16=wp, defer_stats=bool(defer) and _Env.sync_bn is None)
        ctx.save_for_backward(x, w, xadd)
        ctx.cfg = (stride, pad, groups, b is not None)
        ctx.sinks = (_Sinks.get(w), _Sinks.get(b))
        ctx.acc = acc if xadd is None else None
        return z

    @staticmethod
    def backward(ctx, dz):
        x, w, xadd = ctx.saved_tensors
        stride, pad, groups, has_b = ctx.cfg
        dz = _c(dz)
        sw, sb = ctx.sinks
        dw, db = _conv_wgrad(x, dz, tuple(w.shape), sw, sb, stride=stride, pad=pad, groups=groups, xadd=xadd,
                             bias=has_b)
        dx = None
        if ctx.needs_input_grad[0] or (xadd is not None and ctx.needs_input_grad[3]):
            dx = _conv_dgrad(dz, x, w, stride, pad, groups, ctx.acc)
        return (dx if ctx.needs_input_grad[0] else None, None if sw is not None else dw,
                None if sb is not None else db,
                dx if (xadd is not None and ctx.needs_input_grad[3]) else None, None, None, None, None, None, None, None)


class BNActFn(Function):
    """y = act(res_mode(BN_train(z), res)); statistics come from `stats` (filled by
    the producing kernel) or are reduced here; running stats updated in place."""

    @staticmethod
    def forward(ctx, z, gamma, beta, slope, res, stats, bn, act, res_mode, out_dtype, acc_res=None, acc_z=None, lazy_dz=False, out_stats=False):
        """lazy_dz: z's producer is StemConvFn, the only reader of dz -- the backward then runs the reduce half only and hands
        dy + the prepared descriptor on (_Env.lazy_bn); the stem's weight-gradient kernel forms dz itself"""
        Cc = z.shape[-1]
        count = z.numel() // Cc
        ctx.lazy = bool(lazy_dz and STEM_LAZY_BN and res is None and act in (ACT_NONE, ACT_RELU) and z.dtype == torch.bfloat16
                        and Cc == 32 and acc_z is None)
        if stats is None:
            stats = ops.zeros_f32((2, Cc), z.device)
            ops.channel_stats(z, stats=(stats[0], stats[1]), defer_stats=_Env.sync_bn is None)
        if _Env.sync_bn is not None:
            _Env.sync_bn.all_reduce(stats, stats)
            count *= _Env.world
        scale, shift, mean, invstd = ops.bn_finalize((stats[0], stats[1]), count, gamma, beta,
                                                     bn.running_mean, bn.running_var, BN_MOMENTUM, bn.eps)
        # out_stats: a BatchNorm reads y next (LEDHead's norm -> act -> conv on the stem maps): its batch statistics come out
        # of this pass (conv_module picks them up from _Env.out_stats by the tensor's address)
        st_y = None
        if out_stats and OUT_STATS and res is None and ops.affine_stats_ok(z, act, out_dtype):
            st_y = ops.zeros_f32((2, Cc), z.device)
        y = ops.affine_act(z, scale, shift, act=act, slope=slope, res=res, res_mode=res_mode,
                           out_dtype=out_dtype, stats=(st_y[0], st_y[1]) if st_y is not None else None)
        if st_y is not None:
            _Env.out_stats[y.data_ptr()] = st_y
        ctx.save_for_backward(z, res, scale, shift, mean, invstd, slope)
        ctx.cfg = (act, res_mode, count)
        ctx.sinks = (_Sinks.get(gamma), _Sinks.get(beta), _Sinks.get(slope))
        ctx.acc_res = acc_res if res is not None else None
        ctx.acc_z = acc_z
        return y

    @staticmethod
    def backward(ctx, dy):
        z, res, scale, shift, mean, invstd, slope = ctx.saved_tensors
        act, res_mode, count = ctx.cfg
        dy = _c(dy)
        if ctx.lazy and dy.dtype == torch.bfloat16:
            st = T.bn_act_bwd_reduce(z, dy, scale=scale, shift=shift, mean=mean, invstd=invstd, act=act, count=count,
                                     sinks=ctx.sinks, sync=_Env.sync_bn is not None, no_dz=True)
            T.bn_act_bwd_sync(st, _Env.sync_bn)
            _Env.lazy_bn.clear()                    # (one stem per model; an entry nobody collected -- frozen stem -- must not linger)
            _Env.lazy_bn[dy.data_ptr()] = st
            _, _, dgamma, dbeta, _ = T._bn_bwd_result(st)
            return dy, dgamma, dbeta, None, None, None, None, None, None, None, None, None, None, None
        want_dres = res is not None and ctx.needs_input_grad[4]
        prev = _take(ctx.acc_res, dy) if want_dres else None
        prev_z = _take(ctx.acc_z, z)
        dz, dres, dgamma, dbeta, dslope = T.bn_act_bwd(
            z, dy, scale=scale, shift=shift, mean=mean, invstd=invstd, act=act, slope=slope, res=res,
            res_mode=res_mode, count=count, want_dres=want_dres, sync=_Env.sync_bn, sinks=ctx.sinks, dres_add=prev,
            dz_add=prev_z)
        if want_dres and ctx.acc_res is not None:
            ctx.acc_res.put(dres, prev is not None)
        if ctx.acc_z is not None:
            ctx.acc_z.put(dz, prev_z is not None)
        return dz, dgamma, dbeta, dslope, dres, None, None, None, None, None, None, None, None, None


OUT_STATS = _knob_int('LEDN_OUT_STATS', 1)      # statistics of a BatchNorm's input from the affine pass that writes it (ledn_affine_desc.stat_sum)
STEM_LAZY_BN = _knob_int('LEDN_STEM_LAZY_BN', 1)   # the stem BatchNorm's apply half inside the stem weight-gradient kernel (ledn_stem_conv_wgrad_bn)
WGRAD_DEFER = _knob_int('LEDN_WGRAD_DEFER', 1)   # one summing launch for all weight gradients of a step (ops.WgradDefer)
WGRAD_SLOT_MAXPIX = _knob_int('LEDN_WGRAD_SLOT_MAXPIX', 0)   # with LEDN_WGRAD_SLOT: only layers of at most this many output pixels (the short, latency-bound 1/8-resolution ones)
WGRAD_SLOT = _knob_int('LEDN_WGRAD_SLOT', 0)   # auxiliary stream of the weight gradients, 0 = launch stream (measured r02n: 782 img/s on its own stream vs 892 on the launch stream; round-3 kernels: 1098 vs 1225)


def _conv_wgrad(x, dz, w_shape, sw, sb, **kw):
    """Weight (+bias) gradient of a convolution.  With gradient sinks it has no consumer inside the step
    (it reduces into the flat gradient buffer, read by the exchange / SGD only): it runs on its own HIP
    stream, concurrently with the data-gradient -> BatchNorm-backward chain that IS the critical path."""
    if (WGRAD_SLOT and ops.MULTI_STREAM and dz.is_cuda and sw is not None
            and (sb is not None or not kw.get('bias'))
            and (not WGRAD_SLOT_MAXPIX or dz.shape[0] * dz.shape[1] * dz.shape[2] <= WGRAD_SLOT_MAXPIX)
            and ops._slot(dz) == 0):
        cur = torch.cuda.current_stream(dz.device)
        side = ops._aux_stream(dz.device, WGRAD_SLOT)
        side.wait_stream(cur)
        for t in (x, dz, *kw.values()):
            if isinstance(t, torch.Tensor):
                t.record_stream(side)
        with torch.cuda.stream(side):
            ops.conv2d_wgrad(x, dz, w_shape, dw_out=sw, db_out=sb, **kw)
        return None, None
    return ops.conv2d_wgrad(x, dz, w_shape, dw_out=sw, db_out=sb, **kw)


def _conv_dgrad(dz, x, w, stride, pad, groups, acc=None):
    """data gradient of conv2d(x, w): MFMA path when the flipped pack applies (see ConvFn.backward).
    acc: gradient fan-in of x (_Acc): the partial gradient of x's other consumers enters as the epilogue's residual."""
    wp = None
    cin_f = w.shape[1] * groups     # dgrad kernel: "Cin" = Cout_f (16-multiple), "Cout" = Cin_f
    if (dz.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and w.shape[0] % 16 == 0
            and (cin_f % 16 == 0 or cin_f <= 8) and ops.mfma_weight_ok(w, groups)):
        wp = get_pack(w, 1, groups)
    # (addend: the MFMA kernel's epilogue residual; f32 activations: the direct kernel's.  The narrow bf16 direct
    #  kernels have no residual input -- their gradient joins the old way)
    prev = _take(acc, x) if ((wp is not None and cin_f % 8 == 0) or x.dtype == torch.float32) else None
    dx = ops.conv2d(dz, w, stride=stride, pad=pad, groups=groups, transposed=True,
                    out_hw=(x.shape[1], x.shape[2]), out_dtype=x.dtype, w_bf16=wp, res=prev,
                    res_mode=RES_ADD if prev is not None else RES_NONE)
    if acc is not None:
        acc.put(dx, prev is not None)
    return dx


HEAD_WGRAD = _knob_int('LEDN_HEAD_WGRAD', 1)     # the two-class heads' weight gradient inside their BatchNorm-backward reduce pass (csrc/head_bwd.hip)


class BNActConvFn(Function):
    """z = conv2d(act(BN_train(x)), w) + b with the BatchNorm + activation (none / ReLU / PReLU) folded
    into the convolution's input staging: act(BN(x)) is never written (one write + one read of the
    activation less, forward and backward).  backward: dy = dgrad(dz); dw = wgrad(pre(x), dz) with the
    same prologue; then the BatchNorm/activation backward on (x, dy)."""
    ACTS = (ACT_NONE, ACT_RELU, ACT_PRELU)

    @staticmethod
    def forward(ctx, x, gamma, beta, slope, w, b, stats_in, bn, act, stride, pad, groups, stats_out, out_dtype, acc=None):
        Cc = x.shape[-1]
        count = x.numel() // Cc
        if stats_in is None:
            stats_in = ops.zeros_f32((2, Cc), x.device)
            ops.channel_stats(x, stats=(stats_in[0], stats_in[1]), defer_stats=_Env.sync_bn is None)
        if _Env.sync_bn is not None:
            _Env.sync_bn.all_reduce(stats_in, stats_in)
            count *= _Env.world
        scale, shift, mean, invstd = ops.bn_finalize((stats_in[0], stats_in[1]), count, gamma, beta,
                                                     bn.running_mean, bn.running_var, BN_MOMENTUM, bn.eps)
        wp = None
        if x.dtype == torch.bfloat16 and ops.mfma_weight_ok(w, groups) and out_dtype in (None, torch.bfloat16):
            wp = get_pack(w, 0, groups)
        st = (stats_out[0], stats_out[1]) if stats_out is not None else None
        z = ops.conv2d(x, w, stride=stride, pad=pad, groups=groups, in_scale=scale, in_shift=shift, in_act=act,
                       in_slope=slope, out_shift=b, stats=st, out_dtype=out_dtype, w_bf16=wp,
                       defer_stats=st is not None and _Env.sync_bn is None)   # a BNActFn on stats_out follows
        ctx.save_for_backward(x, w, scale, shift, mean, invstd, slope)
        ctx.cfg = (act, count, stride, pad, groups, b is not None)
        ctx.sinks = (_Sinks.get(gamma), _Sinks.get(beta), _Sinks.get(slope), _Sinks.get(w), _Sinks.get(b))
        ctx.acc = acc
        return z

    @staticmethod
    def backward(ctx, dz):
        x, w, scale, shift, mean, invstd, slope = ctx.saved_tensors
        act, count, stride, pad, groups, has_b = ctx.cfg
        sg, sb_, ss, sw, sbias = ctx.sinks
        dz = _c(dz)
        # LEDHead's two-class heads: no dy tensor -- the BatchNorm backward recomputes it from dz, and its reduce pass also
        # forms the head's weight gradient (ops_train.bn_act_bwd head=)
        head = (dz, w) if T.head_bwd_ok(x, dz, w, stride, pad, groups, act) else None
        if head is not None and HEAD_WGRAD:
            dw = sw if sw is not None else torch.zeros_like(w, dtype=torch.float32)
            db = (sbias if sbias is not None else torch.zeros(w.shape[0], dtype=torch.float32, device=w.device)) if has_b else None
            head = (dz, w, dw, db)
        else:
            dw, db = _conv_wgrad(x, dz, tuple(w.shape), sw, sbias, stride=stride, pad=pad, groups=groups, in_scale=scale,
                                 in_shift=shift, in_act=act, in_slope=slope, bias=has_b)
        dy = _conv_dgrad(dz, x, w, stride, pad, groups) if head is None else None
        prev = _take(ctx.acc, x)
        dx, _, dgamma, dbeta, dslope = T.bn_act_bwd(x, dy, scale=scale, shift=shift, mean=mean, invstd=invstd,
                                                     act=act, slope=slope, count=count, sync=_Env.sync_bn,
                                                     sinks=(sg, sb_, ss), dz_add=prev, head=head)
        if ctx.acc is not None:
            ctx.acc.put(dx, prev is not None)
        return (dx, dgamma, dbeta, dslope, None if sw is not None else dw, None if sbias is not None else db,
                None, None, None, None, None, None, None, None, None)


# BNActConvFn where a BatchNorm(+act) output feeds exactly one convolution: 0 = off, 1 = the norm->act->conv
# modules of LEDHead only, 2 = also BasicBlock conv1->conv2 and the SESP expansion (LEDN_FUSE_BN_CONV)
FUSE_BN_INTO_CONV = _knob_int('LEDN_FUSE_BN_CONV', 1)   # measured: 0, 1, 2 within 0.3 % of each other (r01q)


class ActFn(Function):
    """y = act(x [+ xadd]) without normalisation (stage ReLUs, xa = x + r).  acc: gradient fan-in of x (_Acc)."""

    @staticmethod
    def forward(ctx, x, xadd, act, acc=None, acc_xadd=None):
        y = ops.affine_act(x, act=act, xadd=xadd)
        ctx.save_for_backward(x, xadd)
        ctx.act = act
        ctx.acc = acc if (act != ACT_NONE and xadd is None) else None
        ctx.acc_xadd = acc_xadd if xadd is not None else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, xadd = ctx.saved_tensors
        dy = _c(dy)
        if ctx.act == ACT_NONE:
            if ctx.acc_xadd is not None:          # the shortcut's gradient IS dy: first partial of that tensor's fan-in
                ctx.acc_xadd.put(dy, False)
            return dy, (dy if xadd is not None else None), None, None, None
        assert xadd is None
        prev = _take(ctx.acc, x)
        dz, _, _, _, _ = T.bn_act_bwd(x, dy, act=ctx.act, dz_add=prev)
        if ctx.acc is not None:
            ctx.acc.put(dz, prev is not None)
        return dz, None, None, None, None


class DwFn(Function):
    @staticmethod
    def forward(ctx, x, w, stride, pad, dil, group_size, ext1, stats):
        st = (stats[0], stats[1]) if stats is not None else None
        # (every DwFn with statistics is followed at once by its BatchNorm finalize: sesp, getb)
        z = ops.dwconv2d(x, w, stride=stride, pad=pad, dil=dil, group_size=group_size, ext1=ext1, stats=st,
                         defer_stats=st is not None and _Env.sync_bn is None)
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, pad, tuple(dil), group_size, ext1)
        return z

    @staticmethod
    def backward(ctx, dz):
        x, w = ctx.saved_tensors
        stride, pad, dil, gs, ext1 = ctx.cfg
        sink = _DwBanks.grad_sink(w)
        dx, dw = T.dwconv2d_bwd(x, _c(dz), w, stride=stride, pad=pad, dil=dil, group_size=gs, ext1=ext1,
                                need_dx=ctx.needs_input_grad[0], dw_out=sink)
        return dx, (None if sink is not None else dw), None, None, None, None, None, None


class DwPackFn(Function):
    """Depthwise filters in PyTorch's [n,1,KH,KW] layout -> the channel-last bank the depthwise
    kernels read (one launch); backward adds the bank's gradient straight into the trainer's
    gradient views (one launch) or, without sinks, hands autograd per-filter views of it.
    Steady state (see _DwBanks): forward returns the bank packed by the step's table launch and the bank gradient
    never comes back here (the consumers accumulate it into the bank's persistent buffer)."""

    @staticmethod
    def forward(ctx, stacked, *weights):
        ctx.stacked = stacked
        ctx.set_materialize_grads(False)        # steady state: no gradient arrives (see backward)
        ctx.save_for_backward(*weights)
        ctx.sinks = [_Sinks.get(w) for w in weights]
        key = (tuple(id(w) for w in weights), bool(stacked))
        if _DwBanks.frozen:
            e = _DwBanks.entries.get(key)
            if e is not None:
                return e['packed'].view_as(e['packed'])
        packed = T.dw_pack([w.detach() for w in weights], stacked)
        if not _DwBanks.frozen and all(isinstance(w, nn.Parameter) for w in weights):
            _DwBanks.entries[key] = dict(weights=weights, stacked=bool(stacked), shape=tuple(packed.shape))
        return packed

    @staticmethod
    def backward(ctx, dpacked):
        weights = ctx.saved_tensors
        if dpacked is None:       # every consumer accumulated into the bank's persistent gradient (_DwBanks.flush)
            return (None,) + (None,) * len(weights)
        dpacked = _c(dpacked)
        if all(s is not None for s in ctx.sinks):
            T.dw_unpack_grad(weights, ctx.sinks, dpacked, ctx.stacked)
            return (None,) + (None,) * len(weights)
        grads, c0 = [], 0
        for k, w in enumerate(weights):
            n = w.shape[0]
            g = dpacked[k] if ctx.stacked else dpacked[:, :, c0:c0 + n]
            grads.append(g.permute(2, 0, 1).unsqueeze(1))
            c0 += n
        return (None, *grads)


class PyrFn(Function):
    @staticmethod
    def forward(ctx, x, w, dil, stride):
        y = ops.sesp_pyramid(x, w, dil, stride)
        ctx.save_for_backward(x, w)
        ctx.cfg = (tuple(dil), stride)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        sink = _DwBanks.grad_sink(w)
        dx, dw = T.sesp_pyramid_bwd(x, _c(dy), w, ctx.cfg[0], ctx.cfg[1], dw_out=sink)
        return dx, (None if sink is not None else dw), None, None


class BilinearFn(Function):
    @staticmethod
    def forward(ctx, x, add, size, out_dtype):
        ctx.in_hw = (x.shape[1], x.shape[2])
        ctx.x_dtype = x.dtype
        ctx.has_add = add is not None
        return ops.bilinear(x, size, add=add, out_dtype=out_dtype)

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        dx = T.bilinear_bwd(dy, ctx.in_hw, out_dtype=ctx.x_dtype) if ctx.needs_input_grad[0] else None
        return dx, (dy if ctx.has_add else None), None, None


class AvgPoolFn(Function):
    @staticmethod
    def forward(ctx, x, acc=None):
        ctx.in_hw = (x.shape[1], x.shape[2])
        ctx.acc = acc
        ctx.like = x
        return ops.avgpool3x3s2(x)

    @staticmethod
    def backward(ctx, dy):
        prev = _take(ctx.acc, ctx.like)
        dx = T.avgpool3x3s2_bwd(_c(dy), ctx.in_hw, add=prev)
        if ctx.acc is not None:
            ctx.acc.put(dx, prev is not None)
        return dx, None


multi_pool = ops.multi_pool      # (the pool pyramid lives with the other tensor-level wrappers: the eval path uses it too)


class MultiPoolFn(Function):
    """adaptive average pools of xa to 4x4, 8x8, 16x16, 1x1 (f32)."""
    SIZES = (4, 8, 16, 1)

    @staticmethod
    def forward(ctx, xa):
        ctx.shape, ctx.dtype = xa.shape, xa.dtype
        return multi_pool(xa, MultiPoolFn.SIZES)

    @staticmethod
    def backward(ctx, *dps):
        dxa = torch.zeros(ctx.shape, dtype=ctx.dtype, device=dps[0].device)
        T.mfaf_bwd_combine(dxa, None, None, [_c(d) for d in dps])
        return dxa


class WindowAttnFn(Function):
    @staticmethod
    def forward(ctx, qkv, biasT, heads, ws):
        ctx.save_for_backward(qkv, biasT)
        ctx.cfg = (heads, ws)
        return ops.window_attn(qkv, biasT, heads, ws)

    @staticmethod
    def backward(ctx, dout):
        qkv, biasT = ctx.saved_tensors
        dqkv, dbias = T.window_attn_bwd(qkv, biasT, _c(dout), *ctx.cfg)
        return dqkv, dbias, None, None


class RelPosBiasFn(Function):
    """table [(2ws-1)^2, heads] -> biasT [heads, T, T] (ledn_relpos_bias); backward straight into the gradient sink"""

    @staticmethod
    def forward(ctx, table, index):
        ctx.save_for_backward(index)
        ctx.shape = tuple(table.shape)
        ctx.sink = _Sinks.get(table)
        return ops.relpos_bias(table.detach(), index)

    @staticmethod
    def backward(ctx, dbias):
        (index,) = ctx.saved_tensors
        dt = ctx.sink if ctx.sink is not None else ops.zeros_f32(ctx.shape, dbias.device)
        ops.relpos_bias_bwd(_c(dbias), index, dt)
        return (None if ctx.sink is not None else dt), None


class GetbPoolFn(Function):
    @staticmethod
    def forward(ctx, a, local, ws, acc_local=None):
        ctx.ws = ws
        ctx.acc_local = acc_local
        return ops.getb_pool(a, local, ws)

    @staticmethod
    def backward(ctx, dout):
        dout = _c(dout)
        if ctx.acc_local is not None:             # d local = dout: first partial of n1's gradient fan-in
            ctx.acc_local.put(dout, False)
        return T.getb_pool_bwd(dout, ctx.ws), dout, None, None


class MfafFrontFn(Function):
    """xa = x + r and its adaptive average pools to 4x4, 8x8, 16x16, 1x1 (Muti_AFF, classification/model_utils.py:
    402-423).  Backward: d xa = d(local branch) + the adjoint pools of the four context gradients, and because xa
    = x + r the same tensor is the gradient of x and of r: ONE kernel (ledn_mfaf_bwd_combine) adds it IN PLACE
    onto the partial gradients the gate kernel left for x and r (acc_x / acc_r, see _Acc) -- no separate
    elementwise gradient adds."""

    @staticmethod
    def forward(ctx, x, r, acc_x, acc_r):
        xa = ops.affine_act(x, act=ACT_NONE, xadd=r)
        ctx.like = x.detach()
        ctx.accs = (acc_x, acc_r)
        return (xa,) + multi_pool(xa, MultiPoolFn.SIZES)

    @staticmethod
    def backward(ctx, dxl, *dps):
        like = ctx.like
        acc_x, acc_r = ctx.accs
        dps = [_c(d) for d in dps]
        dxl = _c(dxl) if dxl is not None else None
        px, pr = _take(acc_x, like), _take(acc_r, like)
        if px is not None and pr is not None and px.data_ptr() != pr.data_ptr():
            T.mfaf_bwd_combine(px, pr, dxl, dps)          # px += dxa, pr += dxa
            acc_x.put(px, True)
            acc_r.put(pr, True)
            return px, pr, None, None
        dxa = torch.zeros(like.shape, dtype=like.dtype, device=like.device)
        T.mfaf_bwd_combine(dxa, None, dxl, dps)
        return dxa, dxa, None, None


def _bn_stats_group(raws, bns, gammas, betas, given=None):
    """Batch statistics of several BatchNorms whose inputs are all available: the channel sums of each
    (given[k]: already accumulated by the producer's epilogue), ONE grouped SyncBN all-reduce for all of
    them (when data-parallel), then the finalizes.  -> ([(scale, shift, mean, invstd)], [count])"""
    sync = _Env.sync_bn
    stats, counts = [], []
    for k, raw in enumerate(raws):
        Cc = raw.shape[-1]
        st = given[k] if given is not None else None
        if st is None:
            st = ops.zeros_f32((2, Cc), raw.device)
            # (deferred rows need the finalize as the very next ledn call: only without the collective, one by one)
            ops.channel_stats(raw, stats=(st[0], st[1]), defer_stats=False)
        stats.append(st)
        counts.append(raw.numel() // Cc * (_Env.world if sync is not None else 1))
    if sync is not None:
        with sync.group(raws[0]):
            for st in stats:
                sync.all_reduce(st, st)
    fin = [ops.bn_finalize((st[0], st[1]), counts[k], gammas[k], betas[k], bns[k].running_mean,
                           bns[k].running_var, BN_MOMENTUM, bns[k].eps) for k, st in enumerate(stats)]
    return fin, counts


def _bn_bwd_group(items):
    """items: list of kwargs of T.bn_act_bwd_reduce (each with 'z', 'dy').  All reduces, ONE grouped SyncBN
    all-reduce, all applies.  -> list of (dz, dres, dgamma, dbeta, dslope)."""
    sync = _Env.sync_bn
    sts = [T.bn_act_bwd_reduce(it.pop('z'), it.pop('dy'), sync=sync is not None, **it) for it in items]
    if sync is not None:
        with sync.group(sts[0].z):
            for st in sts:
                T.bn_act_bwd_sync(st, sync)
    return [T.bn_act_bwd_apply(st) for st in sts]


class MultiBNActFn(Function):
    """n independent y_k = act(BN_train(z_k)) whose inputs are all available (MFAF's five first-level
    context MLPs): one grouped SyncBN all-reduce per direction instead of n.  Used on the data-parallel
    path only; numerically the same as n BNActFn."""

    @staticmethod
    def forward(ctx, n, bns, act, given, *args):
        zs, gb = args[:n], args[n:]
        gammas, betas = gb[0::2], gb[1::2]
        fin, counts = _bn_stats_group(zs, bns, gammas, betas, given)
        ys, saved = [], []
        for k, z in enumerate(zs):
            scale, shift, mean, invstd = fin[k]
            ys.append(ops.affine_act(z, scale, shift, act=act))
            saved += [z, scale, shift, mean, invstd]
        ctx.save_for_backward(*saved)
        ctx.cfg = (n, act, counts)
        ctx.sinks = [(_Sinks.get(gammas[k]), _Sinks.get(betas[k]), None) for k in range(n)]
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        n, act, counts = ctx.cfg
        sv = ctx.saved_tensors
        items = [dict(z=sv[5 * k], dy=_c(dys[k]), scale=sv[5 * k + 1], shift=sv[5 * k + 2], mean=sv[5 * k + 3],
                      invstd=sv[5 * k + 4], act=act, count=counts[k], sinks=ctx.sinks[k]) for k in range(n)]
        outs = _bn_bwd_group(items)
        dgb = []
        for o in outs:
            dgb += [o[2], o[3]]
        return (None, None, None, None, *[o[0] for o in outs], *dgb)


class MfafTailFn(Function):
    """The five trailing BatchNorms (batch statistics) + sigmoid gate + blend of
    Muti_AFF (classification/model_utils.py:377-400,425-428) in one kernel."""

    @staticmethod
    def forward(ctx, x, r, xl, c1, c2, c3, xg, bns, out_relu, accs, *gb):
        raws = [xl, c1, c2, c3, xg]
        ctx.accs = accs or (None, None)
        affs, saved = [], []
        givens = [_Env.ctx_fin.pop(raw.data_ptr(), None) if k else None for k, raw in enumerate(raws)]
        if _Env.sync_bn is not None:
            own = [k for k in range(5) if givens[k] is None]
            f_own, c_own = _bn_stats_group([raws[k] for k in own], [bns[k] for k in own], [gb[2 * k] for k in own],
                                           [gb[2 * k + 1] for k in own])
            fin = list(givens)
            counts = [-(raw.numel() // raw.shape[-1]) for raw in raws]
            for k, f, c in zip(own, f_own, c_own):
                fin[k], counts[k] = f, c
        else:
            fin, counts = [], []
            for k, raw in enumerate(raws):
                Cc = raw.shape[-1]
                count = raw.numel() // Cc
                given = givens[k]
                if given is not None:           # statistics + finalize already done by MfafCtxFn
                    fin.append(given)
                    counts.append(-count)       # (negative: the BatchNorm backward is MfafCtxFn's too)
                    continue
                stats = ops.zeros_f32((2, Cc), raw.device)
                ops.channel_stats(raw, stats=(stats[0], stats[1]), defer_stats=True)
                bn = bns[k]
                fin.append(ops.bn_finalize((stats[0], stats[1]), count, gb[2 * k], gb[2 * k + 1],
                                           bn.running_mean, bn.running_var, BN_MOMENTUM, bn.eps))
                counts.append(count)
        for scale, shift, mean, invstd in fin:
            affs.append((scale, shift))
            saved += [scale, shift, mean, invstd]
        act = ACT_RELU if out_relu else ACT_NONE
        out = ops.mfaf_gate(x, r, xl, [c1, c2, c3, xg], affs, act=act)
        ctx.save_for_backward(x, r, xl, c1, c2, c3, xg, *saved)
        ctx.act, ctx.counts = act, counts
        ctx.sinks = [(_Sinks.get(gb[2 * k]), _Sinks.get(gb[2 * k + 1]), None) for k in range(5)]
        return out

    @staticmethod
    def backward(ctx, dout):
        x, r, xl, c1, c2, c3, xg = ctx.saved_tensors[:7]
        sv = ctx.saved_tensors[7:]
        raws = [xl, c1, c2, c3, xg]
        affs = [(sv[4 * k], sv[4 * k + 1]) for k in range(5)]
        dx, dr, ds, dctx = T.mfaf_gate_bwd(x, r, xl, [c1, c2, c3, xg], affs, _c(dout), act=ctx.act)
        gys = [ds] + dctx
        own = [k for k in range(5) if ctx.counts[k] > 0]
        outs = _bn_bwd_group([dict(z=raws[k], dy=gys[k], scale=sv[4 * k], shift=sv[4 * k + 1], mean=sv[4 * k + 2],
                                   invstd=sv[4 * k + 3], count=ctx.counts[k], sinks=ctx.sinks[k]) for k in own])
        draws, dgb = [None] * 5, [None] * 10
        for k, o in zip(own, outs):
            draws[k] = o[0]
            dgb[2 * k], dgb[2 * k + 1] = o[2], o[3]
        for k in range(5):
            if ctx.counts[k] < 0:               # MfafCtxFn runs this BatchNorm's backward: pass dy through
                draws[k] = gys[k]
        for acc, g in zip(ctx.accs, (dx, dr)):       # first partial gradient of x / r (MfafFrontFn adds d xa onto it)
            if acc is not None:
                acc.put(g, False)
        return (dx, dr, *draws, None, None, None, *dgb)


STEM_DIRECT = _knob_int('LEDN_STEM_DIRECT', 1)


class StemConvFn(Function):
    """z = conv3x3/s2 (3 -> 32) of the normalised planar batch + per-channel statistics of z (ledn_stem_conv), weight
    gradient from the same batch (ledn_stem_conv_wgrad); the input needs no gradient (ddrnet.py:123-130)."""

    @staticmethod
    def forward(ctx, x, w, pre, stats):
        s, b, mp, valid, pad_val = pre
        wp = ops.pack_conv_weights(ops.stem_weight_as_1x1(w.detach()), 0)
        z = ops.stem_conv(x, wp, s, b, mp, valid, pad_val, stats=(stats[0], stats[1]), defer_stats=_Env.sync_bn is None)
        ctx.save_for_backward(x, s, b, mp, valid)
        ctx.pad_val = pad_val
        ctx.sink = _Sinks.get(w)
        ctx.wshape = tuple(w.shape)
        return z

    @staticmethod
    def backward(ctx, dz):
        x, s, b, mp, valid = ctx.saved_tensors
        dw = ctx.sink if ctx.sink is not None else ops.zeros_f32(ctx.wshape, dz.device)
        dz = _c(dz)
        st = _Env.lazy_bn.pop(dz.data_ptr(), None)       # BNActFn(lazy_dz): dz is still the gradient of act(BN(z))
        ops.stem_conv_wgrad(x, dz, dw, s, b, mp, valid, ctx.pad_val, bn_desc=st.d if st is not None else None)
        return None, (None if ctx.sink is not None else dw), None, None


class GradReadyFn(Function):
    """Identity.  Its backward runs when every consumer of `x` has delivered its gradient, i.e. when all
    backward kernels downstream of this point have been issued: the Trainer starts the gradient all-reduce
    of the parameters behind it (`tag`) there, overlapped with the rest of the backward."""

    @staticmethod
    def forward(ctx, x, tag):
        ctx.tag = tag
        return x.view_as(x)

    @staticmethod
    def backward(ctx, dy):
        if _Env.grad_ready is not None:
            _Env.grad_ready(ctx.tag)
        return dy, None


def grad_ready(x, tag):
    return GradReadyFn.apply(x, tag) if (_Env.grad_ready is not None and x.requires_grad) else x


class OhemFn(Function):
    @staticmethod
    def forward(ctx, logits, target, thres, min_kept, loss_weight, ignore_label):
        out, work = T.ohem_ce_fwd(logits, target, thres, min_kept, loss_weight, ignore_label)
        ctx.save_for_backward(logits, target, work, out)
        ctx.cfg = (loss_weight, ignore_label)
        ctx.mark_non_differentiable(out)
        ctx.set_materialize_grads(False)        # (no zero tensor for the non-differentiable output)
        loss = out[0].clone()   # 0-dim view copy (4 bytes)
        return loss, out

    @staticmethod
    def backward(ctx, dloss, _dout):
        logits, target, work, out = ctx.saved_tensors
        dl = T.ohem_ce_bwd(logits, target, work, out, dloss, ctx.cfg[0], ctx.cfg[1])
        return dl, None, None, None, None, None


# --------------------------------------------------------------------------- #
# train-mode forward of the blocks (mirrors blocks.py; BatchNorm on batch statistics)
# --------------------------------------------------------------------------- #
def _stats(c, ref):
    return ops.zeros_f32((2, c), ref.device)


def relu(x, acc=None):
    return ActFn.apply(x, None, ACT_RELU, acc)


def conv_bn_act(x, conv, bn, act=ACT_NONE, slope=None, res=None, res_mode=RES_NONE, xadd=None, out_dtype=None,
                acc=None, acc_res=None, out_stats=False):
    """acc / acc_res: gradient fan-in (_Acc) of x / res when they are aliases handed out by fanout(); out_stats: a BatchNorm
    reads the output next (see BNActFn)"""
    st = _stats(conv.out_channels, x) if bn is not None else None
    z = ConvFn.apply(x, conv.weight, conv.bias, xadd, conv.stride[0], conv.padding[0], conv.groups, st,
                     out_dtype if bn is None else None, bn is not None, acc)
    if bn is None:
        assert act == ACT_NONE and res is None
        return z
    return BNActFn.apply(z, bn.weight, bn.bias, slope, res, st, bn, act, res_mode, out_dtype, acc_res, None, False, out_stats)


def bn_act(x, bn, act=ACT_NONE, slope=None, res=None, res_mode=RES_NONE, stats=None, out_dtype=None, acc_z=None):
    return BNActFn.apply(x, bn.weight, bn.bias, slope, res, stats, bn, act, res_mode, out_dtype, None, acc_z)


_ACT = {None: ACT_NONE, 'relu': ACT_RELU, 'relu6': ACT_RELU6}


def conv_module(m, x, act_override=None, res=None, res_mode=RES_NONE, out_dtype=None, acc=None, acc_res=None, out_stats=False):
    """blocks.ConvModule in training mode (acc / acc_res, out_stats: see conv_bn_act)."""
    act = _ACT[m.act] if act_override is None else act_override
    if m.norm_first:
        if FUSE_BN_INTO_CONV and act in BNActConvFn.ACTS:
            st_in = _Env.out_stats.pop(x.data_ptr(), None)     # left by the pass that wrote x (BNActFn out_stats), else None
            z = BNActConvFn.apply(x, m.bn.weight, m.bn.bias, None, m.conv.weight, m.conv.bias, st_in, m.bn, act,
                                  m.stride, m.padding, m.conv.groups, None, out_dtype, acc)
        else:
            z = ConvFn.apply(bn_act(x, m.bn, act), m.conv.weight, m.conv.bias, None, m.stride, m.padding,
                             m.conv.groups, None, out_dtype)
        return z if res is None else ActFn.apply(z, res, ACT_NONE)      # (res_mode ADD: the PPM shortcut)
    return conv_bn_act(x, m.conv, m.bn if m.with_norm else None, act, res=res, res_mode=res_mode,
                       out_dtype=out_dtype, acc=acc, acc_res=acc_res, out_stats=out_stats and m.with_norm)


def basic_block(m, x, final_relu=False, pre=None):
    """pre=(x alias, residual alias, acc): the caller already fanned x out (it has further consumers)"""
    if pre is not None:
        x, xr, acc = pre
    else:
        (x, xr), acc = fanout(x, 2)       # consumers: conv1, and the shortcut (identity residual or 1x1 downsample)
    if m.downsample is not None:
        res, acc_res = conv_bn_act(xr, m.downsample[0], m.downsample[1], acc=acc), None
    else:
        res, acc_res = xr, acc
    act = ACT_RELU if (m.act_out or final_relu) else ACT_NONE
    c1, c2 = m.conv1, m.conv2
    if (FUSE_BN_INTO_CONV >= 2 and not c1.norm_first and not c2.norm_first and c1.with_norm and c2.with_norm
            and _ACT[c1.act] in BNActConvFn.ACTS):
        # conv1 -> [BN1 + ReLU folded into conv2's input staging] -> conv2 -> BN2 (+res) -> act
        st1 = _stats(c1.conv.out_channels, x)
        z1 = ConvFn.apply(x, c1.conv.weight, c1.conv.bias, None, c1.conv.stride[0], c1.conv.padding[0],
                          c1.conv.groups, st1, None, False, acc)
        st2 = _stats(c2.conv.out_channels, x)
        z2 = BNActConvFn.apply(z1, c1.bn.weight, c1.bn.bias, None, c2.conv.weight, c2.conv.bias, st1, c1.bn,
                               _ACT[c1.act], c2.conv.stride[0], c2.conv.padding[0], c2.conv.groups, st2, None)
        return BNActFn.apply(z2, c2.bn.weight, c2.bn.bias, None, res, st2, c2.bn, act, RES_ADD, None, acc_res)
    out = conv_module(c1, x, acc=acc)
    return conv_module(c2, out, act_override=act, res=res, res_mode=RES_ADD, acc_res=acc_res)


def bottleneck(m, x, final_relu=False):
    """blocks.Bottleneck in training mode (basic_block.py:207-221)."""
    (x, xr), acc = fanout(x, 2)           # consumers: conv1, the shortcut
    if m.downsample is not None:
        res, acc_res = conv_bn_act(xr, m.downsample[0], m.downsample[1], acc=acc), None
    else:
        res, acc_res = xr, acc
    act = ACT_RELU if (m.act_out or final_relu) else ACT_NONE
    out = conv_module(m.conv2, conv_module(m.conv1, x, acc=acc))
    return conv_module(m.conv3, out, act_override=act, res=res, res_mode=RES_ADD, acc_res=acc_res)


def sesp(m, x):
    residual = (m.stride == 2 and not m.spatial) or (m.stride == 1 and m.nIn == m.nOut)
    xr, acc = x, None
    if residual:
        (x, xr), acc = fanout(x, 2)       # consumers: the projection conv, the block's shortcut
    o1 = conv_bn_act(x, m.proj_1x1.conv, m.proj_1x1.bn, ACT_PRELU, slope=m.proj_1x1.act.weight, acc=acc)
    w1 = DwPackFn.apply(True, *[d.conv.weight for d in m.spp_dw])
    p = PyrFn.apply(o1, w1, m.dil, m.stride)
    w2 = DwPackFn.apply(False, *[d.conv.weight for d in m.spp_dw_v2])
    exp = m.conv_1x1_exp
    if m.stride == 2 and not m.spatial:
        act3, slope3, res, acc_res = ACT_NONE, None, AvgPoolFn.apply(xr, acc), None
    else:
        act3, slope3 = ACT_PRELU, m.module_act.weight
        res = xr if (m.stride == 1 and m.nIn == m.nOut) else None
        acc_res = acc if res is not None else None
    rm = RES_ADD if res is not None else RES_NONE
    st = _stats(m.nOut, x)
    z = DwFn.apply(p, w2, 1, -1, [d + 1 for d in m.dil], m.n, False, st)   # its BN finalize comes next
    if FUSE_BN_INTO_CONV >= 2:
        # BN(cat) + PReLU folded into the expansion conv's input staging
        st3 = _stats(exp.conv.out_channels, x)
        z3 = BNActConvFn.apply(z, m.br_after_cat.bn.weight, m.br_after_cat.bn.bias, m.br_after_cat.act.weight,
                               exp.conv.weight, exp.conv.bias, st, m.br_after_cat.bn, ACT_PRELU,
                               exp.conv.stride[0], exp.conv.padding[0], exp.conv.groups, st3, None)
        return BNActFn.apply(z3, exp.bn.weight, exp.bn.bias, slope3, res, st3, exp.bn, act3, rm, None, acc_res)
    cat = bn_act(z, m.br_after_cat.bn, ACT_PRELU, slope=m.br_after_cat.act.weight, stats=st)
    return conv_bn_act(cat, exp.conv, exp.bn, act3, slope=slope3, res=res, res_mode=rm, acc_res=acc_res)


def cespb(m, x):
    for blk in m:
        x = sesp(blk, x)
    return x


class AvgPool2dFn(Function):
    @staticmethod
    def forward(ctx, x, k, stride, pad):
        ctx.cfg = ((x.shape[1], x.shape[2]), k, stride, pad)
        return ops.avgpool2d(x, k, stride, pad)

    @staticmethod
    def backward(ctx, dy):
        return ops.avgpool2d_bwd(_c(dy), *ctx.cfg), None, None, None


class GlobalPoolFn(Function):
    """x -> its per-image channel means [N,1,1,C] in x's dtype (AdaptiveAvgPool2d((1,1)))."""

    @staticmethod
    def forward(ctx, x):
        ctx.shape, ctx.dtype = x.shape, x.dtype
        g = ops.adaptive_avgpool(x, 1)
        return g if g.dtype == x.dtype else ops.affine_act(g, out_dtype=x.dtype)

    @staticmethod
    def backward(ctx, dg):
        dx = torch.zeros(ctx.shape, dtype=ctx.dtype, device=dg.device)
        dg = _c(dg)
        T.mfaf_bwd_combine(dx, None, None, [dg if dg.dtype == torch.float32 else ops.affine_act(dg, out_dtype=torch.float32)])
        return dx


def ppm(m, x):
    """blocks.PPM (DAPPM / PAPPM, utils/ppm.py:119-129,178-192) in training mode."""
    hw = (x.shape[1], x.shape[2])
    n = m.num_scales
    x_ = conv_module(m.scales[0], x)

    def up(i):
        pooled = AvgPool2dFn.apply(x, *m.pools[i - 1]) if i < n - 1 else GlobalPoolFn.apply(x)
        return conv_module(m.scales[i][1], pooled)
    if m.kind == 'dappm':
        feats = [x_]
        for i in range(1, n):
            feats.append(conv_module(m.processes[i - 1], BilinearFn.apply(up(i), feats[i - 1], hw, None)))
        cat = torch.cat(feats, dim=-1)
    else:
        ups = [BilinearFn.apply(up(i), x_, hw, None) for i in range(1, n)]
        cat = torch.cat([x_, conv_module(m.processes, torch.cat(ups, dim=-1))], dim=-1)
    return conv_module(m.compression, cat, res=conv_module(m.shortcut, x), res_mode=RES_ADD)


def getb(m, x):
    a = m.attn
    (x_n, x_r), acc_x = fanout(x, 2)                      # consumers: norm1, the attention shortcut
    n1 = bn_act(x_n, m.norm1, acc_z=acc_x)
    (n1q, n1l), acc_n1 = fanout(n1, 2)                    # consumers: the qkv conv, the local (pooled) mixing term
    qkv = ConvFn.apply(n1q, a.qkv[0].weight, None, None, 1, 0, 1, None, None, False, acc_n1)
    att = WindowAttnFn.apply(qkv, RelPosBiasFn.apply(a.relative_position_bias_table, a.relative_position_index), m.heads, m.ws)
    mix = GetbPoolFn.apply(att, n1l, m.ws, acc_n1)
    wdw = DwPackFn.apply(False, a.proj[0].weight)
    st = _stats(m.dim, x)
    z = DwFn.apply(mix, wdw, 1, (m.ws - 1) // 2, [1, 1, 1, 1], m.dim, True, st)
    pj = bn_act(z, a.proj[1], stats=st)
    x1 = ActFn.apply(ConvFn.apply(pj, a.proj[2].weight, None, None, 1, 0, 1, None, None), x_r, ACT_NONE, None, acc_x)
    (x1n, x1r), acc_x1 = fanout(x1, 2)                    # consumers: norm2, the MLP shortcut
    n2 = bn_act(x1n, m.norm2, acc_z=acc_x1)
    h = ActFn.apply(ConvFn.apply(n2, m.mlp.fc1.weight, m.mlp.fc1.bias, None, 1, 0, 1, None, None), None, ACT_RELU6)
    y = ConvFn.apply(h, m.mlp.fc2.weight, m.mlp.fc2.bias, None, 1, 0, 1, None, None)
    return ActFn.apply(y, x1r, ACT_NONE, None, acc_x1)


TEST_HOOKS = {}     # tests only: 'edge' -> a given SEAM edge map [N,h,w,1] f32 instead of the kernel's (freezes the
                    # percentile binarisation for the deterministic whole-step gradient test)
CTX_FORKS = _knob_int('LEDN_CTX_FORKS', 7)    # bit s: context branch of stage 3+s on the aux stream (round 3: 7 -> 1225 img/s, 3 -> 1170, 0 -> 1139)
SEAM_SLOT = _knob_int('LEDN_SEAM_SLOT', 0)     # stream of the SEAM edge map: 0 = main (measured 895 vs 878 img/s on its own stream; round 3: 1225 vs 1204)
MFAF_FORK = _knob_int('LEDN_MFAF_FORK', 0)   # measured: 785 vs 876 img/s with the forks in TRAINING (inference gains 4 %: blocks.MFAF)


class MfafCtxFn(Function):
    """The four pooled-context MLPs of Muti_AFF (conv1x1 + bias -> BatchNorm on batch statistics -> ReLU -> conv1x1 +
    bias) AND the batch statistics / backward of their trailing BatchNorms as one forward and one backward launch
    sequence (ledn_mfaf_ctx_fwd / _bwd) instead of ~18 tiny launches per scale and step.  The trailing BatchNorm's
    affine pair travels to the gate kernel through _Env.ctx_fin; MfafTailFn hands back the gradient with respect to
    the BatchNorm OUTPUT.  (Every tensor the backward needs goes through save_for_backward: holding the outputs on
    ctx as attributes is a reference cycle, and its collection inside a graph capture crashed the capture.)"""

    @staticmethod
    def forward(ctx, seqs, tails, *args):
        pooled, params = [_c(p) for p in args[:4]], args[4:]
        if not FUSE_MFAF_TAIL:
            tails = None
        ctx.sync, ctx.world = _Env.sync_bn, _Env.world
        z2s, saved = T.mfaf_ctx_fwd(pooled, seqs, True, tails=tails, sync=ctx.sync, world=ctx.world)
        ctx.seqs, ctx.tails = seqs, tails
        ctx.save_for_backward(*pooled, *saved['z1'], *saved['bn1'], *(saved['bn2'] or ()), *z2s)
        ctx.sinks = [[_Sinks.get(p) for p in (c1.weight, c1.bias, bn.weight, bn.bias, c2.weight, c2.bias)]
                     for (c1, bn, c2) in seqs]
        ctx.sinks2 = [[_Sinks.get(bn.weight), _Sinks.get(bn.bias)] for bn in tails] if tails is not None else None
        ctx.has = [p is not None for p in params]
        for z2, bn2 in zip(z2s, saved['bn2'] or ()):
            _Env.ctx_fin[z2.data_ptr()] = (bn2[0], bn2[1], bn2[2], bn2[3])
        return tuple(z2s)

    @staticmethod
    def backward(ctx, *dy):
        sv = ctx.saved_tensors
        pooled, z1, bn1 = list(sv[0:4]), list(sv[4:8]), list(sv[8:12])
        if ctx.tails is not None:
            saved = dict(z1=z1, bn1=bn1, bn2=list(sv[12:16]), z2=list(sv[16:20]))
        else:
            saved = dict(z1=z1, bn1=bn1)
        dps, grads = T.mfaf_ctx_bwd(pooled, saved, [_c(d) for d in dy], ctx.seqs, ctx.sinks, tails=ctx.tails,
                                    sinks2=ctx.sinks2, sync=ctx.sync, world=ctx.world)
        flat = []
        for gk in grads:
            flat += gk[:6]
        for gk in grads:
            flat += gk[6:] if ctx.tails is not None else [None, None]
        return (None, None, *dps, *[g if h else None for g, h in zip(flat, ctx.has)])


FUSE_MFAF_CTX = _knob_int('LEDN_FUSE_MFAF_CTX', 1)
FUSE_MFAF_TAIL = _knob_int('LEDN_FUSE_MFAF_TAIL', 1)   # ... and their trailing BatchNorms
FUSE_MFAF_SYNC = _knob_int('LEDN_FUSE_MFAF_SYNC', 1)   # ... under SyncBN too (phases + all-reduces in between)


def mfaf(m, x, r, out_relu=False, pre=None):
    """pre=(x alias for the gate, x alias for xa, acc): the caller already fanned x out (further consumers)"""
    if pre is not None:
        x, x_f, acc_x = pre
    else:
        (x, x_f), acc_x = fanout(x, 2)
    (r, r_f), acc_r = fanout(r, 2)
    front = MfafFrontFn.apply(x_f, r_f, acc_x, acc_r)
    xa, pooled = front[0], front[1:]

    def mlp(seq, off, inp, out_dtype=None):
        mid = conv_bn_act(inp, seq[off], seq[off + 1], ACT_RELU)
        c1 = seq[off + 3]
        return ConvFn.apply(mid, c1.weight, c1.bias, None, 1, 0, 1, None, out_dtype), seq[off + 4]
    ctx_seqs = [getattr(m, name) for name, _ in m.POOLS]
    fused = (FUSE_MFAF_CTX and all(pz.dtype == torch.float32 and pz.shape[-1] == 64 for pz in pooled)
             and all(sq[1].out_channels == 16 and sq[4].out_channels == 64 for sq in ctx_seqs))
    if _Env.sync_bn is not None and not (fused and FUSE_MFAF_SYNC):
        # data-parallel: the five first-level BatchNorms (local + four pooled contexts) see their inputs
        # together -> one grouped SyncBN all-reduce per direction instead of five
        seqs = [(m.local_att, 0, xa)] + [(getattr(m, name), 1, pz) for (name, _), pz in zip(m.POOLS, pooled)]
        sts = [_stats(seq[off].out_channels, xa) for seq, off, _ in seqs]
        zs = [ConvFn.apply(inp, seq[off].weight, seq[off].bias, None, seq[off].stride[0], seq[off].padding[0],
                           seq[off].groups, st, None, False) for (seq, off, inp), st in zip(seqs, sts)]
        bns1 = [seq[off + 1] for seq, off, _ in seqs]
        gb1 = []
        for bn in bns1:
            gb1 += [bn.weight, bn.bias]
        mids = MultiBNActFn.apply(5, bns1, ACT_RELU, sts, *zs, *gb1)
        raws = [ConvFn.apply(mid, seq[off + 3].weight, seq[off + 3].bias, None, 1, 0, 1, None, None)
                for mid, (seq, off, _) in zip(mids, seqs)]
        bns = [seq[off + 4] for seq, off, _ in seqs]
    else:
        # the four pooled-context MLPs are chains of tiny launch-bound kernels, independent of each other and of
        # the local branch: each runs on its own auxiliary stream while the local branch (full-resolution convs)
        # runs on the main one (forward here, backward through autograd's stream affinity)
        # (data-parallel: the fused sequence runs in phases with one all-reduce of the [4,2,C] statistics between them)
        forks, ctx, bns_ctx = [], [], []
        seqs = ctx_seqs
        if fused:
            triples = [(sq[1], sq[2], sq[4]) for sq in seqs]
            bns_ctx = [sq[5] for sq in seqs]
            params = []
            for c1, bn, c2 in triples:
                params += [c1.weight, c1.bias, bn.weight, bn.bias, c2.weight, c2.bias]
            for bn in bns_ctx:
                params += [bn.weight, bn.bias]
            ctx = list(MfafCtxFn.apply(triples, bns_ctx, *pooled, *params))
        for idx, ((name, _), pz) in enumerate(zip(m.POOLS, pooled) if not fused else ()):
            f = ops.Fork(pz, 3 + idx if MFAF_FORK else 0)
            with f:
                c, bn = mlp(getattr(m, name), 1, pz)
            forks.append((f, c))
            ctx.append(c)
            bns_ctx.append(bn)
        xl, bn_l = mlp(m.local_att, 0, xa)
        for f, c in forks:
            f.join(c)
        raws, bns = [xl] + ctx, [bn_l] + bns_ctx
    gb = []
    for bn in bns:
        gb += [bn.weight, bn.bias]
    return MfafTailFn.apply(x, r, *raws, bns, out_relu, (acc_x, acc_r), *gb)


def lednet_forward_train(m, x, pre=None):
    """LEDNet.forward in training mode -> (c3, c5, x1, x2) NCHW views."""
    from .lednet import to_nchw_view
    H, W = x.shape[2:]
    out_size = (math.ceil(H / 8), math.ceil(W / 8))
    s, b, mp, valid, pad_val = (tuple(pre) + (None, 0.0))[:5] if pre is not None else (None, None, None, None, 0.0)
    s0 = m.stem['0']
    _Env.out_stats = {}         # (per forward: an entry nobody collected must not meet another tensor at the same address)
    if (STEM_DIRECT and m.act_dtype == torch.bfloat16 and m.in_channels == 3 and m.channels == 32
            and x.dtype in (torch.uint8, torch.float32, torch.bfloat16) and s0.conv.bias is None):
        # the first stem convolution and its weight gradient straight from the planar batch (ledn_stem_conv /
        # ledn_stem_conv_wgrad): the [pixels][32] patch matrix of the form below (268 MB at 16 x 1024^2, written once and
        # read twice per step) is never materialised
        st = _stats(m.channels, x)
        z = StemConvFn.apply(x.contiguous(), s0.conv.weight, (s, b, mp, valid, pad_val), st)
        x1 = BNActFn.apply(z, s0.bn.weight, s0.bn.bias, None, None, st, s0.bn, ACT_RELU, RES_NONE, None, None, None, True, True)
    elif m.act_dtype == torch.bfloat16 and 9 * m.in_channels <= 32 and m.channels % 32 == 0:
        # stem as a K=32 GEMM on the MFMA path: im2col patches straight from the planar batch
        # (normalisation folded in; the input needs no gradient) x reshaped weight
        st = _stats(m.channels, x)
        z = ConvFn.apply(ops.im2col_stem_planar(x.contiguous(), s, b, mp, valid, pad_val), ops.stem_weight_as_1x1(s0.conv.weight),
                         None, None, 1, 0, 1, st, None, True)
        x1 = BNActFn.apply(z, s0.bn.weight, s0.bn.bias, None, None, st, s0.bn, ACT_RELU, RES_NONE, None)
    else:
        x1 = conv_module(s0, ops.nchw_to_nhwc(x.contiguous(), m.act_dtype, s, b, mp, valid, pad_val))
    # x1 and x2 also feed LEDHead's head_x1 / head_x2: fan them out here, the head picks its accumulators up
    # from _Env.head_acc (its consumer is a BNActConvFn whose BatchNorm-backward takes the addend)
    (x1, x1_head), acc1 = fanout(x1, 2)
    x2 = conv_module(m.stem['1'], x1, acc=acc1, out_stats=True)      # (head_x2's BatchNorm reads x2)
    (x2a, x2b, x2_head), acc2 = fanout(x2, 3)
    _Env.head_acc = {'x1': (acc1, x1_head.data_ptr()), 'x2': (acc2, x2_head.data_ptr())}
    x1, x2 = x1_head, x2_head
    y = basic_block(m.stem['2'][1], basic_block(m.stem['2'][0], None, pre=(x2a, x2b, acc2)), final_relu=True)
    y = basic_block(m.stem['4'][1], basic_block(m.stem['4'][0], y), final_relu=True)
    y = grad_ready(y, 'post_stem')     # backward: every parameter gradient outside the stem is complete here
    # context branch and SEAM edge map on auxiliary streams between the fusion points (ops.Fork);
    # autograd runs each backward kernel on the stream of its forward
    with ops.Fork(y, SEAM_SLOT) as fe, torch.no_grad():
        # the binarised edge map is piecewise constant: no gradient (ddrnet_speed.py:290-338)
        seg = conv_module(m.seam.conv_1, y.detach(), out_dtype=torch.float32)
        edge = ops.seam_edge(seg, m.seam.percentile, m.seam.fixed_threshold, 0.1)
        if TEST_HOOKS.get('edge') is not None:
            edge = TEST_HOOKS['edge'].to(edge.device, edge.dtype).reshape(edge.shape)
        if TEST_HOOKS.get('capture') is not None:
            TEST_HOOKS['capture']['edge'] = edge.detach().clone()
    # stage 3
    with ops.Fork(y, 1 if CTX_FORKS & 1 else 0) as f3:
        x_c = cespb(m.layer3, y)
        if m.getb_stage3:
            x_c = getb(m.getb1, x_c)
        comp = BilinearFn.apply(conv_module(m.compression_1, relu(x_c)), None, out_size, None)
    x_s = cespb(m.layer3_, y)
    f3.join(x_c, comp)
    (xs_d, xs_t, xs_f), acc_xs = fanout(x_s, 3)       # consumers: relu -> down_1, the MFAF gate, MFAF's xa = x + r
    x_c = conv_module(m.down_1, relu(xs_d, acc_xs), res=x_c, res_mode=RES_ADD)
    x_s = mfaf(m.aff1, None, comp, pre=(xs_t, xs_f, acc_xs))
    (c3, x_s), acc_c3 = fanout(x_s, 2)                # consumers: LEDHead's aux_head (c3), relu -> layer4_
    _Env.head_acc['c3'] = (acc_c3, c3.data_ptr())
    # stage 4
    with ops.Fork(x_c, 1 if CTX_FORKS & 2 else 0) as f4:
        x_c = cespb(m.layer4, relu(x_c))
        comp = BilinearFn.apply(conv_module(m.compression_2, relu(x_c)), None, out_size, None)
    x_s = cespb(m.layer4_, relu(x_s, acc_c3))
    (xs_d, xs_t, xs_f), acc_xs = fanout(x_s, 3)
    d = conv_module(m.down_2[0], relu(xs_d, acc_xs))
    f4.join(x_c, comp)
    x_c = conv_module(m.down_2[1], d, res=x_c, res_mode=RES_ADD)
    x_s = mfaf(m.aff2, None, comp, pre=(xs_t, xs_f, acc_xs))
    fe.join(edge)
    x_s = conv_module(m.seam.conv_2, edge, res=x_s, res_mode=RES_GATE, out_dtype=x_s.dtype)
    # stage 5
    with ops.Fork(x_c, 1 if CTX_FORKS & 4 else 0) as f5:
        x_c = cespb(m.layer5, relu(x_c))
        x_c = getb(m.getb2, conv_module(m.spp, x_c)) if m.context_tail == 'getb' else ppm(m.spp, x_c)
    x_s = sesp(m.layer5_, relu(x_s))
    f5.join(x_c)
    c5 = BilinearFn.apply(x_c, x_s, out_size, None)
    return tuple(to_nchw_view(t) for t in (c3, c5, x1, x2))


# --------------------------------------------------------------------------- #
# LEDHead training forward + loss (led_head.py:62-75,101-146)
# --------------------------------------------------------------------------- #
def _base_head(seq, x, out_dtype=None, acc=None):
    z = conv_module(seq[0], x, acc=acc)
    return bn_act(z, seq[1], ACT_RELU, out_dtype=out_dtype)


def led_head_forward_train(h, inputs):
    from .lednet import to_nhwc
    c3, c5, x1, x2 = (to_nhwc(t) for t in inputs)
    x1, x2 = grad_ready(x1, 'post_stem'), grad_ready(x2, 'post_stem')    # (the heads' share of that barrier)
    f32 = torch.float32
    xc = _base_head(h.head, c5)
    xc = ConvFn.apply(xc, h.conv_seg.weight, h.conv_seg.bias, None, 1, 0, 1, None, f32)
    accs, _Env.head_acc = _Env.head_acc, {}

    def acc_of(name, t):      # the backbone's fan-in accumulator of this very tensor (same storage), else None
        a = accs.get(name)
        return a[0] if (a is not None and a[1] == t.data_ptr()) else None
    xs = _base_head(h.aux_head, c3, None, acc_of('c3', c3))
    xs = ConvFn.apply(xs, h.aux_cls_seg.weight, h.aux_cls_seg.bias, None, 1, 0, 1, None, f32)
    h1 = _base_head(h.head_x1, x1, f32, acc_of('x1', x1))
    h2 = _base_head(h.head_x2, x2, f32, acc_of('x2', x2))
    return xc, xs, h1, h2


def fuse_loss_half(logit, h1, h2, hw):
    """led_head.py:106-131: the fusion pyramid up to floor(size/2); NHWC f32."""
    H, W = hw
    r = BilinearFn.apply(logit, h2, (H // 4, W // 4), None)
    return BilinearFn.apply(r, h1, (H // 2, W // 2), None)


def fuse_loss(logit, h1, h2, hw):
    """led_head.py:106-138: floor(size/4), floor(size/2), size; NHWC f32."""
    return BilinearFn.apply(fuse_loss_half(logit, h1, h2, hw), None, tuple(hw), None)


def ohem_loss(crit, score, target):
    """OhemCrossEntropy.forward on NCHW(-view) logits."""
    from .lednet import to_nhwc
    lg = to_nhwc(score, torch.float32)
    loss, _ = OhemFn.apply(lg, target.contiguous(), crit.thresh, crit.min_kept, crit.loss_weight,
                           crit.ignore_label)
    return loss


class OhemUpFn(Function):
    """OhemCrossEntropy on resize(src -> label size) with the resize folded into the loss kernels (exact 2x)."""

    @staticmethod
    def forward(ctx, src, target, thres, min_kept, loss_weight, ignore_label):
        out, work = T.ohem_ce_up_fwd(src, target, thres, min_kept, loss_weight, ignore_label)
        ctx.save_for_backward(src, target, work, out)
        ctx.cfg = (loss_weight, ignore_label)
        ctx.mark_non_differentiable(out)
        ctx.set_materialize_grads(False)
        loss = out[0].clone()
        return loss, out

    @staticmethod
    def backward(ctx, dloss, _dout):
        src, target, work, out = ctx.saved_tensors
        return T.ohem_ce_up_bwd(src, target, work, out, dloss, ctx.cfg[0], ctx.cfg[1]), None, None, None, None, None


class OhemUp2Fn(Function):
    """Both OhemCrossEntropy losses of LEDHead.loss_by_feat on resize(src_k -> label size) in one launch set
    (ledn_ohem2_up_fwd / _bwd): shared label reads (uint8 copy), no per-pixel loss array."""

    @staticmethod
    def forward(ctx, src0, src1, target, cfg0, cfg1, ignore_label):
        out, work = T.ohem2_up_fwd(src0, src1, target, cfg0, cfg1, ignore_label)
        ctx.save_for_backward(src0, src1, work, out)
        ctx.cfg = (cfg0[2], cfg1[2], ignore_label, (int(target.shape[1]), int(target.shape[2])))
        ctx.mark_non_differentiable(out)
        ctx.set_materialize_grads(False)
        return out[0, 0].clone(), out[1, 0].clone(), out

    @staticmethod
    def backward(ctx, dl0, dl1, _dout):
        src0, src1, work, out = ctx.saved_tensors
        lw0, lw1, ign, hw = ctx.cfg
        zero = None
        if dl0 is None or dl1 is None:          # (one of the losses not part of the objective: its gradient is zero)
            zero = torch.zeros(1, dtype=torch.float32, device=src0.device)
        d0, d1 = T.ohem2_up_bwd(src0, src1, hw, work, out, dl0 if dl0 is not None else zero,
                                dl1 if dl1 is not None else zero, lw0, lw1, ign)
        return d0, d1, None, None, None, None


FUSE_LOSS_RESIZE = _knob_int('LEDN_FUSE_LOSS_RESIZE', 1)
FUSE_LOSS_PAIR = _knob_int('LEDN_FUSE_LOSS_PAIR', 1)     # both losses in one launch set (ohem_fused.hip)


def led_head_loss_by_feat(h, seg_logits, batch_data_samples):
    """LEDHead.loss_by_feat (led_head.py:101-146) on the four NCHW(-view) training logits."""
    from .lednet import to_nhwc
    xc, xs, h1, h2 = (to_nhwc(t, torch.float32) for t in seg_logits)
    labels = [ds.gt_sem_seg.data for ds in batch_data_samples]
    base = labels[0]._base if labels[0]._base is not None else None
    if (base is not None and base.dim() == 4 and base.shape[0] == len(labels) and base.is_contiguous()
            and all(lb._base is base and lb.data_ptr() == base[i].data_ptr() for i, lb in enumerate(labels))):
        label = base                      # the samples are views of ONE resident N x 1 x H x W batch: no copy
    else:
        label = torch.stack(labels, dim=0)
    hw = label.shape[2:]
    y = label.squeeze(1).contiguous()
    c0, c1 = h.loss_decode[0], h.loss_decode[1]
    H, W = hw
    if FUSE_LOSS_RESIZE and H % 2 == 0 and W % 2 == 0 and xc.shape[-1] == 2:
        # the last (exact 2x) resize of each fused output runs inside the loss kernels: the full-resolution logits
        # and their gradient are never written
        ctx2, spa2 = fuse_loss_half(xc, h1, h2, hw), fuse_loss_half(xs, h1, h2, hw)
        if (FUSE_LOSS_PAIR and W % 4 == 0 and c0.ignore_label == c1.ignore_label and 0 <= c0.ignore_label <= 255):
            l0, l1, out = OhemUp2Fn.apply(ctx2, spa2, y, (c0.thresh, c0.min_kept, c0.loss_weight),
                                          (c1.thresh, c1.min_kept, c1.loss_weight), c0.ignore_label)
            return {'loss_context': l0, 'loss_spatial': l1, 'acc_seg': out[0, 1:2]}
        l0, out0 = OhemUpFn.apply(ctx2, y, c0.thresh, c0.min_kept, c0.loss_weight, c0.ignore_label)
        l1, _ = OhemUpFn.apply(spa2, y, c1.thresh, c1.min_kept, c1.loss_weight, c1.ignore_label)
        return {'loss_context': l0, 'loss_spatial': l1, 'acc_seg': out0[1:2]}
    ctx = fuse_loss(xc, h1, h2, hw)
    spa = fuse_loss(xs, h1, h2, hw)
    l0, out0 = OhemFn.apply(ctx, y, c0.thresh, c0.min_kept, c0.loss_weight, c0.ignore_label)
    l1, _ = OhemFn.apply(spa, y, c1.thresh, c1.min_kept, c1.loss_weight, c1.ignore_label)
    return {'loss_context': l0, 'loss_spatial': l1, 'acc_seg': out0[1:2]}


def led_head_loss(h, inputs, batch_data_samples):
    """LEDHead.loss: forward (training) + loss_by_feat (NHWC tensors passed straight through)"""
    from .lednet import to_nchw_view
    return led_head_loss_by_feat(h, tuple(to_nchw_view(t) for t in led_head_forward_train(h, inputs)),
                                 batch_data_samples)


# --------------------------------------------------------------------------- #
# Trainer: SGD(momentum, wd) + PolyLR + data-parallel gradient all-reduce
# --------------------------------------------------------------------------- #
class Trainer:
    """One process per GPU.  Gradients live in ONE flat f32 buffer (parameter
    ``.grad`` tensors are views of it): a single multi-tensor SGD launch updates
    the model and re-zeroes the buffer, and the data-parallel exchange is a few
    large RCCL all-reduces over contiguous slices instead of one per tensor."""

    def __init__(self, model, cfg=None, world_size=1, lr=None, momentum=None, weight_decay=None,
                 max_iters=None, power=0.9, eta_min=0.0, bucket_mb=2.0, direct_grads=True, collectives=None):
        opt = dict((cfg or {}).get('optimizer', {}))
        self.model = model
        self.base_lr = lr if lr is not None else opt.get('lr', 0.01)
        self.momentum = momentum if momentum is not None else opt.get('momentum', 0.9)
        self.wd = weight_decay if weight_decay is not None else opt.get('weight_decay', 5e-4)
        sched = ((cfg or {}).get('param_scheduler') or [dict(power=0.9, eta_min=0, end=80000)])[0]
        self.max_iters = max_iters or sched.get('end', 80000)
        self.power, self.eta_min = sched.get('power', power), sched.get('eta_min', eta_min)
        self.iter = 0
        self.world = world_size
        # gradient layout: [stem | everything else], each in module order.  The stem's backward comes LAST (and is
        # the long high-resolution part): when it starts, every other parameter gradient is complete
        # (GradReadyFn 'post_stem') and that tail of the buffer is all-reduced while the stem's backward runs.
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        late = [p for n, p in named if n.startswith('backbone.stem.')]
        early = [p for n, p in named if not n.startswith('backbone.stem.')]
        self.params = late + early
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.n_late = sum(p.numel() for p in late)
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_mom = torch.zeros(n, dtype=torch.float32, device=dev)
        self.views, self.moms, off = [], [], 0
        for p in self.params:
            k = p.numel()
            self.views.append(self.flat_grad[off:off + k].view_as(p))
            self.moms.append(self.flat_mom[off:off + k].view_as(p))
            off += k
        self.table = None
        self.direct_grads = direct_grads     # backward kernels reduce into the flat gradient buffer
        self._sink_map = {}
        self._arena = ops.ZeroArena(dev)
        self._lr_dev = None
        self._graph = None
        self.bucket_elems = int(bucket_mb * (1 << 20) / 4)
        sync = getattr(model.backbone, 'sync_bn', False) or getattr(model.decode_head, 'sync_bn', False)
        _Env.world = world_size
        # collectives: 'rccl' = ncclAllReduce on the launch streams (rccl.CommSet: capturable in the step's
        # hipGraph), 'torch' = torch.distributed (eager only), default 'auto' = rccl on the GPU when it
        # initialises, else torch.  'rccl' with world_size 1 is the single-GPU self-test of that path.
        mode = collectives or _os.environ.get('LEDN_COLLECTIVES', 'auto')
        # N > 1 default = ONE launch stream: every collective of the step (SyncBN statistics, then the gradient
        # exchange) is issued from a single stream in program order, so the order is identical on all ranks under
        # eager launches AND under hipGraph replay, whatever the runtime does with independent graph branches.  The
        # multi-communicator form (one communicator per branch stream, gradient exchange overlapped with the stem's
        # backward; validated on ONE rank only) deadlocks if two ranks ever start two communicators' kernels from a
        # shared hardware queue in opposite orders -- opt in with LEDN_MULTI_COMM=1 once an 8-GPU run has shown it.
        self.multi_comm = bool(_knob_int('LEDN_MULTI_COMM', 0))     # (experimental: needs LEDN_EXPERIMENTAL=1)
        self._single_stream = (world_size > 1 or mode == 'rccl') and not self.multi_comm   # ('rccl' at N = 1: the self-test
        # of the N > 1 path on one GPU runs what N > 1 would run)
        self.comm = None
        self.dist = None
        if world_size > 1:
            import torch.distributed as dist
            self.dist = dist
            for p in self.params:                       # DDP init: rank-0 weights everywhere
                dist.broadcast(p.data, 0)
            for b in model.buffers():
                if b.is_floating_point():
                    dist.broadcast(b.data, 0)
        if mode in ('auto', 'rccl') and dev.type == 'cuda' and (world_size > 1 or mode == 'rccl'):
            try:
                from . import rccl
                slots = [0] if self._single_stream else sorted(
                    {0} | ({1} if CTX_FORKS else set()) | ({SEAM_SLOT} if SEAM_SLOT else set())
                    | (set(range(3, 7)) if MFAF_FORK else set()))
                self.comm = rccl.CommSet(self.dist.get_rank() if self.dist is not None else 0, world_size, dev,
                                         slots=tuple(slots) + ('grad',))
            except Exception as e:   # noqa: BLE001 -- 'auto': fall back to torch.distributed
                if mode == 'rccl':
                    raise
                import sys
                print(f'[led_net_amd] direct RCCL unavailable ({e!r}); collectives through torch.distributed',
                      file=sys.stderr)
        if self.comm is not None:
            self.coll = _Collective(commset=self.comm)
        elif self.dist is not None:
            self.coll = _Collective(dist=self.dist)
        else:
            self.coll = None
        self._sync_bn = self.coll if sync else None
        self._gstream = (torch.cuda.Stream(device=dev)
                         if (self.coll is not None and dev.type == 'cuda' and not self._single_stream) else None)
        self._ready = {}
        self._early_done = False
        self.overlap_exchange = bool(_knob_int('LEDN_OVERLAP_EXCHANGE', 1)) and not self._single_stream

    @property
    def _all_reduce(self):      # (bench.py / older callers: "does this trainer exchange gradients")
        return self.coll

    # ------------------------------------------------------------------ #
    # resume: what mmengine's CheckpointHook stores besides the weights (`optimizer` = OptimWrapper.state_dict() =
    # torch.optim.SGD.state_dict() over model.parameters() order, `param_schedulers` = [PolyLR.state_dict()])
    def optimizer_state_dict(self):
        """torch.optim.SGD-format state: parameter index = position in ``model.parameters()``; a momentum
        buffer only for parameters that have received a gradient (SGD creates its state lazily)."""
        order = list(self.model.parameters())
        pos = {id(p): i for i, p in enumerate(self.params)}
        live = {id(p) for p in getattr(self, 'live', [])} if self.table is not None else set()
        state = {}
        for i, p in enumerate(order):
            if id(p) in live:
                state[i] = {'momentum_buffer': self.moms[pos[id(p)]].detach().clone().cpu()}
        group = dict(lr=self.lr(), momentum=self.momentum, dampening=0, weight_decay=self.wd, nesterov=False,
                     maximize=False, foreach=None, differentiable=False, fused=None, initial_lr=self.base_lr,
                     params=list(range(len(order))))
        return {'state': state, 'param_groups': [group]}

    def scheduler_state_dict(self):
        return {'last_step': self.iter, 'begin': 0, 'end': self.max_iters, 'power': self.power, 'eta_min': self.eta_min,
                'total_iters': self.max_iters, 'base_values': [self.base_lr], 'by_epoch': False}

    def load_optimizer_state_dict(self, opt_sd, schedulers=None, iter=None):
        """restore the momentum buffers (and the schedule position) saved by optimizer_state_dict /
        an mmengine checkpoint of the same model"""
        order = list(self.model.parameters())
        pos = {id(p): i for i, p in enumerate(self.params)}
        for i, st in opt_sd.get('state', {}).items():
            p = order[int(i)]
            buf = st.get('momentum_buffer')
            if buf is not None and id(p) in pos:
                self.moms[pos[id(p)]].copy_(buf.to(self.flat_mom.device, torch.float32).view_as(p))
        if schedulers:
            self.iter = int(schedulers[0].get('last_step', self.iter))
        if iter is not None:
            self.iter = int(iter)

    def lr(self):
        """mmengine PolyLR (by iteration): (base-eta_min)*(1-it/max)^power + eta_min."""
        t = min(self.iter, self.max_iters) / self.max_iters
        return (self.base_lr - self.eta_min) * (1.0 - t) ** self.power + self.eta_min

    def _attach_grads(self):
        for p, v in zip(self.live, self.live_views):
            p.grad = v

    def _enter(self):
        self._saved_multi_stream = ops.MULTI_STREAM
        if self._single_stream:
            ops.MULTI_STREAM = False         # no branch streams: one ordered sequence of collectives per rank
        ops.PendingRows.entry = None
        self._arena.reset()                  # one fill for every small zeroed scratch of the step
        ops.set_zero_arena(self._arena)
        _Sinks.map = self._sink_map
        ops.WgradDefer.active = bool(WGRAD_DEFER and self._sink_map)
        ops.WgradDefer.pending = []
        _Env.world = self.world
        _Env.sync_bn = self._sync_bn
        self._ready, self._early_done = {}, False
        # the early exchange needs the gradients to land in the flat buffer DURING the backward (sinks)
        _Env.grad_ready = self._on_grad_ready if (self.coll is not None and self.overlap_exchange
                                                  and self._sink_map) else None

    def _leave(self):
        ops.MULTI_STREAM = self._saved_multi_stream
        ops.set_zero_arena(None)
        ops.WgradDefer.active = False
        _Sinks.map = {}
        _Env.sync_bn = None
        _Env.grad_ready = None

    def train_step(self, inputs, data_samples):
        """forward + loss + backward + gradient all-reduce + SGD; returns the loss dict
        (device scalars; no host synchronisation)."""
        self.model.train()
        first = self.table is None
        self._enter()
        try:
            return self._train_step(inputs, data_samples, first)
        finally:
            self._leave()

    def forward_backward(self, inputs, data_samples):
        """forward + loss + backward only (gradients left in ``flat_grad``; no exchange, no SGD):
        the first half of train_step, for tests and gradient inspection.  Needs one completed
        train_step (the flat gradient views are attached there).  The gradient buffer is cleared first: the
        BatchNorm backward reduces its sums INTO the gradient views and reads them back (ops_train.bn_act_bwd),
        which is only right from zero (train_step gets that from the SGD kernel's re-zeroing)."""
        assert self.table is not None, 'run one train_step first'
        self.model.train()
        self.flat_grad.zero_()
        self._enter()
        _Env.grad_ready = None
        try:
            return self._forward_backward(inputs, data_samples, False)
        finally:
            self._leave()

    def _join_side_streams(self):
        """the launch stream waits for the auxiliary streams that have no consumer inside the step
        (weight gradients)"""
        dev = self.flat_grad.device
        if dev.type == 'cuda':
            for slot in {WGRAD_SLOT, T.DW_WGRAD_SLOT} - {0}:
                st = ops._AUX.get((dev, slot))
                if st is not None:
                    torch.cuda.current_stream(dev).wait_stream(st)

    def _exchange(self, lo, hi):
        """all-reduce flat_grad[lo:hi] in buckets on the gradient-exchange stream, after everything queued so
        far on the launch streams"""
        if hi <= lo:
            return
        dev = self.flat_grad.device
        if self._gstream is not None:
            self._gstream.wait_stream(torch.cuda.current_stream(dev))
            for (d, _slot), st in ops._AUX.items():
                if d == dev:
                    self._gstream.wait_stream(st)
            ctx = torch.cuda.stream(self._gstream)
        else:
            import contextlib
            ctx = contextlib.nullcontext()
        with ctx:
            for off in range(lo, hi, self.bucket_elems):
                t = self.flat_grad[off:min(hi, off + self.bucket_elems)]
                self.coll.all_reduce(t, t, slot='grad')

    def _on_grad_ready(self, tag):
        """GradReadyFn callback (host side, inside the backward): 'post_stem' fires three times (the stem's
        output and the two stem tensors the heads read); the third starts the exchange of every
        non-stem gradient."""
        self._ready[tag] = self._ready.get(tag, 0) + 1
        if tag == 'post_stem' and self._ready[tag] == 3 and not self._early_done:
            self._early_done = True
            self._join_side_streams()   # (LEDN_WGRAD_SLOT / LEDN_DW_WGRAD_SLOT: weight gradients written on auxiliary streams)
            _DwBanks.flush()            # (every depthwise filter lives outside the stem: their gradients are complete)
            ops.WgradDefer.finish()     # (and the non-stem convolution weight gradients: summed into the buffer now)
            self._exchange(self.n_late, self.flat_grad.numel())

    def _forward_backward(self, inputs, data_samples, first):
        if first:
            _Packs.reset()
            _DwBanks.reset()
            ops.WgradDefer.reset()
            for p in self.params:
                p.grad = None
        else:
            self._attach_grads()
            if _Packs.table is not None:
                _Packs.table.run()          # all bf16 weight packs in one launch
            if _DwBanks.table is not None:
                _DwBanks.table.run(0)       # all depthwise filter banks in one launch
        losses = self.model(inputs, data_samples, mode='loss')
        total = None
        for k, v in losses.items():
            if 'loss' in k:
                total = v if total is None else total + v
        total.backward()
        self._join_side_streams()           # (the flush / summing launches read what auxiliary streams may have written)
        _DwBanks.flush()                    # all depthwise bank gradients -> the filters' gradient views, one launch
        ops.WgradDefer.finish()             # all convolution weight gradients: partial tiles -> gradient views, one launch
        if first:
            # Parameters that never receive a gradient (SEAM conv_1: the binarised edge
            # map is non-differentiable) are skipped, as torch.optim.SGD skips grad=None.
            idx = [i for i, p in enumerate(self.params) if p.grad is not None]
            self.live = [self.params[i] for i in idx]
            self.live_views = [self.views[i] for i in idx]
            for p, v in zip(self.live, self.live_views):
                v.copy_(p.grad)
            self._attach_grads()
            self.table = T.SgdTable(self.live, self.live_views, [self.moms[i] for i in idx])
            if self.direct_grads:
                self._sink_map = {id(p): v for p, v in zip(self.live, self.live_views)}
            if _Packs.entries:
                _Packs.table = T.PackTable([(w, buf, k[1], k[2]) for k, (w, buf) in _Packs.entries.items()])
                _Packs.frozen = True
            if self.direct_grads and _DwBanks.entries:
                banks = []
                for e in _DwBanks.entries.values():
                    sinks = [self._sink_map.get(id(w)) for w in e['weights']]
                    if any(sk is None for sk in sinks):
                        continue
                    dev0 = e['weights'][0].device
                    e['packed'] = torch.empty(e['shape'], dtype=torch.float32, device=dev0)
                    e['dpacked'] = torch.zeros(e['shape'], dtype=torch.float32, device=dev0)
                    _DwBanks.by_ptr[e['packed'].data_ptr()] = e['dpacked']
                    banks.append((list(e['weights']), sinks, e['stacked'], e['packed'], e['dpacked']))
                _DwBanks.entries = {k: e for k, e in _DwBanks.entries.items() if 'packed' in e}
                if banks:
                    _DwBanks.table = T.DwBankTable(banks)
                    _DwBanks.frozen = True
        return losses

    def _train_step(self, inputs, data_samples, first):
        losses = self._forward_backward(inputs, data_samples, first)
        if self.coll is not None:
            # what the backward did not already start: the stem's gradients (or everything)
            self._exchange(0, self.n_late if self._early_done else self.flat_grad.numel())
            if self._gstream is not None:
                torch.cuda.current_stream(self.flat_grad.device).wait_stream(self._gstream)
        if self._lr_dev is not None and not torch.cuda.is_current_stream_capturing():
            self._lr_dev.fill_(self.lr())       # eager step after a capture(): keep the device-resident rate current
        self.table.step(self.lr(), self.momentum, self.wd, 1.0 / self.world, lr_dev=self._lr_dev)
        self.iter += 1
        # detached: a caller holding the returned losses would otherwise keep the step's autograd graph
        # (and its AccumulateGrad nodes, bound to this step's stream) alive into a later hipGraph capture,
        # where they run on the wrong stream and break the capture
        return {k: v.detach() for k, v in losses.items()}

    # ------------------------------------------------------------------ #
    # hipGraph replay of the whole step (forward + loss + backward + SGD): ~800 launches
    # per step are submitted as ONE graph, the host only refreshes the input buffers and the
    # device-resident learning rate.  With N > 1 ranks the collectives must be stream operations
    # (rccl.Comm): torch.distributed's cannot be captured.
    def capture(self, inputs, data_samples, warmup=3, restore=False):
        """Capture one whole step on (inputs, data_samples).  The `warmup` eager steps before the capture are
        REAL optimizer steps on that batch (allocator / workspace / sink warm-up needs them); self.iter counts
        them.  restore=True makes the capture free of side effects instead: parameters, momentum buffers, module
        buffers (BatchNorm running statistics) and the iteration counter are put back after the warm-up, so the
        first replay() is step `self.iter` of an uninterrupted schedule (tools/train.py: --resume continues exactly).
        Afterwards: replay(), or train_step() (eager) -- both keep the device-resident learning rate current."""
        assert self.dist is None or self.comm is not None, 'graph capture with N > 1 needs the direct RCCL communicator'
        dev = inputs.device
        self._static_in = inputs.clone()
        # ONE resident N x 1 x H x W label batch; the samples are views of it (led_head_loss_by_feat then needs
        # no per-step torch.stack: 134 MB of int64 at 16 x 1024 x 1024)
        self._static_lab = torch.stack([ds.gt_sem_seg.data for ds in data_samples], dim=0)
        from .segmentor import SegDataSample
        self._static_samples = [SegDataSample(gt=self._static_lab[i], metainfo=dict(getattr(ds, 'metainfo', {}) or {}))
                                for i, ds in enumerate(data_samples)]
        self._lr_dev = torch.zeros(1, dtype=torch.float32, device=dev)
        self._valid_host = None         # (a second capture() starts from the new batch's padding extents)
        if inputs.dtype == torch.uint8 and getattr(self.model, 'pre_scale', None) is not None:
            # per-image valid extents (batch padding) live in a resident buffer the graph reads: replay() refreshes it
            self.model._valid_static = self.model.valid_extents(inputs.shape[2:], data_samples, inputs.shape[0]).to(dev)
            self.model._valid_for_ptr = self._static_in.data_ptr()
        snap = None
        if restore:
            snap = ([p.detach().clone() for p in self.params], self.flat_mom.clone(),
                    [b.detach().clone() for b in self.model.buffers()], self.iter)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(warmup, 2 if self.table is None else 0)):     # (the sinks attach in the second step)
                self._lr_dev.fill_(self.lr())
                self.train_step(self._static_in, self._static_samples)
            if snap is not None:
                with torch.no_grad():
                    for p, v in zip(self.params, snap[0]):
                        p.copy_(v)
                    self.flat_mom.copy_(snap[1])
                    for b, v in zip(self.model.buffers(), snap[2]):
                        b.copy_(v)
                self.iter = snap[3]
        torch.cuda.current_stream(dev).wait_stream(side)
        self._lr_dev.fill_(self.lr())
        self._graph = torch.cuda.CUDAGraph()
        # N > 1: other threads of the process (ProcessGroupNCCL's watchdog) keep calling the runtime while this
        # thread captures; 'thread_local' confines the capture-safety checks to the capturing thread
        mode = _os.environ.get('LEDN_CAPTURE_MODE') or ('thread_local' if self.dist is not None else 'global')
        with torch.cuda.graph(self._graph, capture_error_mode=mode):
            self._static_out = self.train_step(self._static_in, self._static_samples)
        self.iter -= 1          # capturing records the step, it does not execute it (PolyLR stays in step)
        return self

    def replay(self, inputs=None, data_samples=None):
        """one captured step; new inputs (same shapes as the captured batch) are copied into the static buffers
        first: pixels, labels and the per-image padding extents.  Stream-ordered: the copies and the graph launch go
        to the current stream, so a batch produced on another stream needs the usual event / wait_stream first."""
        if inputs is not None and inputs is not self._static_in:
            if inputs.shape != self._static_in.shape or inputs.dtype != self._static_in.dtype:
                raise ValueError(f'replay: batch {tuple(inputs.shape)} {inputs.dtype} != captured '
                                 f'{tuple(self._static_in.shape)} {self._static_in.dtype}')
            self._static_in.copy_(inputs, non_blocking=True)
            labs = [ds.gt_sem_seg.data for ds in data_samples]
            base = labs[0]._base
            if (base is not None and base.shape == self._static_lab.shape and base.is_contiguous()
                    and all(lb._base is base and lb.data_ptr() == base[i].data_ptr() for i, lb in enumerate(labs))):
                self._static_lab.copy_(base, non_blocking=True)       # views of one N x 1 x H x W batch: one copy
            else:
                for i, lb in enumerate(labs):
                    self._static_lab[i].copy_(lb, non_blocking=True)
            if getattr(self.model, '_valid_static', None) is not None:
                ext = self.model.valid_extents(inputs.shape[2:], data_samples, inputs.shape[0])
                if getattr(self, '_valid_host', None) is None or not torch.equal(ext, self._valid_host):
                    self._valid_host = ext
                    self.model._valid_static.copy_(ext.to(self.model._valid_static.device))
        self._lr_dev.fill_(self.lr())
        self._graph.replay()
        self.iter += 1
        return self._static_out


def smoke_train_step(model, dev):
    """one tiny train step on the device, loss compared with the CPU oracle by the caller's tests"""
    import led_net_amd as L
    model.train()
    g = torch.Generator().manual_seed(7)
    img = torch.randint(0, 256, (2, 3, 320, 320), dtype=torch.uint8, generator=g).to(dev)
    lab = torch.randint(0, 2, (2, 1, 320, 320), dtype=torch.int64, generator=g)
    lab[:, :, :8] = 255
    samples = [L.SegDataSample(gt=lab[i].to(dev)) for i in range(2)]
    tr = Trainer(model, max_iters=100)
    out = tr.train_step(img, samples)
    vals = {k: float(v.float().reshape(-1)[0]) for k, v in out.items()}
    print('smoke train step:', vals)
    assert all(math.isfinite(v) for v in vals.values())
