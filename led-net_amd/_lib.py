"""ctypes binding of the C ABI in include/ledn.h.

The product path loads exactly one library: ``csrc/libledn_hip.so`` (gfx950).
If it is missing, or a tensor handed to an op is not on a HIP device, the op
raises -- there is no CPU fallback.  (The test-suite may bind the CPU emulation
build of the same kernel sources through :func:`use_library`; nothing in the
product does.)
"""
import contextlib
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
from ._env import knob as _knob  # noqa: E402
HIP_LIB_PATH = _knob('LEDN_HIP_LIB', None) or os.path.join(_HERE, 'csrc', 'libledn_hip.so')   # (override: A/B of library builds)

OK, EINVAL, ELAUNCH = 0, 1, 2
F32, BF16, U8 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_RELU6, ACT_PRELU, ACT_SIGMOID = 0, 1, 2, 3, 4
RES_NONE, RES_ADD, RES_GATE = 0, 1, 2
OPT_CONV_WORKGROUPS, OPT_WGRAD_WORKGROUPS, OPT_STREAM_FAST, OPT_DETERMINISTIC, OPT_BN_FUSED = 0, 1, 2, 3, 4
ESKIP = 3

vp, fp, i32, i64 = C.c_void_p, C.c_void_p, C.c_int, C.c_longlong


class ConvDesc(C.Structure):
    _fields_ = [('x', vp), ('xadd', vp), ('w', fp), ('w_bf16', vp), ('y', vp), ('res', vp),
                ('in_scale', fp), ('in_shift', fp), ('out_scale', fp), ('out_shift', fp),
                ('slope', fp), ('stat_sum', fp), ('stat_sqsum', fp),
                ('ws_co', i64), ('ws_ci', i64), ('ws_tap', i64),
                ('N', i32), ('H', i32), ('W', i32), ('Cin', i32), ('Ho', i32), ('Wo', i32), ('Cout', i32),
                ('KH', i32), ('KW', i32), ('stride', i32), ('pad', i32), ('dil', i32), ('groups', i32),
                ('in_act', i32), ('act_out', i32), ('res_mode', i32),
                ('dtype_x', i32), ('dtype_y', i32), ('transposed', i32), ('in_slope', fp)]


class WgradDesc(C.Structure):
    _fields_ = [('x', vp), ('xadd', vp), ('dz', vp), ('dw', fp), ('db', fp),
                ('in_scale', fp), ('in_shift', fp),
                ('ws_co', i64), ('ws_ci', i64), ('ws_tap', i64),
                ('N', i32), ('H', i32), ('W', i32), ('Cin', i32), ('Ho', i32), ('Wo', i32), ('Cout', i32),
                ('KH', i32), ('KW', i32), ('stride', i32), ('pad', i32), ('dil', i32), ('groups', i32),
                ('in_act', i32), ('dtype_x', i32), ('dtype_dz', i32), ('in_slope', fp)]


class WgradFinishEntry(C.Structure):
    _fields_ = [('part', fp), ('dw', fp), ('ws_co', i64), ('ws_ci', i64), ('ws_tap', i64),
                ('nbx', i32), ('pairs', i32), ('KK', i32), ('ci_tiles', i32), ('Cin', i32), ('Cout', i32),
                ('groups', i32), ('chunk0', i32)]


class DwDesc(C.Structure):
    _fields_ = [('x', vp), ('w', fp), ('y', vp), ('out_scale', fp), ('out_shift', fp), ('slope', fp),
                ('stat_sum', fp), ('stat_sqsum', fp),
                ('N', i32), ('H', i32), ('W', i32), ('C', i32), ('Ho', i32), ('Wo', i32),
                ('KH', i32), ('KW', i32), ('stride', i32), ('pad', i32),
                ('dil', i32 * 4), ('group_size', i32), ('act_out', i32), ('ext1', i32),
                ('dtype_x', i32), ('dtype_y', i32)]


class DwPackDesc(C.Structure):
    _fields_ = [('w', fp * 8), ('dw', fp * 8), ('n', i32 * 8), ('nsrc', i32), ('taps', i32), ('stacked', i32)]


class DwPackEntry(C.Structure):
    _fields_ = [('d', DwPackDesc), ('packed', fp), ('dpacked', fp)]


class MfafCtxDesc(C.Structure):          # mirrors ledn_mfafctx_desc
    _fields_ = [(n, vp * 4) for n in ('pooled', 'z1', 'z2', 'w1', 'b1', 'gamma', 'beta', 'running_mean', 'running_var',
                                      'w2', 'b2', 'bn1')] + \
               [('stats1', vp), ('P', i32 * 4), ('C', i32), ('Ci', i32), ('momentum', C.c_float), ('eps', C.c_float)] + \
               [(n, vp * 4) for n in ('gamma2', 'beta2', 'running_mean2', 'running_var2', 'bn2')] + [('stats2', vp)] + \
               [('phase', i32), ('count_scale', C.c_float)]


class MfafCtxBwdDesc(C.Structure):       # mirrors ledn_mfafctx_bwd_desc
    _fields_ = [(n, vp * 4) for n in ('pooled', 'z1', 'dz2', 'w1', 'w2', 'bn1', 'g', 'dpooled', 'dw1', 'db1', 'dgamma',
                                      'dbeta', 'dw2', 'db2')] + \
               [('sums', vp), ('P', i32 * 4), ('C', i32), ('Ci', i32)] + \
               [(n, vp * 4) for n in ('z2', 'bn2', 'dz2s', 'dgamma2', 'dbeta2')] + [('sums2', vp)] + \
               [('phase', i32), ('count_scale', C.c_float), ('sums_local', vp), ('sums2_local', vp)]


class AugEntry(C.Structure):          # mirrors ledn_aug_entry field for field
    _fields_ = [('img', vp), ('seg', vp), ('H', i32), ('W', i32), ('RH', i32), ('RW', i32),
                ('sx', C.c_double), ('sy', C.c_double), ('oy', i32), ('ox', i32), ('ch', i32), ('cw', i32),
                ('flip', i32), ('bright_on', i32), ('bright_beta', C.c_float), ('contrast_mode', i32),
                ('contrast_on', i32), ('contrast_alpha', C.c_float), ('sat_on', i32), ('sat_alpha', C.c_float),
                ('hue_on', i32), ('hue_delta', i32)]


class PyrDesc(C.Structure):
    _fields_ = [('x', vp), ('w', fp), ('y', vp),
                ('N', i32), ('H', i32), ('W', i32), ('n', i32), ('Ho', i32), ('Wo', i32), ('stride', i32),
                ('dil', i32 * 4), ('dtype_x', i32), ('dtype_y', i32)]


class AffineDesc(C.Structure):
    _fields_ = [('x', vp), ('xadd', vp), ('y', vp), ('res', vp), ('scale', fp), ('shift', fp), ('slope', fp),
                ('P', i64), ('C', i32), ('act', i32), ('res_mode', i32), ('dtype_x', i32), ('dtype_y', i32),
                ('stat_sum', fp), ('stat_sqsum', fp)]


class ResizeDesc(C.Structure):
    _fields_ = [('x', vp), ('add', vp), ('y', vp), ('argmax', vp),
                ('N', i32), ('H', i32), ('W', i32), ('C', i32), ('Ho', i32), ('Wo', i32),
                ('out_nchw', i32), ('dtype_x', i32), ('dtype_y', i32)]


class MfafDesc(C.Structure):
    _fields_ = [('x', vp), ('r', vp), ('xl', vp), ('ctx', fp * 4), ('ctx_size', i32 * 4),
                ('scale', fp * 5), ('shift', fp * 5), ('out', vp),
                ('N', i32), ('H', i32), ('W', i32), ('C', i32), ('dtype', i32), ('act', i32)]


class BnBwdDesc(C.Structure):
    _fields_ = [('z', vp), ('res', vp), ('dy', vp), ('scale', fp), ('shift', fp), ('slope', fp),
                ('mean', fp), ('invstd', fp), ('sum_g', fp), ('sum_gx', fp), ('dslope', fp),
                ('dz', vp), ('dres', vp), ('count', C.c_double), ('P', i64),
                ('C', i32), ('act', i32), ('res_mode', i32), ('bn_mode', i32),
                ('dtype_z', i32), ('dtype_y', i32), ('dz_add', vp), ('dres_add', vp), ('rows', fp)]


BNBWD_ROWS = 32     # include/ledn.h LEDN_BNBWD_ROWS


class HeadBwdDesc(C.Structure):
    _fields_ = [('bn', BnBwdDesc), ('head_dz', vp), ('w', fp), ('dw', fp), ('db', fp),
                ('N', i32), ('H', i32), ('W', i32), ('Co', i32), ('dtype_dz', i32)]


class DwBwdDesc(C.Structure):
    _fields_ = [('x', vp), ('dz', vp), ('w', fp), ('add', vp), ('dx', vp), ('dw', fp),
                ('N', i32), ('H', i32), ('W', i32), ('C', i32), ('Ho', i32), ('Wo', i32),
                ('KH', i32), ('KW', i32), ('stride', i32), ('pad', i32),
                ('dil', i32 * 4), ('group_size', i32), ('ext1', i32), ('dtype', i32)]


class PyrBwdDesc(C.Structure):
    _fields_ = [('x', vp), ('dy', vp), ('w', fp), ('gsum', vp), ('dx', vp), ('dw', fp),
                ('N', i32), ('H', i32), ('W', i32), ('n', i32), ('Ho', i32), ('Wo', i32), ('stride', i32),
                ('dil', i32 * 4), ('dtype', i32)]


class MfafBwdDesc(C.Structure):
    _fields_ = [('x', vp), ('r', vp), ('xl', vp), ('ctx', fp * 4), ('ctx_size', i32 * 4),
                ('scale', fp * 5), ('shift', fp * 5), ('dout', vp), ('dx', vp), ('dr', vp), ('ds', vp),
                ('dctx', fp * 4), ('N', i32), ('H', i32), ('W', i32), ('C', i32), ('dtype', i32), ('act', i32)]


class PackEntry(C.Structure):
    _fields_ = [('w', fp), ('out', vp), ('Cout', i32), ('Cin', i32), ('KK', i32), ('mode', i32), ('groups', i32)]


class SgdEntry(C.Structure):
    _fields_ = [('p', fp), ('g', fp), ('m', fp), ('n', i64)]


_PROTOS = {
    'ledn_bn_act_bwd_reduce': ([C.POINTER(BnBwdDesc), vp], i32),
    'ledn_bn_act_bwd_apply': ([C.POINTER(BnBwdDesc), vp], i32),
    'ledn_head_bwd_supported': ([C.POINTER(HeadBwdDesc)], i32),
    'ledn_head_bwd_reduce': ([C.POINTER(HeadBwdDesc), vp], i32),
    'ledn_head_bwd_apply': ([C.POINTER(HeadBwdDesc), vp], i32),
    'ledn_bn_act_bwd_fused': ([C.POINTER(BnBwdDesc), vp], i32),
    'ledn_bn_act_bwd_fused_check': ([i32, vp], i32),
    'ledn_dwconv2d_bwd_data': ([C.POINTER(DwBwdDesc), vp], i32),
    'ledn_dwconv2d_bwd_weight': ([C.POINTER(DwBwdDesc), vp], i32),
    'ledn_sesp_pyramid_bwd_data': ([C.POINTER(PyrBwdDesc), vp], i32),
    'ledn_sesp_pyramid_bwd_weight': ([C.POINTER(PyrBwdDesc), vp], i32),
    'ledn_bilinear_bwd': ([vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp], i32),
    'ledn_avgpool3x3s2_bwd': ([vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp], i32),
    'ledn_window_attn_bwd': ([vp, fp, vp, fp, fp, i32, i32, i32, i32, i32, i32, i32, vp], i32),
    'ledn_getb_pool_bwd': ([vp, vp, i32, i32, i32, i32, i32, i32, vp], i32),
    'ledn_mfaf_gate_bwd': ([C.POINTER(MfafBwdDesc), vp], i32),
    'ledn_mfaf_bwd_combine': ([vp, vp, vp, C.POINTER(fp), C.POINTER(i32), i32, i32, i32, i32, i32, i32, vp], i32),
    'ledn_ohem_work_floats': ([i64], i64),
    'ledn_ohem_ce_fwd': ([fp, vp, i64, i32, C.c_float, i64, C.c_float, i32, fp, fp, vp], i32),
    'ledn_ohem_ce_bwd': ([fp, vp, i64, i32, i32, fp, fp, fp, C.c_float, fp, vp], i32),
    'ledn_ohem_ce_up_fwd': ([fp, i32, i32, i32, i32, i32, vp, C.c_float, i64, C.c_float, i32, fp, fp, vp], i32),
    'ledn_ohem_ce_up_bwd': ([fp, i32, i32, i32, i32, i32, vp, i32, fp, fp, fp, C.c_float, fp, vp], i32),
    'ledn_ohem2_up_fwd': ([fp, fp, i32, i32, i32, i32, i32, vp, C.c_float, i64, C.c_float, C.c_float, i64, C.c_float, i32,
                           fp, fp, vp], i32),
    'ledn_ohem2_up_bwd': ([fp, fp, i32, i32, i32, i32, i32, i32, fp, fp, fp, fp, C.c_float, C.c_float, fp, fp, vp], i32),
    'ledn_ohem2_work_floats': ([i64], i64),
    'ledn_sgd_step': ([vp, i32, i64, C.c_float, fp, C.c_float, C.c_float, C.c_float, vp], i32),
    'ledn_abi_version': ([], i32),
    'ledn_set_workspace': ([vp, i64], i32),
    'ledn_bind_workspace': ([vp, vp, i64], i32),
    'ledn_set_option': ([i32, i64], i32),
    'ledn_conv2d': ([C.POINTER(ConvDesc), vp], i32),
    'ledn_conv2d_uses_mfma': ([C.POINTER(ConvDesc)], i32),
    'ledn_stats_defer_begin': ([], i32),
    'ledn_stats_defer_end': ([C.POINTER(C.c_void_p), C.POINTER(i32)], i32),
    'ledn_conv2d_deferred_stats': ([C.POINTER(ConvDesc), C.POINTER(C.c_void_p), C.POINTER(i32), vp], i32),
    'ledn_bn_finalize_rows': ([fp, i32, C.c_double, fp, fp, fp, fp, C.c_float, C.c_float, fp, fp, fp, fp, fp, fp, i32, vp], i32),
    'ledn_conv2d_wgrad_uses_mfma': ([C.POINTER(WgradDesc)], i32),
    'ledn_conv2d_wgrad_partial_floats': ([C.POINTER(WgradDesc)], i64),
    'ledn_conv2d_wgrad_partial': ([C.POINTER(WgradDesc), fp, i64, C.POINTER(WgradFinishEntry), vp], i32),
    'ledn_conv2d_wgrad_finish_multi': ([vp, i32, i32, vp], i32),
    'ledn_pack_conv_weights_multi': ([vp, i32, i64, vp], i32),
    'ledn_im2col_stem': ([vp, vp, i32, i32, i32, i32, i32, i32, vp], i32),
    'ledn_im2col_stem_planar': ([vp, i32, vp, i32, i32, i32, i32, i32, i32, fp, fp, vp, vp, C.c_float, vp], i32),
    'ledn_stem_conv': ([vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, fp, fp, vp, vp, C.c_float, fp, fp, i32, fp, fp, vp], i32),
    'ledn_stem_conv_wgrad': ([vp, i32, vp, fp, i32, i32, i32, i32, i32, i32, i32, fp, fp, vp, vp, C.c_float, vp], i32),
    'ledn_stem_conv_wgrad_bn': ([vp, i32, C.POINTER(BnBwdDesc), fp, i32, i32, i32, i32, i32, i32, i32, fp, fp, vp, vp, C.c_float, vp], i32),
    'ledn_pack_conv_weights': ([fp, vp, i32, i32, i32, i32, i32, i32, vp], i32),
    'ledn_conv2d_wgrad': ([C.POINTER(WgradDesc), vp], i32),
    'ledn_dwconv2d': ([C.POINTER(DwDesc), vp], i32),
    'ledn_iou_hist': ([vp, vp, i64, i32, i32, fp, vp], i32),
    'ledn_dw_pack': ([C.POINTER(DwPackDesc), fp, vp], i32),
    'ledn_dw_repack_multi': ([vp, i32, i32, i32, vp], i32),
    'ledn_mfaf_ctx_fwd': ([C.POINTER(MfafCtxDesc), i32, vp], i32),
    'ledn_mfaf_ctx_bwd': ([C.POINTER(MfafCtxBwdDesc), vp], i32),
    'ledn_augment_batch': ([vp, i32, vp, vp, i32, i32, i32, i32, vp], i32),
    'ledn_aug_crop_hist': ([vp, i32, i32, vp, vp], i32),
    'ledn_dw_unpack_grad': ([C.POINTER(DwPackDesc), fp, vp], i32),
    'ledn_sesp_pyramid': ([C.POINTER(PyrDesc), vp], i32),
    'ledn_channel_stats': ([vp, vp, i64, i32, i32, fp, fp, vp], i32),
    'ledn_bn_finalize': ([fp, fp, C.c_double, fp, fp, fp, fp, C.c_float, C.c_float, fp, fp, fp, fp, i32, vp], i32),
    'ledn_affine_act': ([C.POINTER(AffineDesc), vp], i32),
    'ledn_nchw_to_nhwc': ([vp, i32, vp, i32, i32, i32, i32, i32, fp, fp, vp, vp, C.c_float, vp], i32),
    'ledn_bilinear': ([C.POINTER(ResizeDesc), vp], i32),
    'ledn_adaptive_avgpool': ([vp, vp, fp, i32, i32, i32, i32, i32, i32, vp], i32),
    'ledn_relpos_bias': ([fp, vp, fp, i32, i32, i32, vp], i32),
    'ledn_relpos_bias_bwd': ([fp, vp, fp, i32, i32, i32, vp], i32),
    'ledn_avgpool2d': ([vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp], i32),
    'ledn_avgpool2d_bwd': ([vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp], i32),
    'ledn_avgpool3x3s2': ([vp, vp, i32, i32, i32, i32, i32, i32, i32, vp], i32),
    'ledn_window_attn': ([vp, fp, vp, i32, i32, i32, i32, i32, i32, i32, vp], i32),
    'ledn_getb_pool': ([vp, vp, vp, i32, i32, i32, i32, i32, i32, vp], i32),
    'ledn_mfaf_gate': ([C.POINTER(MfafDesc), vp], i32),
    'ledn_seam_edge': ([fp, fp, fp, i32, i32, i32, i32, C.c_float, C.c_float, vp], i32),
}
EXPORTS = tuple(_PROTOS)


ABI_VERSION = 5      # include/ledn.h LEDN_ABI_VERSION: bumped with every struct / signature change


class LednError(RuntimeError):
    pass


class Library:
    """One loaded build of the kernel library."""

    def __init__(self, path, is_hip):
        if not os.path.exists(path):
            raise LednError(
                f'{path} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                f'(hipcc --offload-arch=gfx950).  There is no CPU fallback.')
        self.path = path
        self.is_hip = is_hip
        self.cdll = C.CDLL(path)
        for name, (argtypes, restype) in _PROTOS.items():
            fn = getattr(self.cdll, name)
            fn.argtypes = argtypes
            fn.restype = restype
        v = self.cdll.ledn_abi_version()
        if v != ABI_VERSION:
            raise LednError(f'{path}: ABI version {v}, expected {ABI_VERSION} (stale build? run __graft_entry__.build())')
        self._workspaces = {}
        self._bound = set()

    def ensure_workspace(self, device, slot=0, nfloats=32 << 20, stream=None):
        """the reusable scratch buffer (128 MiB) of two-stage reductions, BOUND TO THE LAUNCH STREAM
        (ledn_bind_workspace): the scratch is reused launch after launch in stream order, so every HIP stream that
        runs ledn kernels (ops.Fork's branch streams, a capture's side stream) has its own buffer and the library
        finds it by the stream argument of the call -- no process-wide "current workspace" to keep in step."""
        key = (device, slot, stream)
        if key not in self._bound:
            ws = self._workspaces.get((device, slot))
            if ws is None:
                import torch
                ws = self._workspaces[(device, slot)] = torch.empty(nfloats, dtype=torch.float32, device=device)
            self.call('ledn_bind_workspace', stream, ws.data_ptr(), ws.numel())
            self._bound.add(key)

    def set_option(self, option, value):
        """launch-shape knob (include/ledn.h LEDN_OPT_*); value <= 0 restores the default"""
        self.call('ledn_set_option', option, value)

    def call(self, name, *args):
        rc = getattr(self.cdll, name)(*args)
        if rc != OK:
            raise LednError(f'{name} failed: ' + {EINVAL: 'invalid argument (shape/dtype/pointer check)',
                                                   ELAUNCH: 'kernel launch error'}.get(rc, str(rc)))


_hip = None
_override = None
# deterministic mode (include/ledn.h LEDN_OPT_DETERMINISTIC): every cross-workgroup reduction in a fixed order, no f32
# atomics -- two runs of a step are bit-identical.  LEDN_DETERMINISTIC=1 or set_deterministic(True); the reference stack's
# switch for the same thing is randomness=dict(seed=..., deterministic=True) (mmengine Runner -> torch.use_deterministic_algorithms)
_DET = bool(int(os.environ.get('LEDN_DETERMINISTIC', '0')))


def set_deterministic(flag=True):
    """switch the library's deterministic mode on / off (for the library in use now and any loaded later)"""
    global _DET
    _DET = bool(flag)
    for lib in (_hip, _override):
        if lib is not None:
            lib.set_option(OPT_DETERMINISTIC, int(_DET))


def is_deterministic():
    return _DET


def get_lib():
    """The library every op dispatches to: libledn_hip.so, loaded on first use."""
    global _hip
    if _override is not None:
        return _override
    if _hip is None:
        _hip = Library(HIP_LIB_PATH, is_hip=True)
        for env, opt in (('LEDN_CONV_WGS', 0), ('LEDN_WGRAD_WGS', 1)):      # A/B measurements of the launch-shape knobs
            if _knob(env, None) is not None:                                 # (experimental: _env.py)
                _hip.set_option(opt, int(_knob(env, None)))
        if _knob('LEDN_STREAM_FAST', None) is not None:         # A/B measurements: bit 0 = BatchNorm / affine streaming
            _hip.set_option(OPT_STREAM_FAST, int(_knob('LEDN_STREAM_FAST', None)))   # kernels, bit 1 = LDS-tiled depthwise 3x3
        _hip.set_option(OPT_DETERMINISTIC, int(_DET))
    return _hip


@contextlib.contextmanager
def use_library(lib):
    """TEST HOOK ONLY: temporarily bind another build (the CPU emulation)."""
    global _override
    prev = _override
    _override = lib
    lib.set_option(OPT_DETERMINISTIC, int(_DET))
    try:
        yield lib
    finally:
        _override = prev
