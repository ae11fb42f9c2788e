"""csrc/stream_fast.hip (16-B-per-lane streaming kernels of the bf16 BatchNorm / activation passes) against
(a) the generic kernels behind the same C-ABI entry (LEDN_OPT_STREAM_FAST toggled) and (b) plain torch fp32
math on the same bf16 inputs.  Shapes include a ragged tail (vector count not a multiple of the 1024 vectors a
workgroup owns) and every residual / activation mode.  Emulator on CPU; the same tests on the MI355X (-m gpu)."""
import pytest
import torch

OPT_STREAM_FAST = 2
BF = torch.bfloat16


def _toggle(on):
    from led_net_amd import _lib
    _lib.get_lib().set_option(OPT_STREAM_FAST, 11 if on else 0)


def _both(fn):
    """fn() under the fast kernels and under the generic ones"""
    try:
        _toggle(True)
        a = fn()
        _toggle(False)
        b = fn()
    finally:
        _toggle(True)
    return a, b


def _rnd(be, *shape, scale=1.0, shift=0.0):
    return be((torch.randn(*shape) * scale + shift).to(BF))


SHAPES = [(2, 33, 31, 64), (1, 40, 52, 16), (3, 17, 23, 128), (2, 24, 40, 512), (1, 129, 67, 32)]


@pytest.mark.parametrize('shape', SHAPES)
@pytest.mark.parametrize('act,res_mode,xadd', [('relu', 'add', False), ('prelu', 'add', False), (None, 'gate', False),
                                               ('relu6', None, False), (None, None, True), ('prelu', None, False)])
def test_affine_act_fast(be, shape, act, res_mode, xadd):
    from led_net_amd import ops
    torch.manual_seed(3)
    C = shape[-1]
    x, res, xa = _rnd(be, *shape, scale=2), _rnd(be, *shape), _rnd(be, *shape)
    sc, sh, sl = be(torch.rand(C) + 0.5), be(torch.randn(C)), be(torch.rand(C) * 0.3)
    A = {'relu': ops.ACT_RELU, 'relu6': ops.ACT_RELU6, None: ops.ACT_NONE, 'prelu': ops.ACT_PRELU}[act]
    R = {'add': ops.RES_ADD, 'gate': ops.RES_GATE, None: ops.RES_NONE}[res_mode]

    def run():
        return ops.affine_act(x, sc, sh, act=A, slope=sl if act == 'prelu' else None, res=res if res_mode else None,
                              res_mode=R, xadd=xa if xadd else None)
    fast, gen = _both(run)
    v = x.float() + (xa.float() if xadd else 0)
    v = v * sc + sh
    if res_mode == 'add':
        v = v + res.float()
    elif res_mode == 'gate':
        v = v * res.float() + res.float()
    if act == 'relu':
        v = v.clamp(min=0)
    elif act == 'relu6':
        v = v.clamp(0, 6)
    elif act == 'prelu':
        v = torch.where(v > 0, v, v * sl)
    # bf16 output: one rounding of the same fp32 value -> at most 1 ulp from torch, identical between the kernels
    # up to fma contraction (1 ulp on rare elements)
    torch.testing.assert_close(fast.float().cpu(), v.cpu(), rtol=1e-2, atol=1e-2)
    assert (fast.float() - gen.float()).abs().max().item() <= 2e-2 * max(1.0, v.abs().max().item())
    assert (fast != gen).float().mean().item() < 0.02


@pytest.mark.parametrize('shape', SHAPES)
@pytest.mark.parametrize('act,res_mode', [('relu', 'add'), ('prelu', 'add'), (None, 'gate'), ('relu6', None),
                                          ('prelu', None), (None, None)])
def test_bn_act_bwd_fast(be, shape, act, res_mode):
    from led_net_amd import ops, ops_train as T
    torch.manual_seed(4)
    C = shape[-1]
    z, dy, res = _rnd(be, *shape, scale=2, shift=0.5), _rnd(be, *shape), _rnd(be, *shape)
    gamma, beta, sl = be(torch.rand(C) + 0.5), be(torch.randn(C)), be(torch.rand(C) * 0.3)
    A = {'relu': ops.ACT_RELU, 'relu6': ops.ACT_RELU6, None: ops.ACT_NONE, 'prelu': ops.ACT_PRELU}[act]
    R = {'add': ops.RES_ADD, 'gate': ops.RES_GATE, None: ops.RES_NONE}[res_mode]
    P = z.numel() // C

    def run():
        st = ops.channel_stats(z)
        scale, shift, mean, invstd = ops.bn_finalize(st, P, gamma, beta)
        out = T.bn_act_bwd(z, dy, scale=scale, shift=shift, mean=mean, invstd=invstd, act=A,
                           slope=sl if act == 'prelu' else None, res=res if res_mode else None, res_mode=R,
                           want_dres=res_mode is not None)
        return out, (st[0].clone(), st[1].clone())
    (fast, st_f), (gen, st_g) = _both(run)
    # statistics: f32 sums of the same bf16 values in a different order
    for a, b in zip(st_f, st_g):
        torch.testing.assert_close(a.cpu(), b.cpu(), rtol=1e-4, atol=1e-2)
    # torch reference on the same bf16 inputs, fp32 math
    zf = z.float().cpu().requires_grad_(True)
    rf = res.float().cpu().requires_grad_(True)
    g_, b_, s_ = (t.float().cpu().requires_grad_(True) for t in (gamma, beta, sl))
    zz = zf.reshape(-1, C)
    mean, var = zz.mean(0), zz.var(0, unbiased=False)
    v = (zf - mean) / torch.sqrt(var + 1e-5) * g_ + b_
    t = v + rf if res_mode == 'add' else (v * rf + rf if res_mode == 'gate' else v)
    y = {'relu': lambda a: a.clamp(min=0), 'relu6': lambda a: a.clamp(0, 6), None: lambda a: a,
         'prelu': lambda a: torch.where(a > 0, a, a * s_)}[act](t)
    y.backward(dy.float().cpu())
    names = ['dz', 'dres', 'dgamma', 'dbeta', 'dslope']
    want = [zf.grad, rf.grad if res_mode else None, g_.grad, b_.grad, s_.grad if act == 'prelu' else None]
    for n, f, g, w in zip(names, fast, gen, want):
        if w is None:
            continue
        f, g = f.float().cpu(), g.float().cpu()
        tol = 3e-2 * w.abs().max().item() + 1e-3      # bf16 outputs / f32 reductions of bf16 products
        assert (f - w).abs().max().item() <= tol, (n, (f - w).abs().max().item(), tol)
        assert (f - g).abs().max().item() <= tol, (n, 'fast vs generic', (f - g).abs().max().item(), tol)


@pytest.mark.parametrize('shape', SHAPES)
def test_channel_stats_fast(be, shape):
    from led_net_amd import ops
    torch.manual_seed(5)
    x, xa = _rnd(be, *shape, scale=2, shift=0.3), _rnd(be, *shape)
    for add in (None, xa):
        (f0, f1), (g0, g1) = _both(lambda: tuple(t.clone() for t in ops.channel_stats(x, xadd=add)))
        v = (x.float() + (add.float() if add is not None else 0)).reshape(-1, shape[-1]).double().cpu()
        torch.testing.assert_close(f0.double().cpu(), v.sum(0), rtol=1e-4, atol=2e-2)
        torch.testing.assert_close(f1.double().cpu(), (v * v).sum(0), rtol=1e-4, atol=2e-2)
        torch.testing.assert_close(f0.cpu(), g0.cpu(), rtol=1e-4, atol=2e-2)
        torch.testing.assert_close(f1.cpu(), g1.cpu(), rtol=1e-4, atol=2e-2)


# --------------------------------------------------------------------------- LDS-tiled depthwise 3x3 (dwconv.hip)
def _dw_ref(x, w, dil, group_size):
    """x [N,H,W,C] f32, w [3,3,C], per-group dilation -> depthwise conv, zero padding = dilation"""
    import torch.nn.functional as F
    N, H, W, C = x.shape
    xs = x.permute(0, 3, 1, 2)
    outs = []
    for g in range(C // group_size):
        sl = slice(g * group_size, (g + 1) * group_size)
        wg = w[:, :, sl].permute(2, 0, 1).unsqueeze(1)
        outs.append(F.conv2d(xs[:, sl], wg, padding=dil[g], dilation=dil[g], groups=group_size))
    return torch.cat(outs, 1).permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize('shape,dil', [((1, 128, 128, 64), (2, 2, 2, 2)), ((1, 70, 250, 64), (2, 3, 4, 5)),
                                       ((2, 96, 96, 32), (1, 1, 1, 1)), ((1, 64, 300, 128), (2, 3, 4, 5))])
def test_dw3x3_lds_tile_forward_stats_and_data_gradient(be, shape, dil):
    """the tiled kernel (>= 16384 pixels, W >= 32, C % 32 == 0) against torch's grouped conv2d: output, the
    per-channel sum / sum of squares of the epilogue, the flipped-tap data gradient with an addend; ragged tiles"""
    from led_net_amd import ops, ops_train as T
    N, H, W, C = shape
    gs = C // 4
    g = torch.Generator().manual_seed(H + W + C)
    x = torch.randn(shape, generator=g).to(torch.bfloat16)
    w = (0.3 * torch.randn(3, 3, C, generator=g))
    want = _dw_ref(x.float(), w, dil, gs)
    st = (torch.zeros(C, device=be.dev), torch.zeros(C, device=be.dev))
    y = ops.dwconv2d(x.to(be.dev), w.to(be.dev), stride=1, pad=-1, dil=dil, group_size=gs, stats=st)
    assert y.dtype == torch.bfloat16
    torch.testing.assert_close(y.float().cpu(), want, rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(st[0].cpu(), want.sum((0, 1, 2)), rtol=2e-3, atol=2e-2 * (N * H * W) ** 0.5)
    torch.testing.assert_close(st[1].cpu(), (want * want).sum((0, 1, 2)), rtol=2e-3, atol=1e-1)
    # data gradient = correlation with the flipped taps (same dilation) + the fan-in addend
    dz = torch.randn(shape, generator=g).to(torch.bfloat16)
    add = torch.randn(shape, generator=g).to(torch.bfloat16)
    wantdx = _dw_ref(dz.float(), w.flip(0, 1), dil, gs) + add.float()
    dx, _ = T.dwconv2d_bwd(x.to(be.dev), dz.to(be.dev), w.to(be.dev), stride=1, pad=-1, dil=dil, group_size=gs,
                           add=add.to(be.dev), need_dw=False)
    torch.testing.assert_close(dx.float().cpu(), wantdx, rtol=2e-2, atol=3e-2)


@pytest.mark.parametrize('shape', [(1, 128, 128, 32), (2, 100, 90, 16), (1, 70, 250, 64)])
def test_pyramid_bwd_lds_tile_equal_dilations(be, shape):
    """SESP pyramid backward on the spatial branch's shape class (dilations [1,1,1,1], >= 16384 pixels): the tiled
    data gradient (prefix-summed filters, no suffix pass) and the weight gradient that forms its suffix sums from dy,
    against torch autograd of the four cumulative depthwise convolutions; ragged tiles"""
    import torch.nn.functional as F
    from led_net_amd import ops_train as T
    N, H, W, n = shape
    g = torch.Generator().manual_seed(H * W + n)
    x = torch.randn(N, n, H, W, generator=g).to(torch.bfloat16).float().requires_grad_(True)
    ws = [(0.3 * torch.randn(n, 1, 3, 3, generator=g)).requires_grad_(True) for _ in range(4)]
    outs = []
    for i in range(4):
        o = F.conv2d(x, ws[i], padding=1, groups=n)
        outs.append(o if i == 0 else o + outs[-1])
    y = torch.cat(outs, 1)
    dy = torch.randn(y.shape, generator=g).to(torch.bfloat16).float()
    y.backward(dy)
    wp = torch.stack([w[:, 0].permute(1, 2, 0) for w in ws]).detach().contiguous()
    to = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(be.dev)     # noqa: E731
    dx, dw = T.sesp_pyramid_bwd(to(x), to(dy), wp.to(be.dev), [1, 1, 1, 1], 1)
    torch.testing.assert_close(dx.float().cpu().permute(0, 3, 1, 2), x.grad, rtol=2e-2, atol=6e-2)
    want_dw = torch.stack([w.grad[:, 0].permute(1, 2, 0) for w in ws])
    torch.testing.assert_close(dw.cpu(), want_dw, rtol=2e-2, atol=2e-2 * float(want_dw.abs().max()))


@pytest.mark.parametrize('shape', [(1, 64, 64, 128), (2, 40, 70, 32), (1, 72, 60, 64)])
def test_dw8x8_lds_tile_forward(be, shape):
    """GETB's 8x8 depthwise conv on the reflect-extended map (UNetFormer_GETB.py:201-204) with the patch in LDS, against
    torch (F.pad reflect (0,1,0,1) + conv2d padding 3): output and per-channel statistics; ragged tiles"""
    import torch.nn.functional as F
    from led_net_amd import ops
    N, H, W, C = shape
    g = torch.Generator().manual_seed(H * W + C)
    x = torch.randn(N, C, H, W, generator=g).to(torch.bfloat16).float()
    w = 0.1 * torch.randn(C, 1, 8, 8, generator=g)
    want = F.conv2d(F.pad(x, (0, 1, 0, 1), mode='reflect'), w, padding=3, groups=C).permute(0, 2, 3, 1)
    st = (torch.zeros(C, device=be.dev), torch.zeros(C, device=be.dev))
    xh = x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(be.dev)
    wk = w[:, 0].permute(1, 2, 0).contiguous().to(be.dev)
    y = ops.dwconv2d(xh, wk, stride=1, pad=3, dil=(1, 1, 1, 1), group_size=C, ext1=True, stats=st)
    assert tuple(y.shape) == (N, H, W, C)
    torch.testing.assert_close(y.float().cpu(), want, rtol=2e-2, atol=3e-2)
    torch.testing.assert_close(st[0].cpu(), want.sum((0, 1, 2)), rtol=5e-3, atol=3e-2 * (N * H * W) ** 0.5)
    torch.testing.assert_close(st[1].cpu(), (want * want).sum((0, 1, 2)), rtol=5e-3, atol=1e-1)


@pytest.mark.parametrize('shape', [(1, 64, 64, 128), (2, 40, 70, 32)])
def test_dw8x8_lds_tile_data_gradient(be, shape):
    """data gradient of GETB's 8x8 depthwise conv: tiled kernel with flipped taps on the extended map + the fold of the
    reflected row / column (adjoint of the reflect pad), with a fan-in addend, against torch autograd"""
    import torch.nn.functional as F
    from led_net_amd import ops_train as T
    N, H, W, C = shape
    g = torch.Generator().manual_seed(H + W + C)
    x = torch.randn(N, C, H, W, generator=g).to(torch.bfloat16).float().requires_grad_(True)
    w = 0.1 * torch.randn(C, 1, 8, 8, generator=g)
    y = F.conv2d(F.pad(x, (0, 1, 0, 1), mode='reflect'), w, padding=3, groups=C)
    dz = torch.randn(y.shape, generator=g).to(torch.bfloat16).float()
    y.backward(dz)
    add = torch.randn(N, H, W, C, generator=g).to(torch.bfloat16)
    to = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(be.dev)     # noqa: E731
    wk = w[:, 0].permute(1, 2, 0).contiguous().to(be.dev)
    dx, _ = T.dwconv2d_bwd(to(x), to(dz), wk, stride=1, pad=3, dil=(1, 1, 1, 1), group_size=C, ext1=True,
                           add=add.to(be.dev), need_dw=False)
    want = x.grad.permute(0, 2, 3, 1) + add.float()
    torch.testing.assert_close(dx.float().cpu(), want, rtol=2e-2, atol=5e-2)


@pytest.mark.parametrize('shape', [(1, 64, 64, 128), (2, 40, 70, 32)])
def test_dw8x8_lds_tile_weight_gradient(be, shape):
    """weight gradient of GETB's 8x8 depthwise conv (reflect-extended input, zero padding 3) from LDS tiles, against torch
    autograd; accumulates into a pre-filled buffer (gradient-sink semantics)"""
    import torch.nn.functional as F
    from led_net_amd import ops_train as T
    N, H, W, C = shape
    g = torch.Generator().manual_seed(H * 3 + W + C)
    x = torch.randn(N, C, H, W, generator=g).to(torch.bfloat16).float()
    w = (0.1 * torch.randn(C, 1, 8, 8, generator=g)).requires_grad_(True)
    y = F.conv2d(F.pad(x, (0, 1, 0, 1), mode='reflect'), w, padding=3, groups=C)
    dz = torch.randn(y.shape, generator=g).to(torch.bfloat16).float()
    y.backward(dz)
    to = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(be.dev)     # noqa: E731
    wk = w.detach()[:, 0].permute(1, 2, 0).contiguous().to(be.dev)
    pre = torch.full((8, 8, C), 0.5, device=be.dev)
    _, dw = T.dwconv2d_bwd(to(x), to(dz), wk, stride=1, pad=3, dil=(1, 1, 1, 1), group_size=C, ext1=True,
                           need_dx=False, dw_out=pre)
    want = w.grad[:, 0].permute(1, 2, 0) + 0.5
    torch.testing.assert_close(dw.cpu(), want, rtol=2e-2, atol=2e-2 * float(want.abs().max()))


# --------------------------------------------------------------------------- #
# one-pass BatchNorm backward (bn_bwd_fused_kernel: persistent workgroups, z held in LDS across a grid-wide barrier)
# against the two-kernel form on the same tensors -- GPU only: the emulator runs workgroups one after another, so the
# entry point reports LEDN_ESKIP there and ops_train falls back (asserted below).
# --------------------------------------------------------------------------- #
def _bn_bwd_case(dev, C, P, act, res_mode, want_dres, adds, seed):
    from led_net_amd import ops
    g = torch.Generator().manual_seed(seed)
    z = torch.randn(P, C, generator=g).bfloat16().to(dev).view(1, P, 1, C)
    dy = torch.randn(P, C, generator=g).bfloat16().to(dev).view(1, P, 1, C)
    res = torch.randn(P, C, generator=g).bfloat16().to(dev).view(1, P, 1, C) if res_mode != ops.RES_NONE else None
    mean = z.float().mean((0, 1, 2))
    var = z.float().var((0, 1, 2), unbiased=False)
    invstd = (var + 1e-5).rsqrt()
    gamma = (torch.rand(C, generator=g) + 0.5).to(dev)
    beta = (torch.randn(C, generator=g) * 0.1).to(dev)
    scale, shift = gamma * invstd, beta - mean * gamma * invstd
    slope = (torch.rand(C, generator=g) * 0.3).to(dev) if act == ops.ACT_PRELU else None
    dz_add = torch.randn(P, C, generator=g).bfloat16().to(dev).view(1, P, 1, C) if adds else None
    dres_add = torch.randn(P, C, generator=g).bfloat16().to(dev).view(1, P, 1, C) if (adds and want_dres) else None
    return dict(z=z, dy=dy, scale=scale, shift=shift, mean=mean, invstd=invstd, act=act, slope=slope, res=res,
                res_mode=res_mode, count=P, want_dres=want_dres, dz_add=dz_add, dres_add=dres_add)


@pytest.mark.gpu
@pytest.mark.parametrize('C,P', [(64, 262144), (128, 65536), (16, 262144), (32, 100000), (64, 70001)])
@pytest.mark.parametrize('variant', ['prelu', 'relu_res_dres', 'none_res_dres_adds', 'prelu_res', 'none'])
def test_bn_bwd_fused_equals_two_kernels(C, P, variant):
    from led_net_amd import ops, ops_train as T
    dev = torch.device('cuda:0')
    act = {'prelu': ops.ACT_PRELU, 'relu_res_dres': ops.ACT_RELU, 'none_res_dres_adds': ops.ACT_NONE,
           'prelu_res': ops.ACT_PRELU, 'none': ops.ACT_NONE}[variant]
    res_mode = ops.RES_ADD if 'res' in variant else ops.RES_NONE
    kw = _bn_bwd_case(dev, C, P, act, res_mode, 'dres' in variant, 'adds' in variant, 7)
    z, dy = kw.pop('z'), kw.pop('dy')
    outs = []
    for fused in (0, 1):
        T.set_bn_fused(fused)
        try:
            sinks = (torch.zeros(C, device=dev), torch.zeros(C, device=dev),
                     torch.zeros(C, device=dev) if act == ops.ACT_PRELU else None)
            dz, dres, _, _, _ = T.bn_act_bwd(z, dy, sinks=sinks, **kw)
            torch.cuda.synchronize()
            if fused:
                assert _lib_check(C) == 0, 'the fused launch gave up at its barrier'
            outs.append((dz.float().cpu(), None if dres is None else dres.float().cpu(), [None if s is None else s.cpu() for s in sinks]))
        finally:
            T.set_bn_fused(0)
    (dz0, dr0, s0), (dz1, dr1, s1) = outs
    for a, b in zip(s0, s1):
        if a is not None:
            torch.testing.assert_close(b, a, rtol=2e-4, atol=2e-3 * max(1.0, a.abs().max().item()) * 1e-1)
    # dz / dres: the same arithmetic on coefficients that differ by the summation order of the totals: bf16 outputs agree
    # except for a rounding flip here and there
    assert (dz1 - dz0).abs().max().item() <= 2e-2 * max(1.0, dz0.abs().max().item())
    assert ((dz1 - dz0).abs() > 0).float().mean().item() < 0.02
    if dr0 is not None:
        assert torch.equal(dr0, dr1)


def _lib_check(C):
    from led_net_amd import _lib
    lib = _lib.get_lib()
    stream = torch.cuda.current_stream().cuda_stream
    return int(lib.cdll.ledn_bn_act_bwd_fused_check(C, stream))


def test_bn_bwd_fused_is_skipped_on_the_emulator(emu):
    """no grid barrier on the emulator: the entry reports LEDN_ESKIP and ops_train runs reduce + apply"""
    from led_net_amd import _lib, ops, ops_train as T
    lib = _lib.get_lib()
    lib.set_option(_lib.OPT_BN_FUSED, 1)
    try:
        kw = _bn_bwd_case(torch.device('cpu'), 16, 4096, ops.ACT_NONE, ops.RES_NONE, False, False, 3)
        st = T.bn_act_bwd_reduce(kw['z'], kw['dy'], scale=kw['scale'], shift=kw['shift'], mean=kw['mean'], invstd=kw['invstd'],
                                 count=kw['count'], launch=False)
        assert lib.cdll.ledn_bn_act_bwd_fused(st.d, None) == _lib.ESKIP
    finally:
        lib.set_option(_lib.OPT_BN_FUSED, 0)


@pytest.mark.parametrize('C,P,act', [(32, 70001, 'relu'), (64, 40000, 'prelu'), (16, 33000, 'none'), (128, 4099, 'relu')])
def test_affine_with_output_statistics(be, C, P, act):
    """ledn_affine_act with stat_sum / stat_sqsum: the output of the plain pass and the per-channel sums ledn_channel_stats
    finds on it (the statistics are those of the STORED bf16 values), accumulated into the sinks; ragged vector counts"""
    from led_net_amd import ops
    g = torch.Generator().manual_seed(C + P)
    x = torch.randn(1, P, 1, C, generator=g).bfloat16().to(be.dev)
    sc, sh = (torch.rand(C, generator=g) + 0.5).to(be.dev), (torch.randn(C, generator=g) * 0.3).to(be.dev)
    a = {'relu': ops.ACT_RELU, 'prelu': ops.ACT_PRELU, 'none': ops.ACT_NONE}[act]
    sl = (torch.rand(C, generator=g) * 0.3).to(be.dev) if act == 'prelu' else None
    assert ops.affine_stats_ok(x, a)
    want = ops.affine_act(x, sc, sh, act=a, slope=sl)
    ws = (torch.zeros(C, device=be.dev), torch.zeros(C, device=be.dev))
    ops.channel_stats(want, stats=ws)
    st = (torch.full((C,), 0.5, device=be.dev), torch.full((C,), 0.25, device=be.dev))
    got = ops.affine_act(x, sc, sh, act=a, slope=sl, stats=st)
    assert torch.equal(got, want)
    torch.testing.assert_close(st[0] - 0.5, ws[0], rtol=1e-4, atol=1e-3 * float(ws[0].abs().max()) + 1e-3)
    torch.testing.assert_close(st[1] - 0.25, ws[1], rtol=1e-4, atol=1e-4 * float(ws[1].abs().max()) + 1e-3)
