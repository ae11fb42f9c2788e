"""Vectorised bf16 stencil kernels (csrc/stencil.h, stencil_bf16.hip, dwconv.hip dw3x3_bf16_kernel):
SESP pyramid and dilated depthwise 3x3, forward / data gradient / weight gradient, against torch
on the same bf16-rounded operands.  Emulator on CPU, MI355X with -m gpu."""
import pytest
import torch
import torch.nn.functional as F

torch.manual_seed(5)
_DEV = [torch.device('cpu')]


@pytest.fixture(autouse=True)
def _track_device(request):
    _DEV[0] = request.getfixturevalue('be').dev if 'be' in request.fixturenames else torch.device('cpu')
    yield


def D(t):
    return t.to(_DEV[0])


def nhwc16(t):
    return D(t.detach().permute(0, 2, 3, 1).contiguous()).bfloat16()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous().cpu().float()


def r16(t):
    return t.bfloat16().float()


def close16(got, want, tol=1.5e-2):
    """bf16 output rounding: relative to the tensor's scale"""
    scale = float(want.abs().max()) + 1e-6
    err = float((got - want).abs().max())
    assert err <= tol * scale, (err, scale)


@pytest.mark.parametrize('n,hw,dil', [(8, (13, 21), (1, 2, 3, 4)), (16, (9, 40), (2, 3, 4, 5)), (64, (6, 7), (1, 1, 1, 1)),
                                      (32, (17, 5), (1, 2, 3, 4))])
def test_dw3x3_bf16_fwd_bwd(be, n, hw, dil):
    from led_net_amd import ops, ops_train as T
    c = 4 * n
    x = r16(torch.randn(3, c, *hw)).requires_grad_(True)
    ws = [(torch.randn(n, 1, 3, 3) * 0.3).requires_grad_(True) for _ in range(4)]
    z = torch.cat([F.conv2d(x[:, i * n:(i + 1) * n], ws[i], padding=dil[i], dilation=dil[i], groups=n)
                   for i in range(4)], 1)
    wp = torch.cat([w[:, 0].permute(1, 2, 0) for w in ws], 2).detach().contiguous()
    s, b, sl = torch.rand(c) + 0.5, torch.randn(c), torch.rand(c) * 0.3
    v = z * s.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)
    stats = (D(torch.zeros(c)), D(torch.zeros(c)))
    got = ops.dwconv2d(nhwc16(x), D(wp), dil=list(dil), group_size=n, out_scale=D(s), out_shift=D(b),
                       act=ops.ACT_PRELU, slope=D(sl), stats=stats)
    assert got.dtype == torch.bfloat16
    close16(nchw(got), F.prelu(v, sl).detach())
    torch.testing.assert_close(stats[0].cpu(), v.detach().sum((0, 2, 3)), rtol=1e-3, atol=1e-2)
    torch.testing.assert_close(stats[1].cpu(), (v.detach() ** 2).sum((0, 2, 3)), rtol=1e-3, atol=1e-2)
    raw = ops.dwconv2d(nhwc16(x), D(wp), dil=list(dil), group_size=n)     # training form: no epilogue
    close16(nchw(raw), z.detach())
    dz = r16(torch.randn_like(z))
    add = r16(torch.randn_like(x))
    z.backward(dz)
    dx, dw = T.dwconv2d_bwd(nhwc16(x), nhwc16(dz), D(wp), dil=list(dil), group_size=n, add=nhwc16(add))
    close16(nchw(dx), x.grad + add)
    want_dw = torch.cat([w.grad[:, 0].permute(1, 2, 0) for w in ws], 2)
    torch.testing.assert_close(dw.cpu(), want_dw, rtol=2e-3, atol=2e-3 * float(want_dw.abs().max()))


@pytest.mark.parametrize('n,hw,stride', [(8, (13, 10), 1), (16, (9, 33), 1), (64, (5, 6), 1), (16, (13, 10), 2),
                                         (32, (8, 9), 2)])
def test_sesp_pyramid_bf16_fwd_bwd(be, n, hw, stride):
    from led_net_amd import ops, ops_train as T
    dil = [1, 2, 3, 4]
    x = r16(torch.randn(2, n, *hw)).requires_grad_(True)
    ws = [(torch.randn(n, 1, 3, 3) * 0.3).requires_grad_(True) for _ in range(4)]
    outs = []
    for i in range(4):
        o = F.conv2d(x, ws[i], stride=stride, padding=dil[i], dilation=dil[i], groups=n)
        outs.append(o if i == 0 else o + outs[-1])
    y = torch.cat(outs, 1)
    wp = torch.stack([w[:, 0].permute(1, 2, 0) for w in ws]).detach().contiguous()
    got = ops.sesp_pyramid(nhwc16(x), D(wp), dil, stride)
    assert got.dtype == torch.bfloat16
    close16(nchw(got), y.detach())
    dy = r16(torch.randn_like(y))
    y.backward(dy)
    dx, dw = T.sesp_pyramid_bwd(nhwc16(x), nhwc16(dy), D(wp), dil, stride)
    # the suffix sums g_b are rounded to bf16 before the gather (gsum scratch is bf16)
    close16(nchw(dx), x.grad, 3e-2)
    want_dw = torch.stack([w.grad[:, 0].permute(1, 2, 0) for w in ws])
    torch.testing.assert_close(dw.cpu(), want_dw, rtol=2e-2, atol=2e-2 * float(want_dw.abs().max()))
