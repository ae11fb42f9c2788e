"""The reference-precision (f32) convolutions on the matrix cores (csrc/conv_f32.hip: v_mfma_f32_32x32x2_f32) against
torch.nn.functional.conv2d / autograd in f32 -- forward with prologue + affine + statistics + residual + activation,
the data gradient (transposed), grouped 1x1 layers, strides, odd sizes, and the weight / bias gradient.  Tolerance:
f32 products and f32 accumulation on both sides, another summation order: rtol 2e-4 on O(1) values (K up to 1152).
Runs on the emulator (MFMA emulated lane-exactly) and on the MI355X.  Reference semantics: F.conv2d inside mmcv's
ConvModule (mmseg/models/utils/basic_block.py:43-57), nn_layers/espnet_utils.py:22-36 (grouped 1x1)."""
import pytest
import torch
import torch.nn.functional as F

torch.manual_seed(4)
_DEV = [torch.device('cpu')]


@pytest.fixture(autouse=True)
def _track_device(request):
    _DEV[0] = request.getfixturevalue('be').dev if 'be' in request.fixturenames else torch.device('cpu')
    yield


def D(t):
    return t.to(_DEV[0])


def nhwc(t):
    return D(t.detach().permute(0, 2, 3, 1).contiguous())


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous().cpu().float()


CASES = [  # cin, cout, k, stride, groups, (N, H, W)
    (32, 32, 3, 1, 1, (2, 37, 41)), (32, 32, 3, 2, 1, (2, 67, 70)), (64, 64, 3, 1, 1, (2, 33, 40)),
    (64, 128, 3, 2, 1, (2, 64, 70)), (128, 64, 3, 1, 1, (1, 47, 48)), (16, 64, 1, 1, 1, (2, 40, 33)),
    (64, 64, 1, 1, 4, (2, 36, 40)), (64, 16, 1, 1, 4, (2, 36, 40)), (128, 128, 1, 1, 4, (1, 50, 48)),
    (64, 48, 1, 1, 1, (2, 33, 37)), (32, 64, 1, 2, 1, (2, 67, 70)), (128, 384, 1, 1, 1, (1, 47, 50)),
]


def _uses(d_kind, *a, **k):
    from led_net_amd import ops
    return ops.conv2d(*a, _query=True, **k) if d_kind == 'conv' else ops.conv2d_wgrad(*a, _query=True, **k)


@pytest.mark.parametrize('cin,cout,k,stride,groups,nhw', CASES)
def test_f32_mfma_conv_forward(be, cin, cout, k, stride, groups, nhw):
    from led_net_amd import ops
    N, H, W = nhw
    x = torch.randn(N, cin, H, W)
    w = torch.randn(cout, cin // groups, k, k) / (cin // groups * k * k) ** 0.5
    s_in, b_in, sl_in = torch.rand(cin) + 0.5, torch.randn(cin) * 0.1, torch.rand(cin) * 0.3
    s_o, b_o, sl = torch.rand(cout) + 0.5, torch.randn(cout) * 0.1, torch.rand(cout) * 0.3
    pad = k // 2
    xin = F.prelu(x * s_in.view(1, -1, 1, 1) + b_in.view(1, -1, 1, 1), sl_in)
    z = F.conv2d(xin, w, stride=stride, padding=pad, groups=groups)
    v = z * s_o.view(1, -1, 1, 1) + b_o.view(1, -1, 1, 1)
    res = torch.randn_like(v)
    want = F.prelu(v + res, sl)
    stats = (D(torch.zeros(cout)), D(torch.zeros(cout)))
    kw = dict(stride=stride, pad=pad, groups=groups, in_scale=D(s_in), in_shift=D(b_in), in_act=ops.ACT_PRELU,
              in_slope=D(sl_in), out_scale=D(s_o), out_shift=D(b_o), act=ops.ACT_PRELU, slope=D(sl),
              res=nhwc(res), res_mode=ops.RES_ADD)
    assert ops.conv2d(nhwc(x), D(w), _query=True, **kw) == 5, 'not on conv_f32_mfma_kernel'
    got = ops.conv2d(nhwc(x), D(w), stats=stats, **kw)
    assert got.dtype == torch.float32
    torch.testing.assert_close(nchw(got), want, rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(stats[0].cpu(), v.sum((0, 2, 3)), rtol=1e-3, atol=1e-3 * v.abs().sum((0, 2, 3)).max().item())
    torch.testing.assert_close(stats[1].cpu(), (v * v).sum((0, 2, 3)), rtol=1e-3, atol=1e-2)
    # plain call (no prologue / epilogue): the raw convolution
    got2 = ops.conv2d(nhwc(x), D(w), stride=stride, pad=pad, groups=groups)
    torch.testing.assert_close(nchw(got2), F.conv2d(x, w, stride=stride, padding=pad, groups=groups), rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize('cin,cout,k,stride,groups,nhw', CASES)
def test_f32_mfma_conv_dgrad_wgrad(be, cin, cout, k, stride, groups, nhw):
    from led_net_amd import ops
    N, H, W = nhw
    pad = k // 2
    x = torch.randn(N, cin, H, W, requires_grad=True)
    w = (torch.randn(cout, cin // groups, k, k) / (cin // groups * k * k) ** 0.5).requires_grad_(True)
    b = torch.zeros(cout, requires_grad=True)
    z = F.conv2d(x, w, b, stride=stride, padding=pad, groups=groups)
    dz = torch.randn_like(z)
    z.backward(dz)
    add = torch.randn(N, cin, H, W)
    # data gradient = the transposed launch, with the fan-in addend as the epilogue's residual
    kw = dict(stride=stride, pad=pad, groups=groups, transposed=True, out_hw=(H, W), res=nhwc(add), res_mode=ops.RES_ADD)
    if N * H * W >= 2048 and cout % 8 == 0 and (groups == 1 or (cout // groups) % 8 == 0) and cin >= 16:
        assert ops.conv2d(nhwc(dz), D(w.detach()), _query=True, **kw) == 5
    dx = ops.conv2d(nhwc(dz), D(w.detach()), **kw)
    torch.testing.assert_close(nchw(dx), x.grad + add, rtol=2e-4, atol=2e-4)
    # weight / bias gradient, accumulated into caller buffers
    dw0, db0 = torch.randn_like(w.detach()), torch.randn(cout)
    dw, db = D(dw0.clone()), D(db0.clone())
    wk = dict(stride=stride, pad=pad, groups=groups, bias=True)
    assert ops.conv2d_wgrad(nhwc(x), nhwc(dz), tuple(w.shape), _query=True, **wk) == 4, 'not on conv_wgrad_f32_mfma_kernel'
    ops.conv2d_wgrad(nhwc(x), nhwc(dz), tuple(w.shape), dw_out=dw, db_out=db, **wk)
    scale = w.grad.abs().max().item()
    torch.testing.assert_close(dw.cpu() - dw0, w.grad, rtol=1e-3, atol=2e-4 * max(1.0, scale))
    torch.testing.assert_close(db.cpu() - db0, b.grad, rtol=1e-3, atol=1e-3 * max(1.0, b.grad.abs().max().item()))


def test_f32_mfma_wgrad_with_prologue(be):
    """BNActConvFn's weight gradient: dW of conv(prelu(x * s + b)), and the xadd form"""
    from led_net_amd import ops
    x = torch.randn(2, 64, 36, 40)
    xa = torch.randn_like(x)
    s_in, b_in, sl_in = torch.rand(64) + 0.5, torch.randn(64) * 0.1, torch.rand(64) * 0.3
    w = (torch.randn(64, 64, 3, 3) / 24).requires_grad_(True)
    pre = F.prelu((x + xa) * s_in.view(1, -1, 1, 1) + b_in.view(1, -1, 1, 1), sl_in)
    z = F.conv2d(pre, w, padding=1)
    dz = torch.randn_like(z)
    z.backward(dz)
    dw, _ = ops.conv2d_wgrad(nhwc(x), nhwc(dz), tuple(w.shape), stride=1, pad=1, xadd=nhwc(xa), in_scale=D(s_in),
                             in_shift=D(b_in), in_act=ops.ACT_PRELU, in_slope=D(sl_in))
    torch.testing.assert_close(dw.cpu(), w.grad, rtol=1e-3, atol=2e-4 * w.grad.abs().max().item())


def test_f32_mfma_equals_the_direct_kernel_to_rounding(be):
    """the VALU kernel (conv_direct_kernel) and the matrix-core kernel on the same f32 operands: both are fmaf chains
    over K, in different orders -- agreement to a few ulp of the accumulated magnitude"""
    from led_net_amd import ops
    x = torch.randn(1, 32, 48, 48)
    w = torch.randn(32, 32, 3, 3) / 17
    got = ops.conv2d(nhwc(x), D(w), stride=1, pad=1)
    # Cout 8 < 16 keeps a twin launch on the direct kernel: compare the shared 8 output channels
    ref = ops.conv2d(nhwc(x), D(w[:8].contiguous()), stride=1, pad=1)
    assert ops.conv2d(nhwc(x), D(w[:8].contiguous()), stride=1, pad=1, _query=True) == 0
    torch.testing.assert_close(got[..., :8].cpu(), ref.cpu(), rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize('cin,cout,stride,nhw', [(3, 32, 2, (2, 80, 84)), (5, 16, 1, (1, 48, 48))])
def test_f32_mfma_wgrad_narrow_channels(be, cin, cout, stride, nhw):
    """the 3 -> 32 stem: lanes beyond Cin carry zeros (Cout < 16 and Cin < 3 stay on the VALU kernels)"""
    from led_net_amd import ops
    N, H, W = nhw
    x = torch.randn(N, cin, H, W)
    w = (torch.randn(cout, cin, 3, 3) / (cin * 9) ** 0.5).requires_grad_(True)
    b = torch.zeros(cout, requires_grad=True)
    z = F.conv2d(x, w, b, stride=stride, padding=1)
    dz = torch.randn_like(z)
    z.backward(dz)
    wk = dict(stride=stride, pad=1, bias=True)
    assert ops.conv2d_wgrad(nhwc(x), nhwc(dz), tuple(w.shape), _query=True, **wk) == 4
    dw, db = ops.conv2d_wgrad(nhwc(x), nhwc(dz), tuple(w.shape), **wk)
    torch.testing.assert_close(dw.cpu(), w.grad, rtol=1e-3, atol=2e-4 * max(1.0, w.grad.abs().max().item()))
    torch.testing.assert_close(db.cpu(), b.grad, rtol=1e-3, atol=1e-3 * max(1.0, b.grad.abs().max().item()))
