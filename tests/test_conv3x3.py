"""Register-direct 3x3 stride-1 convolution on the matrix cores (csrc/conv3x3.hip): forward (raw, + bias, + channel
statistics, inference epilogue), data gradient (raw, + fan-in addend), 32 input channels, 32 / 64 / 128 output
channels, ragged widths and heights (strips, segments, image borders) -- against torch on bf16-rounded operands and
against the general MFMA kernel (conv_mfma.hip: LEDN_OPT_STREAM_FAST bit 6 off).  Runs on the emulator
(v_mfma_f32_16x16x32_bf16 and the DPP row shifts emulated lane-exactly) and on the GPU.

Reference call sites these replace: the 3x3 ConvModules of mmseg/models/utils/basic_block.py:43-57 (BasicBlock conv1 /
conv2 of the DDRNet stem, ddrnet.py:123-149)."""
import pytest
import torch
import torch.nn.functional as F

torch.manual_seed(7)
_DEV = [torch.device('cpu')]
MASK = 27 + 64          # LEDN_OPT_STREAM_FAST with the 3x3 register kernel on


@pytest.fixture(autouse=True)
def _track_device(request):
    _DEV[0] = request.getfixturevalue('be').dev if 'be' in request.fixturenames else torch.device('cpu')
    from led_net_amd import _lib
    lib = _lib.get_lib()
    lib.set_option(2, MASK)
    yield
    lib.set_option(2, -1)


def D(t):
    return t.to(_DEV[0])


def nhwc(t):
    return D(t.detach().permute(0, 2, 3, 1).contiguous())


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous().cpu().float()


def r16(t):
    return t.bfloat16().float()


# cin, cout, (N, H, W): widths that are / are not multiples of the strip, heights around the 8 / 16-row segments
CASES = [(32, 32, (2, 9, 33)), (32, 32, (1, 17, 16)), (32, 64, (1, 5, 40)), (32, 32, (2, 8, 32)), (32, 32, (1, 19, 13)),
         (32, 128, (1, 3, 17)), (32, 32, (1, 1, 1)), (32, 64, (1, 2, 50)), (32, 32, (1, 35, 70))]


@pytest.mark.parametrize('cin,cout,nhw', CASES)
def test_conv3x3_forward(be, cin, cout, nhw):
    from led_net_amd import ops, _lib
    N, H, W = nhw
    x = r16(torch.randn(N, cin, H, W))
    w = torch.randn(cout, cin, 3, 3) / (9 * cin) ** 0.5
    b = torch.randn(cout) * 0.3
    want = F.conv2d(x, r16(w), b, padding=1)
    wp = ops.pack_conv_weights(D(w), 0, 1)
    xb = nhwc(x).bfloat16()
    assert ops.conv2d_kernel_id(xb, D(w), pad=1, w_bf16=wp, out_shift=D(b)) == 3
    got = ops.conv2d(xb, D(w), pad=1, out_shift=D(b), w_bf16=wp)
    torch.testing.assert_close(nchw(got), want, rtol=1e-2, atol=2e-2)
    stats = (D(torch.zeros(cout)), D(torch.zeros(cout)))
    got2 = ops.conv2d(xb, D(w), pad=1, out_shift=D(b), stats=stats, w_bf16=wp)
    torch.testing.assert_close(nchw(got2), want, rtol=1e-2, atol=2e-2)
    npx = N * H * W
    torch.testing.assert_close(stats[0].cpu(), want.sum((0, 2, 3)), rtol=1e-3, atol=2e-3 * npx)
    torch.testing.assert_close(stats[1].cpu(), (want * want).sum((0, 2, 3)), rtol=2e-3, atol=2e-3 * npx)
    # inference epilogue: y = relu(z * scale + shift + res)
    sc, sh = torch.rand(cout) + 0.5, torch.randn(cout) * 0.2
    res = r16(torch.randn(N, cout, H, W))
    want3 = torch.relu(F.conv2d(x, r16(w), None, padding=1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) + res)
    got3 = ops.conv2d(xb, D(w), pad=1, out_scale=D(sc), out_shift=D(sh), act=ops.ACT_RELU, res=nhwc(res).bfloat16(),
                      res_mode=ops.RES_ADD, w_bf16=wp)
    torch.testing.assert_close(nchw(got3), want3, rtol=1e-2, atol=3e-2)
    # the general MFMA kernel on the same operands: same products, same f32 accumulation up to order
    lib = _lib.get_lib()
    lib.set_option(2, 27)
    try:
        assert ops.conv2d_kernel_id(xb, D(w), pad=1, w_bf16=wp, out_shift=D(b)) == 1
        ref = ops.conv2d(xb, D(w), pad=1, out_shift=D(b), w_bf16=wp)
    finally:
        lib.set_option(2, MASK)
    torch.testing.assert_close(got.float().cpu(), ref.float().cpu(), rtol=8e-3, atol=2e-3)


@pytest.mark.parametrize('cin,cout,nhw', CASES)
def test_conv3x3_dgrad_with_addend(be, cin, cout, nhw):
    """data gradient of a 3x3 stride-1 layer with `cout` inputs and `cin` outputs (mode-1 pack: flipped taps, the
    kernel's output channels = the forward's input channels), alone and with another consumer's partial gradient
    added in the epilogue"""
    from led_net_amd import ops
    N, H, W = nhw
    x = r16(torch.randn(N, cout, H, W)).requires_grad_(True)        # the layer maps cout -> cin here
    w = r16(torch.randn(cin, cout, 3, 3) / (9 * cout) ** 0.5).requires_grad_(True)
    z = F.conv2d(x, w, padding=1)
    dz = r16(torch.randn_like(z))
    z.backward(dz)
    wp1 = ops.pack_conv_weights(D(w.detach()), 1, 1)
    dzb = nhwc(dz).bfloat16()
    if True:
        assert ops.conv2d_kernel_id(dzb, D(w.detach()), pad=1, transposed=True, out_hw=(H, W), w_bf16=wp1) == 3
    dx = ops.conv2d(dzb, D(w.detach()), pad=1, transposed=True, out_hw=(H, W), w_bf16=wp1)
    scale = float(x.grad.abs().max())
    torch.testing.assert_close(nchw(dx), x.grad, rtol=1e-2, atol=1e-2 * scale)
    prev = r16(torch.randn(N, cout, H, W))
    dx2 = ops.conv2d(dzb, D(w.detach()), pad=1, transposed=True, out_hw=(H, W), w_bf16=wp1,
                     res=nhwc(prev).bfloat16(), res_mode=ops.RES_ADD)
    torch.testing.assert_close(nchw(dx2), r16(x.grad) + prev, rtol=1e-2, atol=1.5e-2 * max(scale, 1.0))


@pytest.mark.parametrize('wgs', [1, 5])
def test_conv3x3_many_tasks_per_wave(be, wgs):
    """LEDN_OPT_CONV_WORKGROUPS = 1 / 5: every wave walks several (image, segment, strip) tasks"""
    from led_net_amd import ops, _lib
    lib = _lib.get_lib()
    lib.set_option(0, wgs)
    try:
        x = r16(torch.randn(3, 32, 37, 70))
        w = torch.randn(32, 32, 3, 3) / 17.0
        want = F.conv2d(x, r16(w), None, padding=1)
        wp = ops.pack_conv_weights(D(w), 0, 1)
        stats = (D(torch.zeros(32)), D(torch.zeros(32)))
        got = ops.conv2d(nhwc(x).bfloat16(), D(w), pad=1, stats=stats, w_bf16=wp)
        torch.testing.assert_close(nchw(got), want, rtol=1e-2, atol=2e-2)
        torch.testing.assert_close(stats[0].cpu(), want.sum((0, 2, 3)), rtol=1e-3, atol=2e-3 * 3 * 37 * 70)
    finally:
        lib.set_option(0, 0)


@pytest.mark.parametrize('cout,pro,f32out,nhw', [(2, True, False, (2, 13, 45)), (2, True, True, (1, 17, 32)), (2, False, False, (1, 9, 33)),
                                                 (4, True, False, (1, 20, 19)), (1, True, True, (1, 5, 70))])
def test_conv3x3_narrow_head_forward(be, cout, pro, f32out, nhw):
    """the two-class heads' convolution in training (led_head.py:44-51,84-99: BatchNorm -> ReLU -> conv3x3 32 -> 2, then its
    own BatchNorm): input prologue folded in (zero padding AFTER the activation), bias, raw bf16 / f32 output --
    conv3x3_reg_kernel<.., C33_NARROW> against torch and against conv_mfma_kernel's narrow epilogue"""
    from led_net_amd import ops, _lib
    N, H, W = nhw
    x = r16(torch.randn(N, 32, H, W))
    w = r16(torch.randn(cout, 32, 3, 3) / 17.0)
    b = torch.randn(cout) * 0.3
    s_in, b_in = torch.rand(32) + 0.5, torch.randn(32) * 0.3
    xin = r16(F.relu(x * s_in.view(1, -1, 1, 1) + b_in.view(1, -1, 1, 1))) if pro else x
    want = F.conv2d(xin, w, b, padding=1)
    kw = dict(pad=1, out_shift=D(b), w_bf16=ops.pack_conv_weights(D(w), 0))
    lib = _lib.get_lib()
    lib.set_option(2, MASK + 128)          # (the narrow variant is opt-in: bit 7)
    if pro:
        kw.update(in_scale=D(s_in), in_shift=D(b_in), in_act=ops.ACT_RELU)
    if f32out:
        kw.update(out_dtype=torch.float32)
    xb = nhwc(x).bfloat16()
    assert ops.conv2d_kernel_id(xb, D(w), **kw) == 3
    got = ops.conv2d(xb, D(w), **kw)
    assert got.dtype == (torch.float32 if f32out else torch.bfloat16)
    tol = dict(rtol=1e-3, atol=2e-3) if f32out else dict(rtol=1e-2, atol=2e-2)
    torch.testing.assert_close(nchw(got), want, **tol)
    lib.set_option(2, 27)
    try:
        assert ops.conv2d_kernel_id(xb, D(w), **kw) == 1
        ref = ops.conv2d(xb, D(w), **kw)
    finally:
        lib.set_option(2, MASK)
    torch.testing.assert_close(got.float().cpu(), ref.float().cpu(), rtol=8e-3, atol=2e-3)
