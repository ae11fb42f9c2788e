"""Register-direct 1x1 convolution on the matrix cores (csrc/conv1x1.hip): forward (raw, + bias, + channel
statistics), data gradient (raw, + fan-in addend), dense and grouped, ragged pixel counts -- against torch on
bf16-rounded operands and against the general MFMA kernel (conv_mfma.hip, selected with LEDN_OPT_STREAM_FAST
bit 4 off).  Runs on the emulator (v_mfma_f32_16x16x32_bf16 emulated lane-exactly) and on the GPU.

Reference call sites these replace: the 1x1 ConvModules / CBR / C of nn_layers/espnet_utils.py:22-36 (SESP
proj_1x1, conv_1x1_exp: eesp.py:39,70), classification/model_utils.py:360-400 (Muti_AFF local / context MLPs),
backbones/UNetFormer_GETB.py:83-85,111 (qkv, Mlp), ddrnet.py:68-75 (compression)."""
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


def nhwc(t):
    return D(t.detach().permute(0, 2, 3, 1).contiguous())


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous().cpu().float()


def r16(t):
    return t.bfloat16().float()


def _uses_reg_kernel(x, w, groups, wp, **kw):
    """the C ABI's own answer: 2 = conv1x1_mfma_kernel"""
    from led_net_amd import ops
    return ops.conv2d_kernel_id(x, w, groups=groups, w_bf16=wp, **kw)


# cin, cout, groups, (N, H, W)  -- pixel counts that are / are not multiples of the 32-pixel iteration
CASES = [
    (32, 32, 1, (2, 9, 33)), (64, 64, 1, (2, 8, 32)), (64, 64, 4, (1, 13, 27)), (64, 16, 4, (2, 7, 19)),
    (16, 64, 4, (2, 5, 41)), (16, 64, 1, (1, 6, 16)), (128, 64, 1, (1, 9, 17)), (64, 128, 1, (1, 11, 16)),
    (128, 128, 4, (2, 6, 21)), (128, 32, 1, (1, 8, 24)), (32, 128, 1, (1, 5, 23)), (128, 16, 1, (1, 4, 20)),
    (48, 48, 1, (1, 7, 13)),
]


@pytest.mark.parametrize('cin,cout,g,nhw', CASES)
def test_conv1x1_forward(be, cin, cout, g, nhw):
    from led_net_amd import ops, _lib
    N, H, W = nhw
    x = r16(torch.randn(N, cin, H, W))
    w = torch.randn(cout, cin // g, 1, 1) / (cin // g) ** 0.5
    b = torch.randn(cout) * 0.3
    want = F.conv2d(x, r16(w), b, groups=g)
    wp = ops.pack_conv_weights(D(w), 0, g)
    xb = nhwc(x).bfloat16()
    expect_reg = (cout // 16) in (1, 2, 4, 8)          # 48 output channels: three M-tiles, the general kernel
    # raw + bias
    assert (_uses_reg_kernel(xb, D(w), g, wp, out_shift=D(b)) == 2) == expect_reg
    got = ops.conv2d(xb, D(w), groups=g, out_shift=D(b), w_bf16=wp)
    torch.testing.assert_close(nchw(got), want, rtol=1e-2, atol=2e-2)
    # + statistics (of z + bias, what the BatchNorm that follows normalises)
    stats = (D(torch.zeros(cout)), D(torch.zeros(cout)))
    got2 = ops.conv2d(xb, D(w), groups=g, out_shift=D(b), stats=stats, w_bf16=wp)
    torch.testing.assert_close(nchw(got2), want, rtol=1e-2, atol=2e-2)
    npx = N * H * W
    torch.testing.assert_close(stats[0].cpu(), want.sum((0, 2, 3)), rtol=1e-3, atol=2e-3 * npx)
    torch.testing.assert_close(stats[1].cpu(), (want * want).sum((0, 2, 3)), rtol=2e-3, atol=2e-3 * npx)
    # the general MFMA kernel on the same operands: same products, same f32 accumulation up to order
    lib = _lib.get_lib()
    lib.set_option(2, 11)
    try:
        assert _uses_reg_kernel(xb, D(w), g, wp, out_shift=D(b)) == 1
        ref = ops.conv2d(xb, D(w), groups=g, out_shift=D(b), w_bf16=wp)
    finally:
        lib.set_option(2, -1)          # the default mask
    torch.testing.assert_close(got.float().cpu(), ref.float().cpu(), rtol=8e-3, atol=1e-3)


@pytest.mark.parametrize('cin,cout,g,nhw', CASES)
def test_conv1x1_dgrad_with_addend(be, cin, cout, g, nhw):
    """data gradient (mode-1 pack: the kernel's output channels = the forward's input channels), alone and with the
    partial gradient of another consumer added in the epilogue (bf16(z) + addend, as conv_mfma's EPI_RAW_ACC)"""
    from led_net_amd import ops
    N, H, W = nhw
    x = r16(torch.randn(N, cin, H, W)).requires_grad_(True)
    w = r16(torch.randn(cout, cin // g, 1, 1) / (cin // g) ** 0.5).requires_grad_(True)
    z = F.conv2d(x, w, groups=g)
    dz = r16(torch.randn_like(z))
    z.backward(dz)
    wp1 = ops.pack_conv_weights(D(w.detach()), 1, g)
    dzb = nhwc(dz).bfloat16()
    dx = ops.conv2d(dzb, D(w.detach()), groups=g, transposed=True, out_hw=(H, W), w_bf16=wp1)
    scale = float(x.grad.abs().max())
    torch.testing.assert_close(nchw(dx), x.grad, rtol=1e-2, atol=1e-2 * scale)
    prev = r16(torch.randn(N, cin, H, W))
    dx2 = ops.conv2d(dzb, D(w.detach()), groups=g, transposed=True, out_hw=(H, W), w_bf16=wp1,
                     res=nhwc(prev).bfloat16(), res_mode=ops.RES_ADD)
    torch.testing.assert_close(nchw(dx2), r16(x.grad) + prev, rtol=1e-2, atol=1.5e-2 * max(scale, 1.0))


@pytest.mark.parametrize('wgs', [1, 9, 0])
def test_conv1x1_many_iterations_few_workgroups(be, wgs):
    """LEDN_OPT_CONV_WORKGROUPS = 1 / 9: 2-3 / 18-27 workgroups, every wave walks ~30 / ~4 grid-stride iterations with
    a ragged last one (the prefetch of the next iteration, as at the 1024 x 1024 sizes); more than 16 workgroups:
    statistics through the per-workgroup rows + finish / deferred protocol, fewer: atomics"""
    from led_net_amd import ops, _lib
    lib = _lib.get_lib()
    lib.set_option(0, wgs)
    try:
        _many_iterations(ops)
    finally:
        lib.set_option(0, 0)


def _many_iterations(ops):
    cin = cout = 64
    N, H, W = 3, 61, 67
    x = r16(torch.randn(N, cin, H, W))
    w = torch.randn(cout, cin // 4, 1, 1) / 4.0
    want = F.conv2d(x, r16(w), groups=4)
    wp = ops.pack_conv_weights(D(w), 0, 4)
    stats = (D(torch.zeros(cout)), D(torch.zeros(cout)))
    got = ops.conv2d(nhwc(x).bfloat16(), D(w), groups=4, stats=stats, w_bf16=wp)
    torch.testing.assert_close(nchw(got), want, rtol=1e-2, atol=2e-2)
    torch.testing.assert_close(stats[0].cpu(), want.sum((0, 2, 3)), rtol=1e-3, atol=0.5)
    torch.testing.assert_close(stats[1].cpu(), (want * want).sum((0, 2, 3)), rtol=2e-3, atol=2.0)


@pytest.mark.parametrize('cin,cout,g,act', [(64, 64, 4, 'prelu'), (64, 64, 1, 'relu'), (32, 32, 1, 'none'), (64, 16, 4, 'prelu'),
                                           (16, 64, 1, 'relu')])
def test_conv1x1_input_prologue(be, cin, cout, g, act):
    """the producer's BatchNorm (+ ReLU / PReLU) folded into the convolution's input (training fusion level 2: SESP
    cat -> expansion, eesp.py:99-104): pre(x) = act(x * scale + shift) applied to the fragments in registers, with
    channel statistics of the output; ragged pixel count (the tail must stay zero AFTER the prologue)"""
    from led_net_amd import ops
    N, H, W = 2, 9, 21
    x = r16(torch.randn(N, cin, H, W))
    w = torch.randn(cout, cin // g, 1, 1) / (cin // g) ** 0.5
    sc, sh, sl = torch.rand(cin) + 0.5, torch.randn(cin) * 0.3, torch.rand(cin) * 0.4
    pre = x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    pre = {'prelu': lambda t: F.prelu(t, sl), 'relu': F.relu, 'none': lambda t: t}[act](pre)
    want = F.conv2d(r16(pre), r16(w), groups=g)
    wp = ops.pack_conv_weights(D(w), 0, g)
    kw = dict(groups=g, in_scale=D(sc), in_shift=D(sh), w_bf16=wp,
              in_act={'prelu': ops.ACT_PRELU, 'relu': ops.ACT_RELU, 'none': ops.ACT_NONE}[act],
              in_slope=D(sl) if act == 'prelu' else None)
    xb = nhwc(x).bfloat16()
    assert ops.conv2d_kernel_id(xb, D(w), **kw) == 2
    stats = (D(torch.zeros(cout)), D(torch.zeros(cout)))
    got = ops.conv2d(xb, D(w), stats=stats, **kw)
    torch.testing.assert_close(nchw(got), want, rtol=1e-2, atol=3e-2)
    torch.testing.assert_close(stats[0].cpu(), want.sum((0, 2, 3)), rtol=2e-3, atol=2e-3 * N * H * W)


@pytest.mark.parametrize('cin,cout,g,act,res', [(64, 64, 4, 'prelu', 'add'), (64, 64, 1, 'relu6', None), (32, 32, 1, 'none', 'gate'),
                                               (64, 16, 4, 'relu', None), (128, 128, 4, 'prelu', 'add'), (16, 64, 1, 'relu', 'add')])
def test_conv1x1_inference_epilogue(be, cin, cout, g, act, res):
    """folded BatchNorm (scale / shift), residual (add | gate) and activation in the epilogue: the inference form of the
    1x1 ConvModules (y = act(res_mode(z * scale + shift, res)), ledn.h) against torch and against the general kernel"""
    from led_net_amd import ops, _lib
    N, H, W = 2, 7, 23
    x = r16(torch.randn(N, cin, H, W))
    w = torch.randn(cout, cin // g, 1, 1) / (cin // g) ** 0.5
    sc, sh, sl = torch.rand(cout) + 0.5, torch.randn(cout) * 0.3, torch.rand(cout) * 0.4
    v = F.conv2d(x, r16(w), groups=g) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    r = r16(torch.randn_like(v))
    if res == 'add':
        v = v + r
    elif res == 'gate':
        v = v * r + r
    want = {'prelu': lambda t: F.prelu(t, sl), 'relu': F.relu, 'relu6': F.relu6, 'none': lambda t: t}[act](v)
    wp = ops.pack_conv_weights(D(w), 0, g)
    kw = dict(groups=g, out_scale=D(sc), out_shift=D(sh), w_bf16=wp,
              act={'prelu': ops.ACT_PRELU, 'relu': ops.ACT_RELU, 'relu6': ops.ACT_RELU6, 'none': ops.ACT_NONE}[act],
              slope=D(sl) if act == 'prelu' else None,
              res=nhwc(r).bfloat16() if res else None,
              res_mode={'add': ops.RES_ADD, 'gate': ops.RES_GATE, None: ops.RES_NONE}[res])
    xb = nhwc(x).bfloat16()
    assert ops.conv2d_kernel_id(xb, D(w), **kw) == 2
    got = ops.conv2d(xb, D(w), **kw)
    torch.testing.assert_close(nchw(got), want, rtol=1.5e-2, atol=3e-2)
    lib = _lib.get_lib()
    lib.set_option(2, 11)
    try:
        ref = ops.conv2d(xb, D(w), **kw)
    finally:
        lib.set_option(2, -1)
    torch.testing.assert_close(got.float().cpu(), ref.float().cpu(), rtol=8e-3, atol=2e-3)
