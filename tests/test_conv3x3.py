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


@pytest.mark.parametrize('shape,valid', [((2, 3, 64, 128), None), ((1, 3, 70, 150), None), ((2, 3, 72, 136), [(72, 136), (50, 101)]),
                                         ((1, 3, 33, 47), [(20, 31)])])
def test_stem_conv_register_direct(be, shape, valid):
    """ledn_stem_conv on stem_conv_reg_kernel (LEDN_OPT_STREAM_FAST bit 6: the 27 patch values of a pixel gathered into
    one K = 32 fragment, no LDS): the same comparisons as the LDS-window form in tests/test_conv_mfma.py -- against
    ledn_im2col_stem_planar + ledn_conv2d (inference epilogue; raw + statistics) and against torch's conv2d on the
    normalised, batch-padded input (ddrnet.py:123-130, data_preprocessor.py:98-151)."""
    from test_conv_mfma import test_fused_stem_conv_equals_im2col_plus_gemm as case
    case(be, shape, valid)


@pytest.mark.parametrize('shape,valid', [((2, 3, 64, 128), None), ((1, 3, 70, 150), None), ((2, 3, 72, 136), [(72, 136), (50, 101)]),
                                         ((1, 3, 33, 47), [(20, 31)])])
def test_stem_conv_wgrad_from_planar_batch(be, shape, valid):
    """ledn_stem_conv_wgrad (both MFMA operands gathered pixel-major, no patch matrix) against torch autograd of the
    stem convolution on the normalised, batch-padded input (bf16-rounded operands), accumulated INTO dw; and against the
    im2col'ed path it replaces (ledn_im2col_stem_planar + ledn_conv2d_wgrad of the 1x1 view)."""
    from led_net_amd import ops
    N, C, H, W = shape
    g = torch.Generator().manual_seed(H * W + 1)
    x = torch.randint(0, 256, shape, dtype=torch.uint8, generator=g)
    mean, std = torch.tensor([123.675, 116.28, 103.53]), torch.tensor([58.395, 57.12, 57.375])
    sc, sh = 1.0 / std, -mean / std
    xn = x.float()[:, [2, 1, 0]] * sc.view(1, 3, 1, 1) + sh.view(1, 3, 1, 1)
    if valid is not None:
        for i, (vh, vw) in enumerate(valid):
            xn[i, :, vh:, :] = 0.25
            xn[i, :, :, vw:] = 0.25
    xn = r16(xn)
    w = torch.zeros(32, 3, 3, 3, requires_grad=True)
    z = F.conv2d(xn, w, stride=2, padding=1)
    dz = r16(torch.randn(z.shape, generator=g))
    z.backward(dz)
    cmap = torch.tensor([2, 1, 0], dtype=torch.int32, device=_DEV[0])
    vt = torch.tensor(valid, dtype=torch.int32, device=_DEV[0]) if valid is not None else None
    dw = torch.full((32, 3, 3, 3), 0.5, device=_DEV[0])
    out = ops.stem_conv_wgrad(D(x), nhwc(dz).bfloat16(), dw, D(sc), D(sh), cmap, vt, 0.25)
    assert out is dw
    scale = float(w.grad.abs().max())
    torch.testing.assert_close(dw.cpu() - 0.5, w.grad, rtol=2e-3, atol=2e-3 * scale)
    patches = ops.im2col_stem_planar(D(x), D(sc), D(sh), cmap, vt, 0.25)
    dw1, _ = ops.conv2d_wgrad(patches, nhwc(dz).bfloat16(), (32, 32, 1, 1))
    ref = dw1.reshape(32, 32)[:, :27].reshape(32, 3, 3, 3).permute(0, 3, 1, 2)     # columns (kh 3 + kw) 3 + c -> OIHW
    torch.testing.assert_close(dw.cpu() - 0.5, ref.cpu(), rtol=1e-3, atol=1e-3 * scale)


@pytest.mark.parametrize('shape,act', [((2, 3, 64, 128), 'relu'), ((1, 3, 70, 150), 'relu'), ((2, 3, 66, 92), 'none')])
def test_stem_conv_wgrad_with_batchnorm_apply_prologue(be, shape, act):
    """ledn_stem_conv_wgrad_bn: the stem weight gradient with dz formed inside the kernel from z, the gradient dy of
    act(BatchNorm(z)) and the BatchNorm-backward sums (the apply half of ledn_bn_act_bwd as an operand prologue) -- against
    the two-launch form (ledn_bn_act_bwd_apply writes dz, ledn_stem_conv_wgrad reads it) on the same tensors."""
    from led_net_amd import ops, ops_train as T
    N, C, H, W = shape
    g = torch.Generator().manual_seed(H + W)
    x = D(torch.randint(0, 256, shape, dtype=torch.uint8, generator=g))
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    z = D(torch.randn(N, Ho, Wo, 32, generator=g).bfloat16())
    dy = D(torch.randn(N, Ho, Wo, 32, generator=g).bfloat16())
    gamma, beta = D(torch.rand(32, generator=g) + 0.5), D(torch.randn(32, generator=g) * 0.1)
    mean = z.float().mean((0, 1, 2))
    invstd = (z.float().var((0, 1, 2), unbiased=False) + 1e-5).rsqrt()
    kw = dict(scale=gamma * invstd, shift=beta - mean * gamma * invstd, mean=mean, invstd=invstd,
              act=ops.ACT_RELU if act == 'relu' else ops.ACT_NONE, count=N * Ho * Wo)
    sc, sh = D(torch.tensor([0.017, 0.0175, 0.0174])), D(torch.tensor([-2.1, -2.0, -1.8]))
    # two launches
    dz, _, dgamma, dbeta, _ = T.bn_act_bwd(z, dy, **kw)
    want = ops.stem_conv_wgrad(x, dz, torch.zeros(32, 3, 3, 3, device=_DEV[0]), sc, sh)
    # reduce half, then the weight gradient with the apply half as its prologue
    st = T.bn_act_bwd_reduce(z, dy, no_dz=True, **kw)
    got = ops.stem_conv_wgrad(x, dy, torch.zeros(32, 3, 3, 3, device=_DEV[0]), sc, sh, bn_desc=st.d)
    _, _, dgamma2, dbeta2, _ = T._bn_bwd_result(st)
    torch.testing.assert_close(dgamma2.cpu(), dgamma.cpu(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(dbeta2.cpu(), dbeta.cpu(), rtol=1e-5, atol=1e-5)
    # (the two-launch form rounds dz to bf16 in memory, the prologue rounds the same f32 value: identical operands)
    torch.testing.assert_close(got.cpu(), want.cpu(), rtol=1e-4, atol=1e-4 * float(want.abs().max()))


@pytest.mark.parametrize('cout,pro,nhw', [(2, True, (3, 21, 44)), (2, False, (1, 35, 64)), (1, True, (2, 9, 70)), (4, True, (1, 17, 33)),
                                          (2, True, (2, 40, 100))])
def test_narrow_output_wgrad_lds_ring(be, cout, pro, nhw):
    """conv3x3_wgrad_narrow_kernel (weight gradient of the two-class heads' 3x3 layers, led_head.py:44-51: x^T through a
    wave-private LDS ring + transposing reads, prologue = the BatchNorm + ReLU in front applied while writing the ring):
    against torch autograd on the bf16-rounded operands, accumulated INTO the caller's buffer, bias gradient included;
    and against conv_wgrad_mfma_kernel's narrow path (bit 6 off)."""
    from led_net_amd import ops, _lib
    N, H, W = nhw
    x = r16(torch.randn(N, 32, H, W))
    s_in, b_in = torch.rand(32) + 0.5, torch.randn(32) * 0.3
    w = (torch.randn(cout, 32, 3, 3) * 0.1).requires_grad_(True)
    pre = r16(F.relu(x * s_in.view(1, -1, 1, 1) + b_in.view(1, -1, 1, 1))) if pro else x
    z = F.conv2d(pre, w, padding=1)
    dz = r16(torch.randn_like(z))
    z.backward(dz)
    kw = dict(pad=1, bias=True)
    if pro:
        kw.update(in_scale=D(s_in), in_shift=D(b_in), in_act=ops.ACT_RELU)
    xb, dzb = nhwc(x).bfloat16(), nhwc(dz).bfloat16()
    assert ops.conv2d_wgrad(xb, dzb, tuple(w.shape), _query=True, **kw) == 2
    base = torch.randn_like(w.detach())
    sink = D(base.clone())
    dw, db = ops.conv2d_wgrad(xb, dzb, tuple(w.shape), dw_out=sink, **kw)
    assert dw is sink
    n = N * H * W
    scale = float(w.grad.abs().max())
    torch.testing.assert_close(dw.cpu() - base, w.grad, rtol=5e-3, atol=2e-5 * n ** 0.5 + 5e-3 * scale)
    torch.testing.assert_close(db.cpu(), dz.sum((0, 2, 3)), rtol=1e-3, atol=1e-3 * n ** 0.5)
    lib = _lib.get_lib()
    lib.set_option(2, 27)
    try:
        assert ops.conv2d_wgrad(xb, dzb, tuple(w.shape), _query=True, **kw) == 1
        ref, _ = ops.conv2d_wgrad(xb, dzb, tuple(w.shape), **kw)
    finally:
        lib.set_option(2, MASK)
    torch.testing.assert_close(dw.cpu() - base, ref.cpu(), rtol=5e-3, atol=5e-3 * scale)


def test_narrow_output_wgrad_unaligned_width_falls_back(be):
    """two output channels and a width that is not a multiple of 4 (the dz fragment's 16-byte pieces would straddle rows):
    the general kernel"""
    from led_net_amd import ops
    x = torch.zeros(1, 9, 45, 32, dtype=torch.bfloat16, device=_DEV[0])
    dz = torch.zeros(1, 9, 45, 2, dtype=torch.bfloat16, device=_DEV[0])
    assert ops.conv2d_wgrad(x, dz, (2, 32, 3, 3), pad=1, _query=True) == 1


@pytest.mark.parametrize('cin,cout,g', [(64, 64, 1), (64, 64, 4), (64, 16, 4), (16, 64, 1), (128, 128, 4), (128, 32, 1), (32, 128, 1),
                                        (16, 16, 1), (128, 16, 1)])
def test_conv1x1_wgrad_wave_autonomous(be, cin, cout, g):
    """conv1x1_wgrad_reg_kernel (weight gradient of the 1x1 stride-1 layers: both operands through wave-private LDS tiles
    and transposing reads, every (ci, co) pair from ONE read of x and dz; partial tiles in conv_wgrad_mfma_kernel's layout,
    summed by its kernels): against torch autograd on the bf16-rounded operands, accumulated INTO the caller's buffer, and
    against conv_wgrad_mfma_kernel (bit 6 off).  32 773 pixels: ragged last 32-pixel chunk, above the 32 768-pixel gate.
    Reference layers: the 1x1 ConvModules / CBR of nn_layers/espnet_utils.py:22-36, eesp.py:39,70, model_utils.py:360-400."""
    from led_net_amd import ops, _lib
    N, H, W = 1, 113, 290 + 0
    x = r16(torch.randn(N, cin, H, W))
    w = (torch.randn(cout, cin // g, 1, 1) * 0.1).requires_grad_(True)
    z = F.conv2d(x, w, groups=g)
    dz = r16(torch.randn_like(z))
    z.backward(dz)
    xb, dzb = nhwc(x).bfloat16(), nhwc(dz).bfloat16()
    base = torch.randn_like(w.detach())
    sink = D(base.clone())
    assert ops.conv2d_wgrad(xb, dzb, tuple(w.shape), groups=g, _query=True) == 3
    dw, _ = ops.conv2d_wgrad(xb, dzb, tuple(w.shape), groups=g, dw_out=sink)
    assert dw is sink
    n = N * H * W
    scale = float(w.grad.abs().max())
    torch.testing.assert_close(dw.cpu() - base, w.grad, rtol=2e-3, atol=2e-5 * n ** 0.5 + 2e-3 * scale)
    lib = _lib.get_lib()
    lib.set_option(2, 27)
    try:
        assert ops.conv2d_wgrad(xb, dzb, tuple(w.shape), groups=g, _query=True) == 1
        ref, _ = ops.conv2d_wgrad(xb, dzb, tuple(w.shape), groups=g)
    finally:
        lib.set_option(2, MASK)
    torch.testing.assert_close(dw.cpu() - base, ref.cpu(), rtol=2e-3, atol=2e-3 * scale)


@pytest.mark.parametrize('cout,nhw', [(64, (2, 9, 33)), (64, (1, 17, 16)), (32, (1, 5, 40)), (128, (1, 3, 17)), (64, (1, 35, 50))])
def test_conv3x3_ring64_forward_and_dgrad(be, cout, nhw):
    """conv3x3_ring64_kernel (64 input channels: input rows through a wave-private LDS ring, one ds_read_b128 per tap and
    chunk; opt-in, LEDN_OPT_STREAM_FAST bit 8): forward raw + bias, + statistics, inference epilogue, and the data
    gradient of a 64 -> 64 layer with a fan-in addend -- against torch and against conv_mfma_kernel."""
    from led_net_amd import ops, _lib
    lib = _lib.get_lib()
    lib.set_option(2, MASK + 256)
    N, H, W = nhw
    x = r16(torch.randn(N, 64, H, W))
    w = torch.randn(cout, 64, 3, 3) / 24.0
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
    sc, sh = torch.rand(cout) + 0.5, torch.randn(cout) * 0.2
    res = r16(torch.randn(N, cout, H, W))
    want3 = torch.relu(F.conv2d(x, r16(w), None, padding=1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) + res)
    got3 = ops.conv2d(xb, D(w), pad=1, out_scale=D(sc), out_shift=D(sh), act=ops.ACT_RELU, res=nhwc(res).bfloat16(),
                      res_mode=ops.RES_ADD, w_bf16=wp)
    torch.testing.assert_close(nchw(got3), want3, rtol=1e-2, atol=3e-2)
    lib.set_option(2, 27)
    try:
        ref = ops.conv2d(xb, D(w), pad=1, out_shift=D(b), w_bf16=wp)
    finally:
        lib.set_option(2, MASK + 256)
    torch.testing.assert_close(got.float().cpu(), ref.float().cpu(), rtol=8e-3, atol=2e-3)
    if cout == 64:      # data gradient of a 64 -> 64 layer (+ addend)
        xg = r16(torch.randn(N, 64, H, W)).requires_grad_(True)
        wg = r16(torch.randn(64, 64, 3, 3) / 24.0).requires_grad_(True)
        z = F.conv2d(xg, wg, padding=1)
        dz = r16(torch.randn_like(z))
        z.backward(dz)
        wp1 = ops.pack_conv_weights(D(wg.detach()), 1, 1)
        prev = r16(torch.randn(N, 64, H, W))
        dx2 = ops.conv2d(nhwc(dz).bfloat16(), D(wg.detach()), pad=1, transposed=True, out_hw=(H, W), w_bf16=wp1,
                         res=nhwc(prev).bfloat16(), res_mode=ops.RES_ADD)
        scale = float(xg.grad.abs().max())
        torch.testing.assert_close(nchw(dx2), r16(xg.grad) + prev, rtol=1e-2, atol=1.5e-2 * max(scale, 1.0))


def test_stem_conv_misaligned_batch_slice(be):
    """a slice of a uint8 batch whose images have an odd byte count starts on an odd address: the wrapper realigns it (the
    kernel reads 4-byte aligned windows and rejects a misaligned base loudly).  RAW-domain inputs (0..255) and no valid
    extents: a wrong byte at the image / tensor ends (the clamped windows: images of 4653 bytes do not start on 4-byte
    boundaries) is an error of tens -- with normalised inputs and padded extents such bytes hid inside the tolerance / the
    padded area."""
    from led_net_amd import ops
    g = torch.Generator().manual_seed(77)
    full = torch.randint(0, 256, (3, 3, 33, 47), dtype=torch.uint8, generator=g).to(_DEV[0])      # 4653 bytes per image
    x = full[1:]
    assert x.data_ptr() % 4 != 0 and x.is_contiguous()
    w = (0.2 * torch.randn(32, 3, 3, 3, generator=g)).to(_DEV[0])
    wp = ops.pack_conv_weights(ops.stem_weight_as_1x1(w), 0)
    got = ops.stem_conv(x, wp)
    want = ops.stem_conv(x.clone(), wp)
    assert torch.equal(got.cpu(), want.cpu())
    ref = F.conv2d(r16(x.float().cpu()), r16(w.cpu()), stride=2, padding=1).permute(0, 2, 3, 1)
    torch.testing.assert_close(got.float().cpu(), ref, rtol=1e-2, atol=2.0)
    dz = torch.randn(2, 17, 24, 32, generator=g).to(_DEV[0], torch.bfloat16)
    dw1 = ops.stem_conv_wgrad(x, dz, torch.zeros(32, 3, 3, 3, device=_DEV[0]))
    dw2 = ops.stem_conv_wgrad(x.clone(), dz, torch.zeros(32, 3, 3, 3, device=_DEV[0]))
    torch.testing.assert_close(dw1.cpu(), dw2.cpu(), rtol=1e-5, atol=1e-2)
    wr = w.detach().cpu().clone().requires_grad_(True)
    F.conv2d(x.float().cpu(), wr, stride=2, padding=1).backward(dz.float().cpu().permute(0, 3, 1, 2))
    torch.testing.assert_close(dw1.cpu(), wr.grad, rtol=1e-4, atol=1e-4 * float(wr.grad.abs().max()))


@pytest.mark.parametrize('shape', [(3, 3, 9, 11), (1, 3, 5, 7), (2, 3, 33, 47), (2, 3, 31, 64)])
def test_stem_kernels_raw_domain_every_border(be, shape):
    """both stem kernels on raw uint8 values without normalisation or padded extents, image sizes whose byte counts are
    not multiples of 4 (window alignment / clamping at every image and tensor end), against torch"""
    from led_net_amd import ops
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randint(0, 256, shape, dtype=torch.uint8, generator=g)
    w = r16(0.2 * torch.randn(32, 3, 3, 3, generator=g))
    wp = ops.pack_conv_weights(ops.stem_weight_as_1x1(D(w)), 0)
    got = ops.stem_conv(D(x), wp)
    wr = w.clone().requires_grad_(True)
    z = F.conv2d(x.float(), wr, stride=2, padding=1)
    torch.testing.assert_close(got.float().cpu(), z.detach().permute(0, 2, 3, 1), rtol=1e-2, atol=2.0)
    dz = r16(torch.randn(z.shape, generator=g))
    z.backward(dz)
    dw = ops.stem_conv_wgrad(D(x), nhwc(dz).bfloat16(), torch.zeros(32, 3, 3, 3, device=_DEV[0]))
    torch.testing.assert_close(dw.cpu(), wr.grad, rtol=1e-4, atol=1e-4 * float(wr.grad.abs().max()))
