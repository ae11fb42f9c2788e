"""MFMA implicit-GEMM convolution engine (forward, data gradient, weight gradient)
against torch on bf16-rounded operands, plus agreement with the direct kernels.
Runs on the emulator (MFMA + ds_read_tr emulated lane-exactly) and on the GPU."""
import pytest
import torch
import torch.nn.functional as F

torch.manual_seed(2)
_DEV = [torch.device('cpu')]


@pytest.fixture(autouse=True)
def _track_device(request):
    """(LEDN_OPT_STREAM_FAST without bit 6: the 3x3 / 32-channel shapes of this file stay on conv_mfma_kernel, which
    still runs them whenever the launch carries an input prologue; tests/test_conv3x3.py covers the register kernel)"""
    _DEV[0] = request.getfixturevalue('be').dev if 'be' in request.fixturenames else torch.device('cpu')
    if 'be' in request.fixturenames:
        from led_net_amd import _lib
        _lib.get_lib().set_option(2, 27)
    yield
    if 'be' in request.fixturenames:
        from led_net_amd import _lib
        _lib.get_lib().set_option(2, -1)


def D(t):
    return t.to(_DEV[0])


def nhwc(t):
    return D(t.detach().permute(0, 2, 3, 1).contiguous())


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous().cpu().float()


def r16(t):
    return t.bfloat16().float()


CASES = [  # cin, cout, k, stride, hw
    (32, 32, 3, 1, (20, 37)), (32, 32, 3, 2, (21, 70)), (32, 64, 3, 2, (18, 40)), (64, 64, 3, 1, (9, 33)),
    (64, 128, 3, 2, (16, 34)), (128, 64, 3, 1, (7, 40)), (128, 256, 3, 2, (10, 36)),
    (128, 384, 1, 1, (8, 24)), (512, 128, 1, 1, (5, 6)), (32, 64, 1, 2, (13, 35)), (256, 64, 1, 1, (6, 33)),
]


@pytest.mark.parametrize('cin,cout,k,stride,hw', CASES)
def test_mfma_conv_forward(be, cin, cout, k, stride, hw):
    from led_net_amd import ops
    x = r16(torch.randn(2, cin, *hw))
    w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
    s_in, b_in = torch.rand(cin) + 0.5, torch.randn(cin) * 0.1
    s_o, b_o, sl = torch.rand(cout) + 0.5, torch.randn(cout) * 0.1, torch.rand(cout) * 0.3
    pad = k // 2
    xin = r16(F.relu(x * s_in.view(1, -1, 1, 1) + b_in.view(1, -1, 1, 1)))
    z = F.conv2d(xin, r16(w), stride=stride, padding=pad)
    v = z * s_o.view(1, -1, 1, 1) + b_o.view(1, -1, 1, 1)
    res = r16(torch.randn_like(v))
    want = F.prelu(v + res, sl)
    stats = (D(torch.zeros(cout)), D(torch.zeros(cout)))
    wp = ops.pack_conv_weights(D(w), 0)
    got = ops.conv2d(nhwc(x).bfloat16(), D(w), stride=stride, pad=pad, in_scale=D(s_in), in_shift=D(b_in),
                     in_act=ops.ACT_RELU, out_scale=D(s_o), out_shift=D(b_o), act=ops.ACT_PRELU, slope=D(sl),
                     res=nhwc(res).bfloat16(), res_mode=ops.RES_ADD, stats=stats, w_bf16=wp)
    assert got.dtype == torch.bfloat16
    torch.testing.assert_close(nchw(got), want, rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(stats[0].cpu(), v.sum((0, 2, 3)), rtol=1e-2, atol=0.05 * v.shape[2] * v.shape[3] ** 0.5)
    # the direct (VALU) kernel on the same bf16 operands agrees to bf16 output rounding
    ref = ops.conv2d(nhwc(x).bfloat16(), D(r16(w)), stride=stride, pad=pad, in_scale=D(s_in), in_shift=D(b_in),
                     in_act=ops.ACT_RELU, out_scale=D(s_o), out_shift=D(b_o), act=ops.ACT_PRELU, slope=D(sl),
                     res=nhwc(res).bfloat16(), res_mode=ops.RES_ADD)
    torch.testing.assert_close(nchw(got), nchw(ref), rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize('cin,cout,k,stride,hw', CASES)
def test_mfma_conv_dgrad_wgrad(be, cin, cout, k, stride, hw):
    from led_net_amd import ops
    pad = k // 2
    x = r16(torch.randn(2, cin, *hw)).requires_grad_(True)
    w = r16(torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5).requires_grad_(True)
    z = F.conv2d(x, w, stride=stride, padding=pad)
    dz = r16(torch.randn_like(z))
    z.backward(dz)
    wp1 = ops.pack_conv_weights(D(w.detach()), 1)
    dx = ops.conv2d(nhwc(dz).bfloat16(), D(w.detach()), stride=stride, pad=pad, transposed=True, out_hw=hw,
                    w_bf16=wp1)
    torch.testing.assert_close(nchw(dx), x.grad, rtol=2e-2, atol=2e-2 * float(x.grad.abs().max()))
    dw, db = ops.conv2d_wgrad(nhwc(x).bfloat16(), nhwc(dz).bfloat16(), tuple(w.shape), stride=stride, pad=pad,
                              bias=True)
    torch.testing.assert_close(dw.cpu(), w.grad, rtol=1e-2, atol=1e-2 * float(w.grad.abs().max()))
    torch.testing.assert_close(db.cpu(), dz.sum((0, 2, 3)), rtol=1e-3, atol=1e-2)


@pytest.fixture
def few_workgroups(be):
    """3 persistent conv workgroups / 2 wgrad workgroups: every workgroup walks several tiles and
    K-chunks (the software-pipelined loop), as at the 1024x1024 sizes."""
    from led_net_amd import _lib
    lib = _lib.get_lib()
    lib.set_option(0, 3)
    lib.set_option(1, 2)
    yield
    lib.set_option(0, 0)
    lib.set_option(1, 0)


@pytest.mark.parametrize('cin,cout,k,stride,hw', [(32, 32, 3, 1, (37, 70)), (64, 64, 3, 1, (35, 40)),
                                                  (32, 64, 3, 2, (40, 67)), (128, 256, 1, 1, (20, 66)),
                                                  (64, 32, 1, 2, (33, 130)), (32, 2, 3, 1, (36, 40))])
def test_mfma_persistent_multi_tile(be, few_workgroups, cin, cout, k, stride, hw):
    test_mfma_conv_forward(be, cin, cout, k, stride, hw)
    if not (k == 1 and stride == 2):
        test_mfma_conv_dgrad_wgrad(be, cin, cout, k, stride, hw)


@pytest.mark.parametrize('cin,cout,hw', [(64, 16, (9, 40)), (64, 64, (12, 33)), (128, 32, (8, 35)), (256, 256, (5, 34))])
def test_mfma_grouped_1x1(be, cin, cout, hw):
    """SESP's grouped (g=4) 1x1 convs on the MFMA path: densified weight pack, Cout = 16 tail."""
    from led_net_amd import ops
    g = 4
    x = r16(torch.randn(2, cin, *hw)).requires_grad_(True)
    w = r16(torch.randn(cout, cin // g, 1, 1) / (cin // g) ** 0.5).requires_grad_(True)
    z = F.conv2d(x, w, groups=g)
    dz = r16(torch.randn_like(z))
    z.backward(dz)
    stats = (D(torch.zeros(cout)), D(torch.zeros(cout)))
    got = ops.conv2d(nhwc(x).bfloat16(), D(w.detach()), groups=g, stats=stats,
                     w_bf16=ops.pack_conv_weights(D(w.detach()), 0, g))
    torch.testing.assert_close(nchw(got), z.detach(), rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(stats[0].cpu(), z.detach().sum((0, 2, 3)), rtol=1e-2, atol=0.5)
    if cout % 32 == 0:
        dx = ops.conv2d(nhwc(dz).bfloat16(), D(w.detach()), groups=g, transposed=True, out_hw=hw,
                        w_bf16=ops.pack_conv_weights(D(w.detach()), 1, g))
        torch.testing.assert_close(nchw(dx), x.grad, rtol=2e-2, atol=2e-2 * float(x.grad.abs().max()))
    dw, _ = ops.conv2d_wgrad(nhwc(x).bfloat16(), nhwc(dz).bfloat16(), tuple(w.shape), groups=g)
    torch.testing.assert_close(dw.cpu(), w.grad, rtol=1e-2, atol=1e-2 * float(w.grad.abs().max()))


@pytest.mark.parametrize('cin,cout,k,hw', [(16, 64, 1, (9, 40)), (64, 16, 1, (12, 33)), (32, 2, 3, (20, 37)),
                                           (64, 2, 1, (10, 34)), (64, 1, 3, (9, 33)), (48, 32, 3, (8, 35))])
def test_mfma_channel_tails(be, cin, cout, k, hw):
    """Cin = 16 / 48 (half-filled K chunk) and Cout = 1, 2 (segmentation heads, SEAM) on the MFMA path."""
    from led_net_amd import ops
    pad = k // 2
    x = r16(torch.randn(2, cin, *hw)).requires_grad_(True)
    w = r16(torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5).requires_grad_(True)
    z = F.conv2d(x, w, padding=pad)
    dz = r16(torch.randn_like(z))
    z.backward(dz)
    assert ops.mfma_weight_ok(w)
    got = ops.conv2d(nhwc(x).bfloat16(), D(w.detach()), pad=pad, w_bf16=ops.pack_conv_weights(D(w.detach()), 0))
    torch.testing.assert_close(nchw(got), z.detach(), rtol=2e-2, atol=2e-2)
    dw, _ = ops.conv2d_wgrad(nhwc(x).bfloat16(), nhwc(dz).bfloat16(), tuple(w.shape), pad=pad)
    torch.testing.assert_close(dw.cpu(), w.grad, rtol=1e-2, atol=1e-2 * float(w.grad.abs().max()))


@pytest.mark.parametrize('cin,cout,k', [(64, 2, 1), (32, 6, 3), (32, 32, 3), (64, 16, 1)])
@pytest.mark.parametrize('epi', ['bias', 'scale', 'relu6', 'gate'])
def test_mfma_epilogue_variants(be, cin, cout, k, epi):
    """bias-only (cls_seg), scale-only, ReLU6 and gated-residual epilogues incl. narrow heads (Cout = 2, 6)."""
    from led_net_amd import ops
    pad = k // 2
    x = r16(torch.randn(2, cin, 11, 37))
    w = r16(torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5)
    z = F.conv2d(x, w, padding=pad)
    sc, sh = torch.rand(cout) + 0.5, torch.randn(cout)
    kw = {}
    if epi == 'bias':
        want, kw = z + sh.view(1, -1, 1, 1), dict(out_shift=D(sh))
    elif epi == 'scale':
        want, kw = z * sc.view(1, -1, 1, 1), dict(out_scale=D(sc))
    elif epi == 'relu6':
        want = F.relu6(z * 8 + sh.view(1, -1, 1, 1))
        kw = dict(out_scale=D(torch.full((cout,), 8.0)), out_shift=D(sh), act=ops.ACT_RELU6)
    else:
        res = r16(torch.randn_like(z))
        want, kw = z * res + res, dict(res=nhwc(res).bfloat16(), res_mode=ops.RES_GATE)
    got = ops.conv2d(nhwc(x).bfloat16(), D(w), pad=pad, w_bf16=ops.pack_conv_weights(D(w), 0), **kw)
    torch.testing.assert_close(nchw(got), want, rtol=2e-2, atol=3e-2)


@pytest.mark.parametrize('cin,cout,k', [(32, 2, 3), (64, 2, 1), (32, 6, 3)])
def test_mfma_narrow_head_f32_output(be, cin, cout, k):
    """LEDHead's head_x1/head_x2 (BN-ReLU-conv3x3(32->2)-BN-ReLU, led_head.py:84-99) keep f32
    logits: the MFMA kernel's narrow-head epilogue stores float directly (prologue + epilogue)."""
    from led_net_amd import ops
    pad = k // 2
    x = r16(torch.randn(2, cin, 13, 45))
    w = r16(torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5)
    s_in, b_in = torch.rand(cin) + 0.5, torch.randn(cin) * 0.1
    sc, sh = torch.rand(cout) + 0.5, torch.randn(cout) * 0.2
    xin = r16(F.relu(x * s_in.view(1, -1, 1, 1) + b_in.view(1, -1, 1, 1)))
    want = F.relu(F.conv2d(xin, w, padding=pad) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    d = ops._lib.ConvDesc()
    got = ops.conv2d(nhwc(x).bfloat16(), D(w), pad=pad, in_scale=D(s_in), in_shift=D(b_in), in_act=ops.ACT_RELU,
                     out_scale=D(sc), out_shift=D(sh), act=ops.ACT_RELU, out_dtype=torch.float32,
                     w_bf16=ops.pack_conv_weights(D(w), 0))
    assert got.dtype == torch.float32
    # f32 store: no output rounding; one prologue value on a bf16 rounding boundary (fma in the kernel, two roundings in
    # the reference) moves an output by ~|w| 2^-8 |x|: seen 2.6e-3 once in 2340 values
    torch.testing.assert_close(nchw(got), want, rtol=1e-3, atol=5e-3)


@pytest.mark.parametrize('dt,hw', [(torch.uint8, (37, 50)), (torch.float32, (16, 21))])
def test_im2col_stem_planar_equals_two_step(be, dt, hw):
    """ledn_im2col_stem_planar (planar batch -> normalised bf16 patches in one kernel) is bit-identical
    to ledn_nchw_to_nhwc followed by ledn_im2col_stem (SegDataPreProcessor: BGR->RGB map, mean/std)."""
    from led_net_amd import ops
    g = torch.Generator().manual_seed(4)
    x = torch.randint(0, 256, (2, 3, *hw), generator=g).to(dt)
    mean, std = torch.tensor([123.675, 116.28, 103.53]), torch.tensor([58.395, 57.12, 57.375])
    s, b = D(1.0 / std), D(-mean / std)
    mp = D(torch.tensor([2, 1, 0], dtype=torch.int32))
    want = ops.im2col_stem(ops.nchw_to_nhwc(D(x), torch.bfloat16, s, b, mp))
    got = ops.im2col_stem_planar(D(x), s, b, mp)
    assert got.shape == want.shape and got.dtype == torch.bfloat16
    assert torch.equal(got.cpu().view(torch.int16), want.cpu().view(torch.int16))


def test_deferred_conv_statistics_finalize(be):
    """ledn_conv2d_deferred_stats + ledn_bn_finalize_rows (statistic rows summed inside the finalize
    launch) against ledn_conv2d + ledn_bn_finalize on the same operands."""
    from led_net_amd import ops
    x = r16(torch.randn(2, 32, 40, 70))
    w = torch.randn(64, 32, 3, 3) / 17.0
    gamma, beta = torch.rand(64) + 0.5, torch.randn(64) * 0.1
    wp = ops.pack_conv_weights(D(w), 0)
    outs = []
    for defer in (False, True):
        stats = (D(torch.zeros(64)), D(torch.zeros(64)))
        rm, rv = D(torch.zeros(64)), D(torch.ones(64))
        z = ops.conv2d(nhwc(x).bfloat16(), D(w), pad=1, stats=stats, w_bf16=wp, defer_stats=defer)
        if defer:
            assert ops.PendingRows.entry is not None, 'this shape must take the partial-row path'
        sc, sh, mean, invstd = ops.bn_finalize(stats, 2 * 40 * 70, D(gamma), D(beta), rm, rv)
        assert ops.PendingRows.entry is None
        outs.append([t.cpu() for t in (z.float(), sc, sh, mean, invstd, rm, rv, stats[0], stats[1])])
    for a, b in zip(*outs):
        torch.testing.assert_close(a, b, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize('cin,cout,k,hw,dz_f32', [(32, 2, 3, (21, 45), False), (64, 2, 1, (16, 23), True),
                                                 (32, 1, 3, (9, 70), False), (128, 2, 3, (12, 12), False)])
def test_two_class_head_wgrad_kernel(be, cin, cout, k, hw, dz_f32):
    """Weight gradients of LEDHead's two-class layers (conv_wgrad_cout2_kernel for the 1x1 classifiers, the MFMA
    narrow path for the 3x3 heads): x streamed once with the producer's BatchNorm + ReLU applied while staging, against torch autograd on the bf16-rounded operands; bias
    gradient and accumulation into a caller buffer included."""
    from led_net_amd import ops
    pad = k // 2
    x = r16(torch.randn(3, cin, *hw))
    s_in, b_in = torch.rand(cin) + 0.5, torch.randn(cin) * 0.1
    w = (torch.randn(cout, cin, k, k) * 0.1).requires_grad_(True)
    pre = F.relu(x * s_in.view(1, -1, 1, 1) + b_in.view(1, -1, 1, 1))
    z = F.conv2d(pre, w, padding=pad)
    dz = torch.randn_like(z)
    dz = dz if dz_f32 else r16(dz)
    z.backward(dz)
    base = torch.randn_like(w.detach())
    sink = D(base.clone())
    dzd = nhwc(dz) if dz_f32 else nhwc(dz).bfloat16()
    dw, db = ops.conv2d_wgrad(nhwc(x).bfloat16(), dzd, tuple(w.shape), pad=pad, in_scale=D(s_in), in_shift=D(b_in),
                              in_act=ops.ACT_RELU, bias=True, dw_out=sink)
    n = 3 * hw[0] * hw[1]
    # (the MFMA path rounds act(BN(x)) to bf16 while staging: 2e-2 of the gradient scale)
    torch.testing.assert_close(dw.cpu() - base, w.grad, rtol=2e-2, atol=2e-4 * n ** 0.5 + 2e-2 * float(w.grad.abs().max()))
    torch.testing.assert_close(db.cpu(), dz.sum((0, 2, 3)), rtol=1e-3, atol=1e-3 * n ** 0.5)


@pytest.mark.parametrize('cin,cout,k,stride,hw', [(32, 32, 3, 1, (20, 37)), (64, 64, 3, 1, (9, 33)), (32, 32, 3, 2, (21, 70)),
                                                  (64, 64, 1, 1, (12, 40)), (128, 64, 3, 1, (7, 40))])
def test_mfma_dgrad_accumulates_partial_gradient(be, cin, cout, k, stride, hw):
    """gradient fan-in folded into the data-gradient kernel: conv2d(transposed, res=partial, RES_ADD) with nothing
    else in the epilogue runs the store-stage accumulate flavour (EPI_RAW_ACC): bf16(dgrad) + partial, i.e. exactly
    the separate elementwise add it replaces (one bf16 rounding of the sum)."""
    from led_net_amd import ops
    x = r16(torch.randn(2, cin, *hw))
    w = torch.randn(cout, cin, k, k) / (cin * k * k) ** 0.5
    pad = k // 2
    Ho, Wo = ops.conv_out_size(hw[0], k, stride, pad, 1), ops.conv_out_size(hw[1], k, stride, pad, 1)
    dz = r16(torch.randn(2, cout, Ho, Wo))
    partial = r16(torch.randn(2, cin, *hw))
    wp = ops.pack_conv_weights(D(w), 1)
    plain = ops.conv2d(nhwc(dz).bfloat16(), D(w), stride=stride, pad=pad, transposed=True, out_hw=hw, w_bf16=wp)
    fused = ops.conv2d(nhwc(dz).bfloat16(), D(w), stride=stride, pad=pad, transposed=True, out_hw=hw, w_bf16=wp,
                       res=nhwc(partial).bfloat16(), res_mode=ops.RES_ADD)
    want = (plain.float() + nhwc(partial)).bfloat16()
    assert torch.equal(fused.cpu(), want.cpu())
    xg = x.clone().requires_grad_(True)
    F.conv2d(xg, r16(w), stride=stride, padding=pad).backward(dz)
    torch.testing.assert_close(nchw(fused), xg.grad + partial, rtol=2e-2, atol=3e-2)


@pytest.mark.parametrize('shape,valid', [((2, 3, 64, 128), None), ((1, 3, 70, 150), None), ((2, 3, 72, 136), [(72, 136), (50, 101)])])
def test_fused_stem_conv_equals_im2col_plus_gemm(be, shape, valid):
    """ledn_stem_conv (normalise + pad + im2col + K=32 MFMA GEMM in one kernel) against the two-kernel path it replaces
    (ledn_im2col_stem_planar + ledn_conv2d): inference epilogue (folded BatchNorm + ReLU) and training flavour (raw z +
    per-channel statistics); ragged tiles and per-image valid extents."""
    from led_net_amd import ops
    N, C, H, W = shape
    g = torch.Generator().manual_seed(H * W)
    x = torch.randint(0, 256, shape, dtype=torch.uint8, generator=g).to(be.dev)
    w = (0.2 * torch.randn(32, 3, 3, 3, generator=g)).to(be.dev)
    mean, std = torch.tensor([123.675, 116.28, 103.53]), torch.tensor([58.395, 57.12, 57.375])
    sc, sh = (1.0 / std).to(be.dev), (-mean / std).to(be.dev)
    cmap = torch.tensor([2, 1, 0], dtype=torch.int32, device=be.dev)
    vt = torch.tensor(valid, dtype=torch.int32, device=be.dev) if valid is not None else None
    w1 = ops.stem_weight_as_1x1(w)
    wp = ops.pack_conv_weights(w1, 0)
    patches = ops.im2col_stem_planar(x, sc, sh, cmap, vt, 0.25)
    osc, osh = (0.5 + torch.rand(32, generator=g)).to(be.dev), (0.1 * torch.randn(32, generator=g)).to(be.dev)
    want = ops.conv2d(patches, w1, out_scale=osc, out_shift=osh, act=ops.ACT_RELU, w_bf16=wp)
    got = ops.stem_conv(x, wp, sc, sh, cmap, vt, 0.25, out_scale=osc, out_shift=osh, act=ops.ACT_RELU)
    assert got.shape == want.shape and got.dtype == torch.bfloat16
    torch.testing.assert_close(got.float().cpu(), want.float().cpu(), rtol=1e-2, atol=1e-2)
    st_w = (torch.zeros(32, device=be.dev), torch.zeros(32, device=be.dev))
    st_g = (torch.zeros(32, device=be.dev), torch.zeros(32, device=be.dev))
    zw = ops.conv2d(patches, w1, stats=st_w, w_bf16=wp)
    zg = ops.stem_conv(x, wp, sc, sh, cmap, vt, 0.25, stats=st_g)
    torch.testing.assert_close(zg.float().cpu(), zw.float().cpu(), rtol=1e-2, atol=1e-2)
    torch.testing.assert_close(st_g[0].cpu(), st_w[0].cpu(), rtol=1e-3, atol=1e-2 * zw.float().abs().sum().item() / 32 / 100)
    torch.testing.assert_close(st_g[1].cpu(), st_w[1].cpu(), rtol=1e-3, atol=1e-1)
    # and against plain torch: conv2d(normalised input, zero padding)
    xn = x.float().cpu()[:, [2, 1, 0]] * sc.cpu().view(1, 3, 1, 1) + sh.cpu().view(1, 3, 1, 1)
    if valid is not None:
        for i, (vh, vw) in enumerate(valid):
            xn[i, :, vh:, :] = 0.25
            xn[i, :, :, vw:] = 0.25
    ref = torch.nn.functional.conv2d(xn, w.cpu(), stride=2, padding=1).permute(0, 2, 3, 1)
    torch.testing.assert_close(zg.float().cpu(), ref, rtol=3e-2, atol=3e-2)


def test_deferred_weight_gradients_one_summing_launch(be):
    """ops.WgradDefer: several convolutions write their partial tiles, ONE ledn_conv2d_wgrad_finish_multi launch sums all of
    them into their sinks (round 4: a workgroup per 32 x 32 tile, 16-byte row walks) == the immediate path, including a
    grouped 1x1, a 1x1 on the wave-autonomous kernel and a sink that already holds a value (accumulation)."""
    from led_net_amd import ops
    shapes = [(32, 32, 3, 1, 1, (2, 24, 40)), (64, 64, 3, 1, 1, (2, 16, 33)), (64, 64, 1, 1, 4, (2, 32, 40)),
              (128, 64, 3, 1, 1, (1, 12, 40)), (64, 128, 1, 1, 1, (2, 32, 40))]
    items = []
    for cin, cout, k, stride, groups, (N, H, W) in shapes:
        x = r16(torch.randn(N, cin, H, W))
        dz = r16(torch.randn(N, cout, H, W))
        wshape = (cout, cin // groups, k, k)
        base = torch.randn(wshape)
        items.append((nhwc(x).bfloat16(), nhwc(dz).bfloat16(), wshape, dict(stride=stride, pad=k // 2, groups=groups), base))
    want = []
    for x, dz, wshape, kw, base in items:
        sink = D(base.clone())
        ops.conv2d_wgrad(x, dz, wshape, dw_out=sink, **kw)
        want.append(sink.cpu())
    ops.WgradDefer.reset()
    ops.WgradDefer.active = True
    try:
        sinks = []
        for x, dz, wshape, kw, base in items:
            sink = D(base.clone())
            sinks.append(sink)
            ops.conv2d_wgrad(x, dz, wshape, dw_out=sink, **kw)
        assert len(ops.WgradDefer.pending) >= 4, 'the MFMA weight gradients were not deferred'
        ops.WgradDefer.finish()
        for got, ref in zip(sinks, want):
            torch.testing.assert_close(got.cpu(), ref, rtol=1e-4, atol=1e-3)
    finally:
        ops.WgradDefer.reset()
