"""Kernel sources run under the CPU emulation build (tests/emu, csrc/emu.h) and
are compared with plain torch ops: indexing / layout / barrier logic checks that
need no GPU.  The same comparisons run on the real device in test_gpu_*.py."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import spec

torch.manual_seed(0)


_DEV = [torch.device('cpu')]


@pytest.fixture(autouse=True)
def _track_device(request):
    _DEV[0] = request.getfixturevalue('be').dev if 'be' in request.fixturenames else torch.device('cpu')
    yield


def D(t):
    """to the backend's device"""
    return t.to(_DEV[0])


def nhwc(t):
    return D(t.permute(0, 2, 3, 1).contiguous())


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous().cpu()


def close(a, b, rt=1e-4, at=1e-5):
    torch.testing.assert_close(a.cpu(), b.cpu(), rtol=rt, atol=at)


@pytest.mark.parametrize('cin,cout,k,stride,groups', [
    (3, 32, 3, 2, 1), (32, 32, 3, 1, 1), (32, 2, 3, 1, 1), (64, 1, 3, 1, 1), (1, 64, 3, 1, 1),
    (64, 16, 1, 1, 4), (64, 64, 1, 1, 4), (16, 64, 1, 1, 1), (32, 64, 1, 2, 1), (20, 12, 3, 2, 1)])
def test_conv2d_forward(be, cin, cout, k, stride, groups):
    from led_net_amd import ops
    x = torch.randn(2, cin, 9, 11)
    w = torch.randn(cout, cin // groups, k, k) * 0.2
    s_in, b_in = torch.rand(cin) + 0.5, torch.randn(cin) * 0.1
    s_o, b_o = torch.rand(cout) + 0.5, torch.randn(cout) * 0.1
    slope = torch.rand(cout) * 0.3
    pad = k // 2
    ref = F.conv2d(F.relu(x * s_in.view(1, -1, 1, 1) + b_in.view(1, -1, 1, 1)), w, stride=stride,
                   padding=pad, groups=groups)
    v = ref * s_o.view(1, -1, 1, 1) + b_o.view(1, -1, 1, 1)
    res = torch.randn_like(v)
    want = F.prelu(v + res, slope)
    stats = (D(torch.zeros(cout)), D(torch.zeros(cout)))
    got = ops.conv2d(nhwc(x), D(w), stride=stride, pad=pad, groups=groups, in_scale=D(s_in), in_shift=D(b_in),
                     in_act=ops.ACT_RELU, out_scale=D(s_o), out_shift=D(b_o), act=ops.ACT_PRELU, slope=D(slope),
                     res=nhwc(res), res_mode=ops.RES_ADD, stats=stats)
    close(nchw(got), want)
    close(stats[0], v.sum((0, 2, 3)), 1e-4, 1e-3)
    close(stats[1], (v * v).sum((0, 2, 3)), 1e-4, 1e-3)


@pytest.mark.parametrize('cin,cout,k,stride,groups,hw', [
    (32, 64, 3, 2, 1, (9, 11)), (32, 64, 3, 2, 1, (10, 12)), (16, 8, 3, 1, 1, (7, 5)),
    (64, 16, 1, 1, 4, (6, 6)), (3, 32, 3, 2, 1, (9, 8)), (32, 64, 1, 2, 1, (9, 11))])
def test_conv2d_dgrad_wgrad(be, cin, cout, k, stride, groups, hw):
    from led_net_amd import ops
    x = torch.randn(2, cin, *hw, requires_grad=True)
    w = (torch.randn(cout, cin // groups, k, k) * 0.2).requires_grad_(True)
    pad = k // 2
    z = F.conv2d(x, w, stride=stride, padding=pad, groups=groups)
    dz = torch.randn_like(z)
    z.backward(dz)
    dx = ops.conv2d(nhwc(dz), D(w.detach()), stride=stride, pad=pad, groups=groups, transposed=True,
                    out_hw=hw)
    close(nchw(dx), x.grad)
    dw, db = ops.conv2d_wgrad(nhwc(x.detach()), nhwc(dz), tuple(w.shape), stride=stride, pad=pad,
                              groups=groups, bias=True)
    close(dw, w.grad, 1e-4, 1e-4)
    close(db, dz.sum((0, 2, 3)), 1e-4, 1e-4)


@pytest.mark.parametrize('n,cin,cout,k,stride,groups,hw', [
    (4, 64, 16, 1, 1, 1, (16, 16)), (16, 16, 64, 1, 1, 1, (4, 4)), (3, 1, 64, 3, 1, 1, (40, 40)),
    (2, 64, 2, 1, 1, 1, (48, 40)), (2, 3, 32, 3, 2, 1, (33, 31)), (2, 20, 12, 3, 1, 2, (17, 9)),
    (1, 72, 80, 1, 1, 1, (37, 3))])
def test_conv2d_wgrad_vector_paths(be, n, cin, cout, k, stride, groups, hw):
    """conv_wgrad_direct_kernel (LDS-staged pixel chunks, workspace partial tiles summed by
    finish_partials) and conv_wgrad_narrow_kernel at pixel counts that take the multi-workgroup
    paths, with the producer's BatchNorm+ReLU folded in (in_scale/in_shift/in_act) and
    accumulation into a caller-provided gradient buffer (dw_out)."""
    from led_net_amd import ops
    x = torch.randn(n, cin, *hw)
    w = (torch.randn(cout, cin // groups, k, k) * 0.2).requires_grad_(True)
    s_in, b_in = torch.rand(cin) + 0.5, torch.randn(cin) * 0.1
    pad = k // 2
    z = F.conv2d(F.relu(x * s_in.view(1, -1, 1, 1) + b_in.view(1, -1, 1, 1)), w, stride=stride, padding=pad,
                 groups=groups)
    dz = torch.randn_like(z)
    z.backward(dz)
    base = torch.randn_like(w.detach())
    sink = D(base.clone())
    dw, db = ops.conv2d_wgrad(nhwc(x), nhwc(dz), tuple(w.shape), stride=stride, pad=pad, groups=groups,
                              in_scale=D(s_in), in_shift=D(b_in), in_act=ops.ACT_RELU, bias=True, dw_out=sink)
    assert dw is sink
    close(dw.cpu() - base, w.grad, 1e-4, 1e-4 * (n * hw[0] * hw[1]) ** 0.5)
    close(db, dz.sum((0, 2, 3)), 1e-4, 1e-4 * (n * hw[0] * hw[1]) ** 0.5)


@pytest.mark.parametrize('cin,cout,k,transposed,dt', [
    (32, 2, 3, True, torch.bfloat16), (32, 2, 3, True, torch.float32), (64, 1, 3, True, torch.float32),
    (64, 2, 1, True, torch.bfloat16), (2, 32, 3, False, torch.float32), (2, 64, 1, False, torch.bfloat16),
    (2, 64, 3, False, torch.bfloat16), (128, 2, 3, True, torch.bfloat16), (32, 1, 3, True, torch.bfloat16)])
def test_conv2d_narrow_input_kernel(be, cin, cout, k, transposed, dt):
    """conv_narrowin_kernel: <= 4 channels in, wide out, stride 1, plain epilogue -- the data
    gradient of LEDHead's 32->2 heads (transposed: dz [.,2] -> dx [.,32]) and plain narrow convs.  The 3x3 bf16 cases
    with 32 k output channels run on conv3x3_narrowin_mfma_kernel (csrc/conv3x3.hip: the 3 x 3 x 2 neighbourhood as one
    K = 32 fragment, weights rounded to bf16 as on every matrix-core path)."""
    from led_net_amd import ops
    if dt == torch.bfloat16 and k == 3:
        kin, kout = (cout, cin) if transposed else (cin, cout)
        xq = torch.zeros(1, 4, 4, kin, dtype=dt, device=_DEV[0])
        kid = ops.conv2d_kernel_id(xq, D(torch.zeros(cout, cin, k, k)), pad=1, transposed=transposed,
                                   **(dict(out_hw=(4, 4)) if transposed else {}))
        assert kid == 4, kid
    pad = k // 2
    w = torch.randn(cout, cin, k, k) * 0.2
    if transposed:
        dz = torch.randn(2, cout, 19, 23)
        if dt == torch.bfloat16:
            dz = dz.bfloat16().float()
        want = torch.nn.grad.conv2d_input((2, cin, 19, 23), w, dz, padding=pad)
        got = ops.conv2d(nhwc(dz).to(dt), D(w), pad=pad, transposed=True, out_hw=(19, 23))
    else:
        x = torch.randn(2, cin, 19, 23)
        if dt == torch.bfloat16:
            x = x.bfloat16().float()
        want = F.conv2d(x, w, padding=pad)
        got = ops.conv2d(nhwc(x).to(dt), D(w), pad=pad)
    assert got.dtype == dt
    tol = 2e-2 if dt == torch.bfloat16 else 1e-4
    close(nchw(got.float()), want, tol, tol)


def test_conv2d_bf16(be):
    from led_net_amd import ops
    x = torch.randn(1, 32, 8, 8)
    w = torch.randn(16, 32, 3, 3) * 0.1
    want = F.conv2d(x.bfloat16().float(), w, padding=1)
    got = ops.conv2d(nhwc(x).bfloat16(), D(w), pad=1)
    assert got.dtype == torch.bfloat16
    close(nchw(got.float()), want, 2e-2, 2e-2)


def test_conv2d_rejects_bad_shapes(be):
    from led_net_amd import ops
    with pytest.raises(ops.LednError):
        ops.conv2d(D(torch.randn(1, 4, 4, 8)), D(torch.randn(4, 6, 3, 3)), pad=1)


@pytest.mark.parametrize('stride', [1, 2])
def test_sesp_pyramid_and_dw(be, stride):
    from led_net_amd import ops
    n, dil = 8, [1, 2, 3, 4]
    x = torch.randn(2, n, 13, 10)
    ws = [torch.randn(n, 1, 3, 3) * 0.3 for _ in range(4)]
    outs = []
    for i in range(4):
        o = F.conv2d(x, ws[i], stride=stride, padding=dil[i], dilation=dil[i], groups=n)
        outs.append(o if i == 0 else o + outs[-1])
    want = torch.cat(outs, 1)
    wp = torch.stack([w[:, 0].permute(1, 2, 0) for w in ws]).contiguous()
    got = ops.sesp_pyramid(nhwc(x), D(wp), dil, stride)
    close(nchw(got), want)
    # stage 2: per-group dilation d+1 with affine + PReLU epilogue and statistics
    ws2 = [torch.randn(n, 1, 3, 3) * 0.3 for _ in range(4)]
    o2 = torch.cat([F.conv2d(outs[i], ws2[i], padding=dil[i] + 1, dilation=dil[i] + 1, groups=n)
                    for i in range(4)], 1)
    s, b, sl = torch.rand(4 * n) + 0.5, torch.randn(4 * n), torch.rand(4 * n) * 0.3
    v = o2 * s.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)
    w2 = torch.cat([w[:, 0].permute(1, 2, 0) for w in ws2], 2).contiguous()
    stats = (D(torch.zeros(4 * n)), D(torch.zeros(4 * n)))
    got2 = ops.dwconv2d(got, D(w2), dil=[d + 1 for d in dil], group_size=n, out_scale=D(s), out_shift=D(b),
                        act=ops.ACT_PRELU, slope=D(sl), stats=stats)
    close(nchw(got2), F.prelu(v, sl))
    close(stats[0], v.sum((0, 2, 3)), 1e-4, 1e-3)


def test_dw8x8_ext1(be):
    from led_net_amd import ops
    c = 8
    x = torch.randn(1, c, 13, 11)
    w = torch.randn(c, 1, 8, 8) * 0.1
    want = F.conv2d(F.pad(x, (0, 1, 0, 1), mode='reflect'), w, padding=3, groups=c)
    got = ops.dwconv2d(nhwc(x), D(w[:, 0].permute(1, 2, 0).contiguous()), pad=3, ext1=True)
    close(nchw(got), want)


@pytest.mark.parametrize('c', [1, 2, 3, 64, 128])
def test_channel_stats_bn_finalize_affine(be, c):
    from led_net_amd import ops
    x = torch.randn(3, c, 17, 9) * 2 + 1
    xs = nhwc(x)
    st = ops.channel_stats(xs)
    close(st[0], x.sum((0, 2, 3)), 1e-4, 1e-3)
    close(st[1], (x * x).sum((0, 2, 3)), 1e-4, 1e-2)
    g, b = torch.rand(c) + 0.5, torch.randn(c)
    rm, rv = torch.zeros(c), torch.ones(c)
    rm2, rv2 = rm.clone(), rv.clone()
    want = F.batch_norm(x, rm2, rv2, g, b, True, 0.1, 1e-5)
    rm, rv = D(rm), D(rv)
    scale, shift, mean, invstd = ops.bn_finalize(st, x.numel() // c, D(g), D(b), rm, rv)
    got = ops.affine_act(xs, scale, shift, act=ops.ACT_RELU)
    close(nchw(got), F.relu(want), 1e-4, 1e-4)
    close(rm, rm2)
    close(rv, rv2, 1e-4, 1e-4)


@pytest.mark.parametrize('src,dst', [((9, 17), (18, 34)), ((9, 17), (35, 67)), ((5, 7), (5, 7)), ((8, 8), (3, 5))])
def test_bilinear(be, src, dst):
    from led_net_amd import ops
    x = torch.randn(2, 4, *src)
    add = torch.randn(2, 4, *dst)
    want = F.interpolate(x, size=dst, mode='bilinear', align_corners=False) + add
    got = ops.bilinear(nhwc(x), dst, add=nhwc(add))
    close(nchw(got), want, 1e-5, 5e-6)
    x2 = torch.relu(torch.randn(2, 2, *src))
    want2 = F.interpolate(x2, size=dst, mode='bilinear', align_corners=False)
    got2, am = ops.bilinear(nhwc(x2), dst, nchw=True, argmax=True)
    close(got2, want2, 1e-5, 1e-6)
    assert torch.equal(am.long().cpu(), got2.argmax(1).cpu())


def test_pools(be):
    from led_net_amd import ops
    x, r = torch.randn(2, 8, 19, 21), torch.randn(2, 8, 19, 21)
    for S in (1, 4, 8, 16):
        close(nchw(ops.adaptive_avgpool(nhwc(x), S, xadd=nhwc(r))), F.adaptive_avg_pool2d(x + r, S))
    close(nchw(ops.avgpool3x3s2(nhwc(x))), F.avg_pool2d(x, 3, 2, 1))


def test_nchw_to_nhwc_preprocess(be):
    from led_net_amd import ops
    img = torch.randint(0, 256, (2, 3, 6, 5), dtype=torch.uint8)
    want = spec.preprocess(img)
    mean, std = torch.tensor([123.675, 116.28, 103.53]), torch.tensor([58.395, 57.12, 57.375])
    got = ops.nchw_to_nhwc(D(img), torch.float32, D((1 / std).contiguous()), D((-mean / std).contiguous()),
                           D(torch.tensor([2, 1, 0], dtype=torch.int32)))
    close(nchw(got), want, 1e-5, 1e-5)


@pytest.mark.parametrize('percentile', [0.8, None])
@pytest.mark.parametrize('hw', [(24, 36), (23, 37), (128, 128), (132, 126)])
def test_seam_edge(be, percentile, hw):
    """(128 x 128 = 16 K pixels: the largest map of the LDS-resident kernel, the size at 1024 x 1024 inputs;
    132 x 126: the generic kernel)"""
    from led_net_amd import ops
    seg = torch.randn(2, 1, *hw, generator=torch.Generator().manual_seed(hw[0] * 1000 + hw[1]))
    want = spec.seam_edge(seg, 'p80' if percentile else 0.1)
    got = ops.seam_edge(nhwc(seg), percentile, 0.1, 0.1)
    mism = (nchw(got) != want).float().mean().item()
    assert mism < 0.01, mism      # fp32 summation-order ties at the threshold only
