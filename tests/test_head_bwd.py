"""BatchNorm / activation backward of LEDHead's two-class heads (norm -> act -> 3x3 conv 32 -> 2, led_head.py:44-51) straight
from the logits' gradient (csrc/head_bwd.hip: ledn_head_bwd_reduce / _apply, dy recomputed on the matrix cores): against
torch autograd of F.batch_norm(training) -> act -> F.conv2d in f32 on the same bf16-rounded tensors, and against the
layer-wise kernel sequence it replaces."""
import pytest
import torch
import torch.nn.functional as F


def _case(dev, N, H, W, act, addend, seed, bias=False):
    from led_net_amd import ops
    C = 32
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, C, H, W, generator=g).bfloat16().float()
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g) * 0.2
    w = (torch.randn(2, C, 3, 3, generator=g) * 0.1).bfloat16().float()     # (the kernels round the filter to bf16)
    slope = torch.rand(C, generator=g) * 0.3 if act == ops.ACT_PRELU else None
    dz = torch.randn(N, 2, H, W, generator=g).bfloat16().float()
    add = torch.randn(N, C, H, W, generator=g).bfloat16().float() if addend else None
    xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    sr = slope.clone().requires_grad_(True) if slope is not None else None
    y = F.batch_norm(xr, None, None, gr, br, training=True, eps=1e-5)
    t = F.relu(y) if act == ops.ACT_RELU else (F.prelu(y, sr) if act == ops.ACT_PRELU else y)
    wr = w.clone().requires_grad_(True)
    bb = torch.zeros(2, requires_grad=True) if bias else None
    z = F.conv2d(t, wr, bb, padding=1)
    z.backward(dz)
    want = dict(dx=xr.grad + (add if addend else 0), dgamma=gr.grad, dbeta=br.grad, dslope=sr.grad if sr is not None else None,
                dw=wr.grad, db=bb.grad if bias else None)
    mean = x.mean((0, 2, 3))
    var = x.var((0, 2, 3), unbiased=False)
    invstd = (var + 1e-5).rsqrt()
    scale, shift = gamma * invstd, beta - mean * gamma * invstd
    nhwc = lambda v: v.permute(0, 2, 3, 1).contiguous()      # noqa: E731
    D = lambda v: None if v is None else v.to(dev)           # noqa: E731
    args = dict(x=D(nhwc(x).bfloat16()), dz=D(nhwc(dz).bfloat16()), w=D(w), scale=D(scale), shift=D(shift), mean=D(mean),
                invstd=D(invstd), slope=D(slope), add=D(nhwc(add).bfloat16()) if addend else None, count=N * H * W)
    return args, want


def _run_head(a, act, dev, wgrad=None):
    from led_net_amd import ops_train as T
    sinks = (torch.zeros(32, device=dev), torch.zeros(32, device=dev), torch.zeros(32, device=dev) if a['slope'] is not None else None)
    assert T.head_bwd_ok(a['x'], a['dz'], a['w'], 1, 1, 1, act)
    head = (a['dz'], a['w']) + (tuple(wgrad) if wgrad is not None else ())
    dx, _, _, _, _ = T.bn_act_bwd(a['x'], None, scale=a['scale'], shift=a['shift'], mean=a['mean'], invstd=a['invstd'], act=act,
                                   slope=a['slope'], count=a['count'], sinks=sinks, dz_add=a['add'], head=head)
    return dx, sinks


@pytest.mark.parametrize('nhw', [(1, 128, 128), (2, 33, 256), (1, 127, 131), (1, 16, 1030)])
@pytest.mark.parametrize('variant', ['relu_bias', 'prelu', 'none_bias'])
def test_head_bwd_weight_gradient(be, nhw, variant):
    """the reduce pass with the head's weight / bias gradient riding along (ragged widths included): against autograd (the
    activation enters the matrix instruction rounded to bf16, as in the layer-wise weight-gradient kernel), accumulating
    into the sinks, two runs bit-identical"""
    from led_net_amd import ops
    act = {'relu_bias': ops.ACT_RELU, 'prelu': ops.ACT_PRELU, 'none_bias': ops.ACT_NONE}[variant]
    a, want = _case(be.dev, *nhw, act, False, nhw[1] * 3 + nhw[2], bias='bias' in variant)
    outs = []
    for _ in range(2):
        dw = torch.zeros(2, 32, 3, 3, device=be.dev)
        db = torch.zeros(2, device=be.dev) if 'bias' in variant else None
        dx, sinks = _run_head(a, act, be.dev, (dw, db))
        outs.append((dx, sinks, dw, db))
    (dx, sinks, dw, db), (dx2, _, dw2, db2) = outs
    assert torch.equal(dw, dw2) and torch.equal(dx, dx2) and (db is None or torch.equal(db, db2))
    torch.testing.assert_close(dw.cpu(), want['dw'], rtol=1e-2, atol=1e-2 * float(want['dw'].abs().max()))
    if db is not None:
        torch.testing.assert_close(db.cpu(), want['db'], rtol=1e-3, atol=1e-3 * float(want['db'].abs().max()))
    torch.testing.assert_close(sinks[0].cpu(), want['dgamma'], rtol=2e-3, atol=2e-3 * float(want['dgamma'].abs().max()))
    got_dx = dx.float().cpu().permute(0, 3, 1, 2)
    torch.testing.assert_close(got_dx, want['dx'], rtol=2e-2, atol=2e-2 * float(want['dx'].abs().max()))
    # accumulation into the sinks
    _run_head(a, act, be.dev, (dw, db))
    torch.testing.assert_close(dw, 2 * dw2, rtol=1e-6, atol=0)


@pytest.mark.parametrize('nhw', [(1, 128, 130), (2, 97, 131), (1, 16, 1030)])
@pytest.mark.parametrize('variant', ['relu_add', 'prelu', 'none', 'relu'])
def test_head_bwd_against_autograd(be, nhw, variant):
    from led_net_amd import ops
    act = {'relu_add': ops.ACT_RELU, 'prelu': ops.ACT_PRELU, 'none': ops.ACT_NONE, 'relu': ops.ACT_RELU}[variant]
    a, want = _case(be.dev, *nhw, act, 'add' in variant, nhw[1] + nhw[2])
    dx, sinks = _run_head(a, act, be.dev)
    got_dx = dx.float().cpu().permute(0, 3, 1, 2)
    # dx is stored as bf16: half an ulp of the largest magnitude + the BatchNorm terms' cancellation
    torch.testing.assert_close(got_dx, want['dx'], rtol=2e-2, atol=2e-2 * float(want['dx'].abs().max()))
    torch.testing.assert_close(sinks[0].cpu(), want['dgamma'], rtol=2e-3, atol=2e-3 * float(want['dgamma'].abs().max()))
    torch.testing.assert_close(sinks[1].cpu(), want['dbeta'], rtol=2e-3, atol=2e-3 * float(want['dbeta'].abs().max()))
    if want['dslope'] is not None:
        torch.testing.assert_close(sinks[2].cpu(), want['dslope'], rtol=2e-3, atol=2e-3 * float(want['dslope'].abs().max()))


def test_head_bwd_accumulates_into_sinks(be):
    """the BatchNorm sinks are accumulated into, not overwritten (they are the trainer's flat gradient buffer)"""
    from led_net_amd import ops, ops_train as T
    a, _ = _case(be.dev, 1, 128, 128, ops.ACT_RELU, True, 5)
    _, s1 = _run_head(a, ops.ACT_RELU, be.dev)
    sinks = (s1[0].clone(), s1[1].clone(), None)
    T.bn_act_bwd(a['x'], None, scale=a['scale'], shift=a['shift'], mean=a['mean'], invstd=a['invstd'], act=ops.ACT_RELU,
                 count=a['count'], sinks=sinks, dz_add=a['add'], head=(a['dz'], a['w']))
    torch.testing.assert_close(sinks[0], 2 * s1[0], rtol=1e-3, atol=1e-3 * float(s1[0].abs().max()))
    torch.testing.assert_close(sinks[1], 2 * s1[1], rtol=1e-3, atol=1e-3 * float(s1[1].abs().max()))


def test_head_bwd_matches_layerwise_kernels(be):
    """the head form vs the layer-wise sequence (transposed conv -> BatchNorm-backward reduce + apply): the same gradients
    up to the bf16 rounding of dy the layer-wise form adds"""
    from led_net_amd import ops, ops_train as T
    a, _ = _case(be.dev, 1, 128, 136, ops.ACT_RELU, True, 11)
    dx, sinks = _run_head(a, ops.ACT_RELU, be.dev)
    x, dz, w = a['x'], a['dz'], a['w']
    dy = ops.conv2d(dz, w, stride=1, pad=1, transposed=True, out_hw=(x.shape[1], x.shape[2]), out_dtype=x.dtype)
    s2 = (torch.zeros(32, device=be.dev), torch.zeros(32, device=be.dev), None)
    dx2, _, _, _, _ = T.bn_act_bwd(x, dy, scale=a['scale'], shift=a['shift'], mean=a['mean'], invstd=a['invstd'],
                                    act=ops.ACT_RELU, count=a['count'], sinks=s2, dz_add=a['add'])
    torch.testing.assert_close(sinks[0].cpu(), s2[0].cpu(), rtol=2e-2, atol=2e-2 * float(s2[0].abs().max()))
    torch.testing.assert_close(sinks[1].cpu(), s2[1].cpu(), rtol=2e-2, atol=2e-2 * float(s2[1].abs().max()))
    torch.testing.assert_close(dx.float().cpu(), dx2.float().cpu(), rtol=3e-2, atol=3e-2 * float(dx2.float().abs().max()))


def test_head_bwd_gate():
    """shapes outside the kernels' gate keep the layer-wise path (BNActConvFn.backward asks head_bwd_ok)"""
    from led_net_amd import ops, ops_train as T
    x = torch.zeros(1, 128, 128, 32, dtype=torch.bfloat16)
    dz = torch.zeros(1, 128, 128, 2, dtype=torch.bfloat16)
    w = torch.zeros(2, 32, 3, 3)
    assert T.head_bwd_ok(x, dz, w, 1, 1, 1, ops.ACT_RELU)
    assert not T.head_bwd_ok(x, dz.float(), w, 1, 1, 1, ops.ACT_RELU)                 # f32 logits
    assert not T.head_bwd_ok(x, dz, w, 2, 1, 1, ops.ACT_RELU)                         # stride
    assert not T.head_bwd_ok(x[:, :64], dz[:, :64], w, 1, 1, 1, ops.ACT_RELU)         # too few pixels
    assert not T.head_bwd_ok(x, torch.zeros(1, 128, 128, 3, dtype=torch.bfloat16), torch.zeros(3, 32, 3, 3), 1, 1, 1, ops.ACT_RELU)
    assert not T.head_bwd_ok(x, dz, w, 1, 1, 1, ops.ACT_RELU6)


@pytest.mark.parametrize('nhw', [(1, 128, 130), (2, 97, 131), (1, 16, 1030), (1, 520, 32)])
@pytest.mark.parametrize('variant', ['relu_bias_f32', 'prelu_bf16', 'plain_bf16', 'relu_bf16'])
def test_head_forward_kernel(be, nhw, variant):
    """head_fwd_kernel (the two-class heads' forward: one channel contraction per pixel for all 18 (tap, class) pairs, the
    taps as a shifted sum) against F.conv2d on the bf16-rounded activation; ragged widths / heights, both output types"""
    from led_net_amd import ops
    N, H, W = nhw
    g = torch.Generator().manual_seed(H * 7 + W)
    x = torch.randn(N, 32, H, W, generator=g).bfloat16().float()
    w = (torch.randn(2, 32, 3, 3, generator=g) * 0.1).bfloat16().float()
    bias = torch.randn(2, generator=g) * 0.1 if 'bias' in variant else None
    sc, sh = torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g) * 0.3
    act = ops.ACT_RELU if 'relu' in variant else (ops.ACT_PRELU if 'prelu' in variant else ops.ACT_NONE)
    sl = torch.rand(32, generator=g) * 0.3 if act == ops.ACT_PRELU else None
    plain = 'plain' in variant
    t = x if plain else x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    if act == ops.ACT_RELU:
        t = F.relu(t)
    elif act == ops.ACT_PRELU:
        t = F.prelu(t, sl)
    want = F.conv2d(t.bfloat16().float(), w, bias, padding=1)
    od = torch.float32 if 'f32' in variant else torch.bfloat16
    D = lambda v: None if v is None else v.to(be.dev)     # noqa: E731
    kw = dict(stride=1, pad=1, out_dtype=od, out_shift=D(bias))
    if not plain:
        kw.update(in_scale=D(sc), in_shift=D(sh), in_act=act, in_slope=D(sl))
    xd = x.permute(0, 2, 3, 1).contiguous().bfloat16().to(be.dev)
    assert ops.conv2d_kernel_id(xd, D(w), **kw) == 6
    got = ops.conv2d(xd, D(w), **kw)
    assert got.dtype == od
    tol = 2e-2 if od == torch.bfloat16 else 2e-3
    torch.testing.assert_close(got.float().cpu().permute(0, 3, 1, 2), want, rtol=tol, atol=tol * float(want.abs().max()))


@pytest.mark.parametrize('shape', [(1, 64, 70, 64), (2, 33, 64, 32), (1, 40, 103, 128)])
@pytest.mark.parametrize('cout,od', [(2, 'f32'), (1, 'bf16'), (2, 'bf16')])
def test_classifier_1x1_narrow_forward_and_weight_gradient(be, shape, cout, od):
    """the 1x1 classifiers (cls_seg / aux_cls_seg, led_head.py:87-98: C -> num_classes <= 2) on the streaming kernels:
    conv1x1_narrow_kernel (forward) and the unrolled 1x1 path of conv_wgrad_cout2_kernel, against torch"""
    from led_net_amd import ops
    N, H, W, Cc = shape
    g = torch.Generator().manual_seed(H + W + Cc + cout)
    x = torch.randn(N, Cc, H, W, generator=g).bfloat16().float().requires_grad_(True)
    w = (torch.randn(cout, Cc, 1, 1, generator=g) * 0.2).requires_grad_(True)
    b = (torch.randn(cout, generator=g) * 0.1).requires_grad_(True)
    z = F.conv2d(x, w, b)
    dz = torch.randn(z.shape, generator=g)
    if od == 'bf16':
        dz = dz.bfloat16().float()
    z.backward(dz)
    dt = torch.float32 if od == 'f32' else torch.bfloat16
    xd = x.detach().permute(0, 2, 3, 1).contiguous().bfloat16().to(be.dev)
    got = ops.conv2d(xd, w.detach().to(be.dev), out_shift=b.detach().to(be.dev), out_dtype=dt)
    tol = 2e-3 if od == 'f32' else 1.5e-2
    torch.testing.assert_close(got.float().cpu().permute(0, 3, 1, 2), z.detach(), rtol=tol, atol=tol * float(z.abs().max()))
    dzd = dz.permute(0, 2, 3, 1).contiguous().to(dt).to(be.dev)
    dw, db = ops.conv2d_wgrad(xd, dzd, (cout, Cc, 1, 1), bias=True)
    torch.testing.assert_close(dw.cpu(), w.grad, rtol=2e-3, atol=2e-3 * float(w.grad.abs().max()))
    torch.testing.assert_close(db.cpu(), b.grad, rtol=2e-3, atol=2e-3 * float(b.grad.abs().max()))
