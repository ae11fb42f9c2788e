"""EncoderDecoder plumbing around the hot path: SegDataPreProcessor (pad + stack, data_preprocessor.py:98-151,
utils/misc.py:30-128) fused into the stem's input kernel, postprocess_result (segmentors/base.py:127-200: un-pad,
flip, resize to ori_shape, argmax) and LEDHead.loss_by_feat (led_head.py:101-146) -- against the CPU oracle."""
import os

import pytest
import torch
import torch.nn.functional as F

from oracle import spec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model(be):
    import led_net_amd as L
    torch.manual_seed(21)
    cfg = L.load_config(os.path.join(ROOT, 'tests', 'data', 'lednet_test_config.py'))
    model = L.MODELS.build(cfg['model'])
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    return L, cfg, model.to(be.dev), sd


def test_preprocessor_pads_and_stacks_like_stack_batch(be):
    L, cfg, model, _ = _model(be)
    pre = L.SegDataPreProcessor(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], size=(40, 48), pad_val=0,
                                seg_pad_val=255, bgr_to_rgb=True).to(be.dev)
    g = torch.Generator().manual_seed(1)
    imgs = [torch.randint(0, 256, (3, 33, 41), dtype=torch.uint8, generator=g),
            torch.randint(0, 256, (3, 40, 37), dtype=torch.uint8, generator=g)]
    labs = [torch.randint(0, 2, (1, 33, 41), generator=g), torch.randint(0, 2, (1, 40, 37), generator=g)]
    samples = [L.SegDataSample(gt=l.clone()) for l in labs]
    out = pre(dict(inputs=[i.clone() for i in imgs], data_samples=samples), training=True)
    assert out['inputs'].shape == (2, 3, 40, 48) and out['inputs'].dtype == torch.uint8
    assert samples[0].metainfo['padding_size'] == (0, 7, 0, 7) and samples[0].metainfo['img_shape'] == (33, 41)
    assert samples[1].metainfo['padding_size'] == (0, 11, 0, 0) and samples[1].metainfo['pad_shape'] == (40, 48)
    assert samples[0].gt_sem_seg.data.shape == (1, 40, 48)
    assert (samples[0].gt_sem_seg.data[:, 33:, :] == 255).all() and (samples[0].gt_sem_seg.data[:, :, 41:] == 255).all()
    assert torch.equal(samples[0].gt_sem_seg.data[:, :33, :41].cpu(), labs[0])
    # the stem's input kernel: normalise, BGR->RGB, pad_val AFTER the normalisation == the reference's order
    from led_net_amd import ops
    valid = torch.tensor([[33, 41], [40, 37]], dtype=torch.int32).to(be.dev)
    y = ops.nchw_to_nhwc(out['inputs'].contiguous(), torch.float32, pre.scale, pre.shift, pre.chan_map, valid, 0.0)
    want = torch.stack([F.pad(spec.preprocess(i[None])[0], (0, 48 - i.shape[2], 0, 40 - i.shape[1]), value=0.0)
                        for i in imgs])
    torch.testing.assert_close(y.permute(0, 3, 1, 2).cpu(), want, rtol=1e-5, atol=1e-5)


def test_predict_unpads_and_resizes_like_postprocess_result(be):
    """test-time padding (test_cfg size_divisor) + ori_shape resize: per image crop, bilinear to ori_shape, argmax"""
    L, cfg, model, sd = _model(be)
    model.eval()
    model.data_preprocessor.test_cfg = dict(size_divisor=64)
    g = torch.Generator().manual_seed(2)
    imgs = [torch.randint(0, 256, (3, 280, 300), dtype=torch.uint8, generator=g) for _ in range(2)]
    samples = [L.SegDataSample(metainfo=dict(ori_shape=(210, 225))), L.SegDataSample(metainfo=dict(ori_shape=(280, 300), flip=True, flip_direction='horizontal'))]
    data = model.data_preprocessor(dict(inputs=imgs, data_samples=samples), training=False)
    assert data['inputs'].shape == (2, 3, 320, 320) and samples[0].metainfo['img_padding_size'] == (0, 20, 0, 40)
    with torch.no_grad():
        out = model(data['inputs'], data['data_samples'], mode='predict')
        # oracle: normalise, zero-pad (pad_val 0 after the normalisation), forward, crop, flip, resize, argmax
        x = torch.stack([F.pad(spec.preprocess(i[None])[0], (0, 20, 0, 40), value=0.0) for i in imgs])
        logits, _ = spec.predict(x, sd)
    for i, ds in enumerate(out):
        lg = logits[i:i + 1, :, :280, :300]
        if ds.metainfo.get('flip'):
            lg = lg.flip(dims=(3,))
        want = F.interpolate(lg, size=ds.metainfo['ori_shape'], mode='bilinear', align_corners=False)[0]
        got = ds.seg_logits.data.cpu()
        assert got.shape == want.shape
        assert (got - want).abs().max().item() < 5e-3
        margin = (want[0] - want[1]).abs()
        mask = ds.pred_sem_seg.data.long().cpu()[0]
        assert ((mask != want.argmax(0)) & (margin > 1e-3)).sum().item() == 0
        assert mask.shape == tuple(ds.metainfo['ori_shape'])


def test_ledhead_loss_by_feat_is_a_method(be):
    """led_head.py:101-146: loss_by_feat(seg_logits, batch_data_samples) on the 4-tuple forward() returns in training"""
    L, cfg, model, sd = _model(be)
    model.train()
    g = torch.Generator().manual_seed(3)
    img = torch.randint(0, 256, (2, 3, 320, 320), dtype=torch.uint8, generator=g)
    lab = torch.randint(0, 2, (2, 1, 320, 320), generator=g)
    lab[:, :, :9] = 255
    samples = [L.SegDataSample(gt=be(lab)[i]) for i in range(2)]
    for c in model.decode_head.loss_decode:
        c.min_kept = 5000
    feats = model.extract_feat(be(img))
    seg_logits = model.decode_head.forward(feats)
    assert len(seg_logits) == 4 and seg_logits[0].shape[1] == 2
    out = model.decode_head.loss_by_feat(seg_logits, samples)
    assert set(out) == {'loss_context', 'loss_spatial', 'acc_seg'}
    # oracle on the SAME logits
    want = spec.led_head_loss(tuple(t.detach().float().cpu() for t in seg_logits), lab,
                              loss_cfg=((0.9, 5000, 1.0), (0.9, 5000, 0.4)))
    for k in out:
        assert abs(float(out[k].reshape(-1)[0]) - float(want[k])) <= 2e-4 * abs(float(want[k])) + 1e-5, k
    # ... and LEDHead.loss is forward + loss_by_feat
    out2 = model.decode_head.loss(feats, samples)
    for k in out:
        assert abs(float(out2[k].reshape(-1)[0]) - float(out[k].reshape(-1)[0])) <= 1e-5 * abs(float(out[k].reshape(-1)[0])) + 1e-6
