"""the train-time augmentation pipeline (RandomResize -> RandomCrop -> RandomFlip ->
PhotoMetricDistortion -> PackSegInputs) as ONE GPU launch per batch (led_net_amd.transforms / csrc/augment.hip)
against the numpy restatement of the reference pipeline (oracle/augment.py), bit for bit, under the same
numpy.random seed -- i.e. the product draws the reference's random numbers in the reference's order.

The parts of the oracle that restate un-vendored third-party code (cv2.resize / cv2.cvtColor / mmcv size rules) have
no golden vector in this image ("parity unpinned", see the oracle's header); they are sanity-pinned here by
published known answers (mmcv's imrescale test sizes, the HSV primaries) and by PIL within its rounding."""
import numpy as np
import pytest
import torch

from oracle import augment as OA

PIPE = [
    dict(type='LoadImageFromFile'),
    dict(type='LoadAnnotations'),
    dict(type='RandomResize', scale=(160, 80), ratio_range=(0.5, 2.0), keep_ratio=True),
    dict(type='RandomCrop', crop_size=(64, 96), cat_max_ratio=0.75),
    dict(type='RandomFlip', prob=0.5),
    dict(type='PhotoMetricDistortion'),
    dict(type='PackSegInputs'),
]


def _samples(seed, n, sizes=None, classes=3):
    g = np.random.RandomState(seed)
    out = []
    for i in range(n):
        h, w = sizes[i] if sizes else (int(g.randint(40, 120)), int(g.randint(60, 200)))
        img = g.randint(0, 256, (h, w, 3)).astype(np.uint8)
        # smooth-ish content so that bilinear interpolation and the HSV sectors are all exercised
        img[: h // 2] = (img[: h // 2].astype(np.int32) // 4 + np.arange(w)[None, :, None] % 192).astype(np.uint8)
        seg = (g.randint(0, classes, (h // 8 + 1, w // 8 + 1)).astype(np.uint8)).repeat(8, 0).repeat(8, 1)[:h, :w]
        seg[:3] = 255
        out.append((img, np.ascontiguousarray(seg)))
    return out


# ------------------------------------------------------------------ oracle sanity pins (no GPU, no product code)
def test_oracle_mmcv_rescale_size_known_answers():
    # mmcv tests/test_image/test_geometric.py::test_imrescale on a 300 x 400 (h x w) image
    assert OA.rescale_size((400, 300), 1.5) == (600, 450)
    assert OA.rescale_size((400, 300), 0.934) == (374, 280)
    assert OA.rescale_size((400, 300), (1000, 600)) == (800, 600)
    assert OA.rescale_size((400, 300), (1000, 200)) == (267, 200)
    assert OA.rescale_size((400, 300), (200, 1000)) == (267, 200)


def test_oracle_hsv_known_answers_and_roundtrip():
    # OpenCV documentation, 8-bit BGR2HSV (H in [0, 180)): pure blue / green / red, white, black, mid grey
    bgr = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [0, 0, 0], [128, 128, 128],
                     [0, 255, 255], [255, 255, 0], [255, 0, 255]]], np.uint8)
    hsv = OA.bgr2hsv_u8(bgr)
    want = [[120, 255, 255], [60, 255, 255], [0, 255, 255], [0, 0, 255], [0, 0, 0], [0, 0, 128],
            [30, 255, 255], [90, 255, 255], [150, 255, 255]]
    assert hsv[0].tolist() == want
    assert np.array_equal(OA.hsv2bgr_u8(hsv), bgr)
    g = np.random.RandomState(0)
    x = g.randint(0, 256, (64, 64, 3)).astype(np.uint8)
    back = OA.hsv2bgr_u8(OA.bgr2hsv_u8(x)).astype(int)
    assert np.abs(back - x.astype(int)).max() <= 4          # 8-bit HSV is lossy by a few levels, never more


def test_oracle_hsv_close_to_pil():
    from PIL import Image
    g = np.random.RandomState(1)
    x = g.randint(0, 256, (32, 32, 3)).astype(np.uint8)
    hsv = OA.bgr2hsv_u8(x).astype(int)
    pil = np.asarray(Image.fromarray(x[..., ::-1].copy(), 'RGB').convert('HSV')).astype(int)    # H in [0, 255]
    sat_ok = hsv[..., 1] > 40                                # hue is ill-conditioned near grey
    dh = np.abs(hsv[..., 0] * 255.0 / 180.0 - pil[..., 0])
    dh = np.minimum(dh, 255 - dh)
    assert dh[sat_ok].max() <= 4.0
    assert np.abs(hsv[..., 1] - pil[..., 1]).max() <= 2 and np.array_equal(hsv[..., 2], pil[..., 2])


def test_oracle_resize_properties_and_pil():
    from PIL import Image
    g = np.random.RandomState(2)
    x = g.randint(0, 256, (37, 53, 3)).astype(np.uint8)
    assert np.array_equal(OA.resize_bilinear_u8(x, (53, 37)), x)
    const = np.full((20, 30, 3), 77, np.uint8)
    assert (OA.resize_bilinear_u8(const, (71, 45)) == 77).all() and (OA.resize_bilinear_u8(const, (11, 9)) == 77).all()
    # exact 2x upscale of a horizontal ramp: interior samples sit at 1/4, 3/4 between neighbours
    ramp = np.tile((np.arange(16) * 16).astype(np.uint8)[None, :, None], (4, 1, 3))
    up = OA.resize_bilinear_u8(ramp, (32, 8))[0, :, 0].astype(int)
    assert up[0] == 0 and up[1] == 4 and up[2] == 12 and up[3] == 20 and up[-1] == 240
    # upscaling: PIL's BILINEAR samples the same positions with float weights -> within one grey level
    big = OA.resize_bilinear_u8(x, (106, 74)).astype(int)
    pil = np.asarray(Image.fromarray(x).resize((106, 74), Image.BILINEAR)).astype(int)
    assert np.abs(big - pil).max() <= 1
    seg = g.randint(0, 5, (37, 53)).astype(np.uint8)
    assert np.array_equal(OA.resize_nearest(seg, (106, 74)), seg.repeat(2, 0).repeat(2, 1))
    assert np.array_equal(OA.resize_nearest(seg, (53, 37)), seg)


def test_oracle_convert_is_the_reference_statement():
    x = np.arange(256, dtype=np.uint8).reshape(16, 16)
    assert np.array_equal(OA.convert(x, alpha=1.5), np.clip(x.astype(np.float32) * 1.5, 0, 255).astype(np.uint8))
    assert OA.convert(x, beta=-20.7).reshape(-1)[:23].tolist() == [0] * 21 + [0, 1]      # 21 - 20.7 = 0.3 -> 0
    assert OA.convert(np.array([200], np.uint8), beta=31.9)[0] == 231       # truncation, not rounding


# ------------------------------------------------------------------ product vs oracle
def _run_product(be, pipe_cfg, samples, seed, out_hw=None):
    from led_net_amd import transforms as T
    pipe = T.Compose(pipe_cfg)
    dev = be.dev
    recs = [dict(img=torch.from_numpy(i).to(dev), gt_seg_map=torch.from_numpy(s).to(dev)) for i, s in samples]
    np.random.seed(seed)
    out = pipe.batch(recs, out_hw=out_hw)
    state = np.random.get_state()[1][:8].tolist(), np.random.get_state()[2]
    return out, state


def _run_oracle(pipe_cfg, samples, seed):
    kw = {c['type']: c for c in pipe_cfg}
    rr, rc, rf = kw['RandomResize'], kw['RandomCrop'], kw['RandomFlip']
    np.random.seed(seed)
    res = [OA.train_pipeline(i, s, rr['scale'], rr['ratio_range'], rc['crop_size'], rc.get('cat_max_ratio', 1.0),
                             rf['prob']) for i, s in samples]
    state = np.random.get_state()[1][:8].tolist(), np.random.get_state()[2]
    return res, state


@pytest.mark.parametrize('seed', [0, 1, 2, 3, 4, 5])
def test_pipeline_bit_exact_vs_oracle(be, seed):
    samples = _samples(100 + seed, 5)
    out, st_p = _run_product(be, PIPE, samples, seed)
    want, st_o = _run_oracle(PIPE, samples, seed)
    assert st_p == st_o, 'the product consumed numpy.random differently from the reference pipeline'
    for k, (img, seg, meta) in enumerate(want):
        got_i, ds = out['inputs'][k].cpu().numpy(), out['data_samples'][k]
        assert got_i.shape == img.shape, (k, got_i.shape, img.shape, meta)
        assert np.array_equal(got_i, img), (k, meta, int(np.abs(got_i.astype(int) - img.astype(int)).max()))
        got_s = ds.gt_sem_seg.data.cpu().numpy()
        assert got_s.dtype == np.int64 and np.array_equal(got_s, seg), (k, meta)
        assert tuple(ds.metainfo['img_shape']) == tuple(meta['img_shape']) and ds.metainfo['flip'] == meta['flip']
    # the batch the views share: padded with pad_val / seg_pad_val outside each sample's extent
    b, lab = out['batch'].cpu(), out['labels'].cpu()
    for k, (img, _, _) in enumerate(want):
        h, w = img.shape[1:]
        assert (b[k, :, h:, :] == 0).all() and (b[k, :, :, w:] == 0).all()
        assert (lab[k, :, h:, :] == 255).all() and (lab[k, :, :, w:] == 255).all()


def test_every_photometric_branch_was_exercised():
    """the seeds above cover brightness / both contrast positions / saturation / hue on and off, flips and
    crops smaller than the crop size"""
    seen = dict(bright=set(), contrast=set(), sat=set(), hue=set(), mode=set(), flip=set(), short=set())
    from led_net_amd import transforms as T
    pipe = T.Compose(PIPE)
    for seed in range(6):
        np.random.seed(seed)
        for img, seg in _samples(100 + seed, 5):
            r = dict(img=torch.from_numpy(img), gt_seg_map=None)
            for t in pipe.transforms:
                t.transform(r)
            p = r['_aug']
            for k in ('bright', 'contrast', 'sat', 'hue'):
                seen[k].add(getattr(p, k) is not None)
            seen['mode'].add(p.mode)
            seen['flip'].add(p.flip)
            seen['short'].add(p.ch < 64 or p.cw < 96)
    assert all(len(v) == 2 for v in seen.values()), seen


def test_crop_retry_path_follows_the_reference_draws(be):
    """cat_max_ratio: a label map dominated by one class forces the 10-try loop; the number of numpy.random draws
    (hence every later decision) must match the reference's"""
    g = np.random.RandomState(7)
    samples = []
    for _ in range(3):
        img = g.randint(0, 256, (90, 150, 3)).astype(np.uint8)
        seg = np.zeros((90, 150), np.uint8)
        seg[60:, 100:] = 1                       # minority class only in one corner
        samples.append((img, seg))
    pipe_cfg = [dict(c) for c in PIPE]
    pipe_cfg[2] = dict(type='RandomResize', scale=(150, 90), ratio_range=(1.0, 1.5), keep_ratio=True)
    pipe_cfg[3] = dict(type='RandomCrop', crop_size=(48, 48), cat_max_ratio=0.75)
    out, st_p = _run_product(be, pipe_cfg, samples, 11)
    want, st_o = _run_oracle(pipe_cfg, samples, 11)
    assert st_p == st_o
    for k, (img, seg, meta) in enumerate(want):
        assert np.array_equal(out['inputs'][k].cpu().numpy(), img), (k, meta)
        assert np.array_equal(out['data_samples'][k].gt_sem_seg.data.cpu().numpy(), seg)


def test_fixed_resize_test_pipeline_and_no_resize(be):
    from led_net_amd import transforms as T
    img, seg = _samples(5, 1, sizes=[(50, 70)])[0]
    dev = be.dev
    pipe = T.Compose([dict(type='LoadImageFromFile'), dict(type='Resize', scale=(140, 60), keep_ratio=True),
                      dict(type='LoadAnnotations'), dict(type='PackSegInputs')])
    out = pipe.batch([dict(img=torch.from_numpy(img).to(dev), gt_seg_map=torch.from_numpy(seg).to(dev))])
    nw, nh = OA.rescale_size((70, 50), (140, 60))
    assert np.array_equal(out['inputs'][0].cpu().numpy(), OA.resize_bilinear_u8(img, (nw, nh)).transpose(2, 0, 1))
    assert out['data_samples'][0].metainfo['ori_shape'] == (50, 70)
    assert out['data_samples'][0].metainfo['scale_factor'] == (nw / 70, nh / 50)
    # identity pipeline: bytes in = bytes out
    pipe = T.Compose([dict(type='PackSegInputs')])
    out = pipe.batch([dict(img=torch.from_numpy(img).to(dev), gt_seg_map=torch.from_numpy(seg).to(dev))])
    assert np.array_equal(out['inputs'][0].cpu().numpy(), img.transpose(2, 0, 1))
    assert np.array_equal(out['data_samples'][0].gt_sem_seg.data.cpu().numpy()[0], seg.astype(np.int64))


def test_rejects_bad_inputs(be):
    from led_net_amd import transforms as T
    from led_net_amd._lib import LednError
    pipe = T.Compose(PIPE)
    with pytest.raises(LednError):
        pipe.batch([dict(img=torch.zeros(8, 8, 3, dtype=torch.float32, device=be.dev), gt_seg_map=None)])
    with pytest.raises(LednError):
        pipe.batch([dict(img=torch.zeros(8, 8, 3, dtype=torch.uint8, device=be.dev),
                         gt_seg_map=torch.zeros(8, 9, dtype=torch.uint8, device=be.dev))])
    with pytest.raises(KeyError):
        T.Compose([dict(type='RandomMosaic', prob=1.0)])


def test_augmented_batch_feeds_the_train_step(be):
    """Compose.batch -> SegDataPreProcessor -> EncoderDecoder.loss: the shapes / metainfo the hot path expects"""
    import os
    import led_net_amd as L
    from led_net_amd import transforms as T
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = L.load_config(os.path.join(root, 'tests', 'data', 'lednet_test_config.py'))
    for c in cfg['model']['decode_head']['loss_decode']:
        c['min_kept'] = 2000
    cfg['model']['data_preprocessor'] = dict(type='SegDataPreProcessor', mean=[123.675, 116.28, 103.53],
                                             std=[58.395, 57.12, 57.375], bgr_to_rgb=True, size=(320, 320),
                                             pad_val=0, seg_pad_val=255)
    torch.manual_seed(3)
    model = L.MODELS.build(cfg['model']).to(be.dev).train()
    pipe = T.Compose([dict(type='RandomResize', scale=(640, 320), ratio_range=(0.5, 2.0), keep_ratio=True),
                      dict(type='RandomCrop', crop_size=(320, 320), cat_max_ratio=0.75),
                      dict(type='RandomFlip', prob=0.5), dict(type='PhotoMetricDistortion'),
                      dict(type='PackSegInputs')])
    np.random.seed(4)
    recs = []
    for img, seg in _samples(9, 2, sizes=[(300, 500), (360, 640)], classes=2):
        recs.append(dict(img=torch.from_numpy(img).to(be.dev), gt_seg_map=torch.from_numpy(seg).to(be.dev)))
    out = pipe.batch(recs)
    data = model.data_preprocessor(dict(inputs=out['inputs'], data_samples=out['data_samples']), training=True)
    assert tuple(data['inputs'].shape) == (2, 3, 320, 320) and data['inputs'].dtype == torch.uint8
    losses = model(data['inputs'], data['data_samples'], mode='loss')
    assert all(torch.isfinite(v).all() for v in losses.values())


def test_reference_dataset_pipelines_build():
    """the train / test pipelines of configs/_base_/datasets/pascal_voc12.py:6-26 (values mirrored here: the GPU box
    has no /root/reference) build into the GPU Compose with the reference's type names and arguments"""
    from led_net_amd import transforms as T
    train_pipeline = [
        dict(type='LoadImageFromFile'), dict(type='LoadAnnotations'),
        dict(type='RandomResize', scale=(2048, 512), ratio_range=(0.5, 2.0), keep_ratio=True),
        dict(type='RandomCrop', crop_size=(512, 512), cat_max_ratio=0.75),
        dict(type='RandomFlip', prob=0.5), dict(type='PhotoMetricDistortion'), dict(type='PackSegInputs')]
    test_pipeline = [dict(type='LoadImageFromFile'), dict(type='Resize', scale=(2048, 512), keep_ratio=True),
                     dict(type='LoadAnnotations'), dict(type='PackSegInputs')]
    names = [type(t).__name__ for t in T.Compose(train_pipeline).transforms]
    assert names == ['RandomResize', 'RandomCrop', 'RandomFlip', 'PhotoMetricDistortion', 'PackSegInputs']
    assert [type(t).__name__ for t in T.Compose(test_pipeline).transforms] == ['Resize', 'PackSegInputs']
    ref = '/root/reference/configs/LED_Net/LEDNet_80k_cityscapes-1024x1024.py'
    import os
    if os.path.exists(ref):            # in the build container: the reference's own file, unchanged
        import led_net_amd as L
        cfg = L.load_config(ref)
        assert [c['type'] for c in cfg['train_pipeline']] == [c['type'] for c in train_pipeline]
        T.Compose(cfg['train_pipeline'])
        T.Compose(cfg['test_pipeline'])
