"""north_star: "outputs matching the reference CPU forward within stated fp tolerance (argmax masks bit-exact)".

f32 activations (GPU): whole-network predict at the config-A shape (512 x 512) against the CPU oracle: fused logits
within 1e-4 absolute and the argmax mask EQUAL to the oracle's wherever the oracle's class margin exceeds 1e-5 -- in
practice everywhere: measured on the MI355X (tools/measure_parity.py, r02w) max |logit error| 2.3e-5 and 0 flips out
of 3 x 524 288 pixels.  bf16 activations: the stated tolerance is in tests/test_bf16.py.
Config E (4 x 3 x 1024 x 2048 inference): the batch-split-invariance property at full size."""
import os

import pytest
import torch

from oracle import spec
from test_blocks import _randomize

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, 'tests', 'data', 'lednet_test_config.py')


@pytest.mark.parametrize('seed,batch', [(1, 1), (2, 2)])      # BASELINE config A is 1 x 3 x 512 x 512
def test_f32_argmax_masks_bit_exact_config_a(seed, batch):
    import led_net_amd as L
    torch.manual_seed(304)
    cfg = L.load_config(CFG)
    model = L.MODELS.build(cfg['model']).eval()
    _randomize(model, seed)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.to('cuda:0')
    g = torch.Generator().manual_seed(seed)
    img = torch.randint(0, 256, (batch, 3, 512, 512), dtype=torch.uint8, generator=g)
    with torch.no_grad():
        want, want_mask = spec.predict(spec.preprocess(img), sd)
        out = model(img.cuda(), mode='predict')
    logits = torch.stack([o.seg_logits.data for o in out]).cpu()
    mask = torch.cat([o.pred_sem_seg.data for o in out]).long().cpu()
    err = (logits - want).abs().max().item()
    margin = (want[:, 0] - want[:, 1]).abs()
    flips = mask != want_mask
    inside = int((margin <= 1e-5).sum())
    print(f'f32 predict {batch}x3x512x512 seed {seed}: max |logit err| {err:.3e}; argmax flips {int(flips.sum())} of {mask.numel()} '
          f'({int((flips & (margin > 1e-5)).sum())} outside the 1e-5 margin band, {inside} pixels inside it)')
    assert err < 1e-4, err
    assert torch.equal(mask[margin > 1e-5], want_mask[margin > 1e-5])
    assert int(flips.sum()) <= inside


def test_inference_batch_split_invariance_config_e():
    """BASELINE config E: 4 x 3 x 1024 x 2048 bf16 inference (the upsample / fusion stress shape): each image's fused
    logits and argmax mask are bit-identical whether computed in the batch or alone, and run to run."""
    import led_net_amd as L
    torch.manual_seed(304)
    cfg = L.load_config(CFG)
    m = L.MODELS.build(cfg['model'])
    m.set_act_dtype(torch.bfloat16)
    m.to('cuda:0').eval()
    g = torch.Generator().manual_seed(304)
    img = torch.randint(0, 256, (4, 3, 1024, 2048), dtype=torch.uint8, generator=g).cuda()
    with torch.no_grad():
        lg_b, mask_b = m.decode_head.predict_with_mask(m.extract_feat(img))
        assert tuple(lg_b.shape) == (4, 2, 1024, 2048) and tuple(mask_b.shape) == (4, 1024, 2048)
        for i in (0, 3):
            lg_1, mask_1 = m.decode_head.predict_with_mask(m.extract_feat(img[i:i + 1]))
            assert torch.equal(lg_1[0], lg_b[i]), (i, float((lg_1[0] - lg_b[i]).abs().max()))
            assert torch.equal(mask_1[0], mask_b[i]), i
        lg_2, mask_2 = m.decode_head.predict_with_mask(m.extract_feat(img))
        assert torch.equal(lg_2, lg_b) and torch.equal(mask_2, mask_b)
    assert torch.isfinite(lg_b).all() and 0 < int(mask_b.sum()) < mask_b.numel()
