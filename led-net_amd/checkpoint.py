"""Checkpoint interop: mmengine-format ``.pth`` files
(``{'state_dict': ..., 'meta': {...}}``, mmengine.runner.checkpoint.load_checkpoint as used by
mmseg/apis/inference.py:59-75) in and out of the HIP model.  The parameter names are the
reference's (``backbone.*``, ``decode_head.{head,aux_head,head_x1,head_x2}.{0.bn,0.conv,1}.*``,
``decode_head.conv_seg.*`` ...), weights stay f32 OIHW, so no repacking is needed on load: the bf16 MFMA
packs and BatchNorm folds are derived caches and are dropped on load."""
import warnings

import torch


def _strip(sd, prefix):
    return {(k[len(prefix):] if k.startswith(prefix) else k): v for k, v in sd.items()}


def load_checkpoint(model, filename, map_location='cpu', strict=False):
    """-> the checkpoint dict (with 'meta'); tolerates a bare state_dict and DDP's 'module.' prefix
    (mmengine strips it the same way).  strict=False reports missing/unexpected keys as a warning,
    like mmengine's load_state_dict."""
    ckpt = torch.load(filename, map_location=map_location, weights_only=False)
    if not isinstance(ckpt, dict):
        raise RuntimeError(f'No state_dict found in checkpoint file {filename}')
    sd = ckpt.get('state_dict', ckpt)
    if 'state_dict' not in ckpt:
        ckpt = {'state_dict': sd, 'meta': {}}
    sd = _strip(sd, 'module.')
    res = model.load_state_dict(sd, strict=strict)
    if not strict and (res.missing_keys or res.unexpected_keys):
        warnings.warn(f'checkpoint {filename}: missing keys {res.missing_keys[:8]}'
                      f'{"..." if len(res.missing_keys) > 8 else ""}, unexpected keys {res.unexpected_keys[:8]}'
                      f'{"..." if len(res.unexpected_keys) > 8 else ""}')
    for m in model.modules():          # derived caches (folded BN, bf16 weight packs) belong to the old weights
        cache = getattr(m, '_cache', None)
        if isinstance(cache, dict):
            cache.clear()
    ckpt.setdefault('meta', {})
    return ckpt


def save_checkpoint(model, filename, meta=None, trainer=None):
    """mmengine layout: {'meta': ..., 'state_dict': cpu tensors} and, with a Trainer, the keys mmengine's
    CheckpointHook adds and Runner.resume reads: 'optimizer' (torch.optim.SGD.state_dict(): the momentum buffers,
    indexed by position in model.parameters()), 'param_schedulers' (the PolyLR position), 'message_hub'
    (runtime_info iter / epoch / max_iters) and meta['epoch' | 'iter' | 'seed' | 'experiment_name'].  The layout is
    restated from mmengine's documented checkpoint format (mmengine is not installed here: no file written by
    the reference stack pins it)."""
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ckpt = {'meta': dict(meta or {}), 'state_dict': sd}
    ckpt['meta'].setdefault('epoch', 0)
    if trainer is not None:
        ckpt['optimizer'] = trainer.optimizer_state_dict()
        ckpt['param_schedulers'] = [trainer.scheduler_state_dict()]
        ckpt['meta'].setdefault('iter', trainer.iter)
        ckpt['meta'].setdefault('seed', 304)
        ckpt['meta'].setdefault('experiment_name', 'led_net_amd')
        ckpt['message_hub'] = {'log_scalars': {}, 'resumed_keys': {'epoch': True, 'iter': True, 'max_iters': True},
                               'runtime_info': {'epoch': 0, 'iter': int(ckpt['meta']['iter']),
                                                'max_iters': int(trainer.max_iters), 'max_epochs': 1}}
    torch.save(ckpt, filename)


def resume(trainer, ckpt):
    """restore the optimizer / schedule state of a checkpoint dict (as returned by load_checkpoint) into a
    Trainer; weights-only checkpoints restart with zero momentum (and say so)."""
    if 'optimizer' in ckpt:
        trainer.load_optimizer_state_dict(ckpt['optimizer'], ckpt.get('param_schedulers'), ckpt.get('meta', {}).get('iter'))
    else:
        warnings.warn('checkpoint has no optimizer state: resuming the weights only (momentum restarts at zero)')
        trainer.iter = int(ckpt.get('meta', {}).get('iter', trainer.iter))
    return trainer


def init_model(config, checkpoint=None, device='cuda:0'):
    """mmseg.apis.init_model (apis/inference.py:25-90) on the built-in registry: build cfg.model, load the
    checkpoint, attach dataset_meta (mmseg 1.x 'dataset_meta', < 1.x 'CLASSES'/'PALETTE'), eval mode."""
    from .config import load_config
    from .registry import MODELS
    cfg = load_config(config) if isinstance(config, str) else config
    mcfg = dict(cfg['model'])
    mcfg.pop('pretrained', None)
    mcfg['train_cfg'] = None
    model = MODELS.build(mcfg)
    if checkpoint is not None:
        ck = load_checkpoint(model, checkpoint, map_location='cpu')
        meta = ck.get('meta', {})
        if 'dataset_meta' in meta:
            model.dataset_meta = meta['dataset_meta']
        elif 'CLASSES' in meta:
            model.dataset_meta = {'classes': meta['CLASSES'], 'palette': meta.get('PALETTE')}
        else:
            warnings.warn('dataset_meta or class names are not saved in the checkpoint\'s meta data')
            model.dataset_meta = {'classes': tuple(str(i) for i in range(model.decode_head.num_classes)),
                                  'palette': None}
    model.cfg = cfg
    model.to(device)
    model.eval()
    return model
