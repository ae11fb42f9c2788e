"""Minimal loader for mmengine-style python configs (``_base_`` inheritance),
enough to parse ``configs/LED_Net/LEDNet_80k_cityscapes-1024x1024.py`` unchanged
when mmengine is absent (SURVEY.md section 5 "Config / flags").
Reference behaviour: mmengine.Config.fromfile (third-party; tools/train.py:62-67)."""
import os


def _merge(base, new):
    out = dict(base)
    for k, v in new.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.get('_delete_', False):
            out[k] = _merge(out[k], v)
        else:
            if isinstance(v, dict):
                v = {a: b for a, b in v.items() if a != '_delete_'}
            out[k] = v
    return out


def load_config(path):
    path = os.path.abspath(path)
    ns = {}
    with open(path) as f:
        exec(compile(f.read(), path, 'exec'), ns)
    cfg = {}
    bases = ns.get('_base_', [])
    if isinstance(bases, str):
        bases = [bases]
    for b in bases:
        cfg = _merge(cfg, load_config(os.path.join(os.path.dirname(path), b)))
    own = {k: v for k, v in ns.items()
           if not k.startswith('_') and not callable(v) and type(v).__name__ != 'module'}
    return _merge(cfg, own)
