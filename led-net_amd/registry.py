"""Plugin registry mirroring ``mmseg.registry.MODELS`` (registry/registry.py:56,
models/builder.py:6-10): ``MODELS.build(dict(type='LEDNet', ...))``.

mmengine/mmseg are not required.  When they ARE importable,
``register_into_mmseg()`` additionally registers the same classes into
``mmseg.registry.MODELS`` under the reference's names so that
``configs/LED_Net/*.py`` + ``tools/train.py`` resolve ``type='LEDNet'`` /
``type='LEDHead'`` / ``type='OhemCrossEntropy'`` to this implementation.
"""


class Registry:
    def __init__(self, name):
        self.name = name
        self._modules = {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            key = name or cls.__name__
            if key in self._modules and not force:
                raise KeyError(f'{key} is already registered in {self.name}')
            self._modules[key] = cls
            return cls
        if module is not None:
            return deco(module)
        return deco

    def get(self, key):
        return self._modules.get(key)

    def build(self, cfg, **default_args):
        if not isinstance(cfg, dict) or 'type' not in cfg:
            raise TypeError(f'cfg must be a dict with a "type" key, got {cfg!r}')
        args = dict(cfg)
        typ = args.pop('type')
        cls = typ if isinstance(typ, type) else self._modules.get(typ)
        if cls is None:
            raise KeyError(f'{typ} is not in the {self.name} registry')
        for k, v in default_args.items():
            args.setdefault(k, v)
        return cls(**args)

    def __contains__(self, key):
        return key in self._modules


MODELS = Registry('model')
BACKBONES = HEADS = LOSSES = SEGMENTORS = MODELS   # mmseg/models/builder.py:6-10


def register_into_mmseg(force=True):
    """Register LEDNet / LEDHead / OhemCrossEntropy into mmseg's own registry.
    Returns False (and does nothing) when mmseg/mmengine are not installed."""
    try:
        from mmseg.registry import MODELS as MM
    except Exception:
        return False
    for key in ('LEDNet', 'LEDHead', 'OhemCrossEntropy'):
        MM.register_module(name=key, force=force, module=MODELS.get(key))
    return True
