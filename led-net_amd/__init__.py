"""led-net_amd: MI355X-native LED-Net forward/backward hot path.

Public surface = the reference's plugin names:
``MODELS.build(dict(type='LEDNet'|'LEDHead'|'OhemCrossEntropy'|'EncoderDecoder', ...))``.
Compute is in csrc/libledn_hip.so (C ABI: include/ledn.h); there is no CPU path.
"""
from . import _lib, ops  # noqa: F401
from ._lib import is_deterministic, set_deterministic  # noqa: F401
from .registry import BACKBONES, HEADS, LOSSES, MODELS, SEGMENTORS, register_into_mmseg  # noqa: F401
from .lednet import LEDNet
from .led_head import LEDHead
from .losses import OhemCrossEntropy  # noqa: F401
from .segmentor import EncoderDecoder, SegDataPreProcessor, SegDataSample  # noqa: F401
from .config import load_config  # noqa: F401
from .train import Trainer  # noqa: F401
from .metrics import IoUMetric  # noqa: F401
from .checkpoint import init_model, load_checkpoint, resume, save_checkpoint  # noqa: F401
from . import transforms  # noqa: F401

MODELS.register_module(module=LEDNet)
MODELS.register_module(module=LEDHead)

__all__ = ['MODELS', 'LEDNet', 'LEDHead', 'OhemCrossEntropy', 'EncoderDecoder', 'SegDataSample',
           'load_config', 'ops', 'register_into_mmseg', 'Trainer']
