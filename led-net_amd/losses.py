"""OhemCrossEntropy (registered under the reference's name) and accuracy.

Mirrors mmseg/models/losses/ohem_cross_entropy_loss.py:11-94 (constructor
arguments, ``loss_name`` property, selection semantics) and
losses/accuracy.py:6-60 (top-1, ignore_index).  The arithmetic runs in the
fused HIP kernels of csrc/attn_loss_opt.hip (ohem_*: softmax prob + CE, exact k-th
smallest by radix select, masked mean, backward).
"""
import torch.nn as nn

from .registry import MODELS


@MODELS.register_module()
class OhemCrossEntropy(nn.Module):
    def __init__(self, ignore_label=255, thres=0.7, min_kept=100000, loss_weight=1.0,
                 class_weight=None, loss_name='loss_ohem'):
        super().__init__()
        if class_weight is not None:
            raise NotImplementedError('class_weight is not used by the LED-Net config (cfg :44,50)')
        self.thresh = thres
        self.min_kept = max(1, min_kept)
        self.ignore_label = ignore_label
        self.loss_weight = loss_weight
        self.loss_name_ = loss_name

    def forward(self, score, target):
        """score: N x C x H x W (any float layout), target: N x H x W int64."""
        from .train import ohem_loss
        return ohem_loss(self, score, target)

    @property
    def loss_name(self):
        return self.loss_name_


def build_loss(cfg):
    return MODELS.build(cfg)
