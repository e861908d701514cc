"""Weight initialisation used by the reference's factory (model/networks.py:45-57,110-112):
orthogonal (gain 1) for every Conv / Linear weight, zero biases."""
import torch
from torch import nn


def init_weights_orthogonal(net, seed=None):
    if seed is not None:
        torch.manual_seed(seed)
    for m in net.modules():
        if isinstance(m, (nn.Conv2d, nn.Linear)):
            nn.init.orthogonal_(m.weight.data, gain=1)
            if m.bias is not None:
                m.bias.data.zero_()
    return net
