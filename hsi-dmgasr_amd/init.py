"""Weight initialisation used by the reference's factory (model/networks.py:45-57,110-112):
orthogonal (gain 1) for every Conv / Linear weight, zero biases."""
import torch
from torch import nn


def init_weights_orthogonal(net, seed=None):
    if seed is not None:
        torch.manual_seed(seed)
    # written through the parameters themselves (not `.data`): the in-place ops bump `Parameter._version`, which is what
    # the modules' packed-weight caches key on (sr3_modules/unet.py:_PackCache)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.orthogonal_(m.weight, gain=1)
                if m.bias is not None:
                    m.bias.zero_()
    return net
