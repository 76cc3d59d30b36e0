"""Weight initialisation hook (reference: networks/initialize.py:59-69).

Upstream's init functions only touch nn.Conv3d / nn.BatchNorm3d / nn.Linear, none of which
exist in these 2-D networks, so `init_weights(net, 'kaiming')` leaves the PyTorch default
Conv2d initialisation in place.  That behaviour is kept (parity of the initial weights).
"""
import torch.nn as nn
from torch.nn import init

__all__ = ['init_weights']


def _make(conv_fn, linear_fn):
    def fn(m):
        if type(m) == nn.Conv3d:
            conv_fn(m.weight.data)
        elif type(m) == nn.BatchNorm3d:
            init.normal_(m.weight.data, 1.0, 0.02)
            init.constant_(m.bias.data, 0.0)
        elif type(m) == nn.Linear:
            linear_fn(m.weight.data)
    return fn


_INITS = {
    'normal': _make(lambda w: init.normal_(w, 0.0, 0.02), lambda w: init.normal_(w, 0.0, 0.02)),
    'xavier': _make(lambda w: init.xavier_normal_(w, gain=1), lambda w: init.xavier_normal_(w, gain=1)),
    'kaiming': _make(lambda w: init.kaiming_normal_(w, a=0, mode='fan_in'),
                     lambda w: init.kaiming_normal_(w, a=0, mode='fan_in')),
    'orthogonal': _make(lambda w: init.orthogonal_(w, gain=1), lambda w: init.orthogonal_(w, gain=1)),
}


def init_weights(net, init_type='kaiming'):
    if init_type not in _INITS:
        raise NotImplementedError('initialization method [%s] is not implemented' % init_type)
    net.apply(_INITS[init_type])
