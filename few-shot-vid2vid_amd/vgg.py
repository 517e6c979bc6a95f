"""VGG19 perceptual loss of the G step (reference models/networks/loss.py:107-128, models/networks/vgg.py:45-59) on
the HIP convolution kernels: 13 conv3x3+ReLU (fused epilogue) and 4 max-pools up to relu5_1, L1 between the five
tapped activations of the generated and the real image.  The weights are frozen, so only data gradients flow.

torchvision's pretrained weights cannot be fetched here; `random_vgg19_weights` is the deterministic He-initialised
stand-in shared with the oracle and the reference shims (throughput and parity do not depend on the values).  A real
checkpoint loads through `load_state_dict` (keys `features.<idx>.weight/bias`, torchvision's numbering).
"""
import math

import torch
import torch.nn as nn

from . import ops
from .conv import ACT_RELU

VGG19_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']
TAPS = [1, 6, 11, 20, 29]                 # relu1_1, relu2_1, relu3_1, relu4_1, relu5_1 (loss.py:110)
TAP_WEIGHTS = [1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0]


def random_vgg19_weights(seed=19):
    """[(weight, bias)] of the 16 VGG19 convolutions, He-normal from one seeded generator (conv order)."""
    g = torch.Generator().manual_seed(seed)
    out, cin = [], 3
    for v in VGG19_CFG:
        if v == 'M':
            continue
        w = torch.randn((v, cin, 3, 3), generator=g) * math.sqrt(2.0 / (9 * cin))
        out.append((w, torch.zeros(v)))
        cin = v
    return out


class VGGActivations(nn.Module):
    def __init__(self, seed=19):
        super().__init__()
        layers, cin, idx = {}, 3, 0
        weights = iter(random_vgg19_weights(seed))
        self.plan = []                        # ('conv', idx) | ('pool', idx); ReLU is fused into the conv epilogue
        for v in VGG19_CFG:
            if idx > TAPS[-1]:
                break
            if v == 'M':
                self.plan.append(('pool', idx)); idx += 1
            else:
                w, b = next(weights)
                conv = nn.Module()
                conv.weight = nn.Parameter(w, requires_grad=False)
                conv.bias = nn.Parameter(b, requires_grad=False)
                layers[str(idx)] = conv
                self.plan.append(('conv', idx)); idx += 2
                cin = v
        self.features = nn.ModuleDict(layers)

    def forward(self, x):
        res = []
        for kind, idx in self.plan:
            if kind == 'pool':
                x = ops.maxpool2(x)
            else:
                m = self.features[str(idx)]
                x = ops.conv2d(x, m.weight, m.bias, stride=1, padding=1, act=ACT_RELU)
                if idx + 1 in TAPS:
                    res.append(x)
        return res


class VGGLoss(nn.Module):
    def __init__(self):
        super().__init__()
        self.vgg = VGGActivations()

    def forward(self, x, y):
        if x.dim() == 5:
            x, y = x.reshape(-1, *x.shape[-3:]), y.reshape(-1, *y.shape[-3:])
        with torch.no_grad():
            y_feats = self.vgg(y)
        x_feats = self.vgg(x)
        loss = 0
        for w, a, b in zip(TAP_WEIGHTS, x_feats, y_feats):
            loss = loss + w * ops.l1_loss(a, b.detach())
        return loss
