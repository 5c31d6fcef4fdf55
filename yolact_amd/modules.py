"""Parameter containers with the reference's state-dict layout.

These nn.Modules only HOLD parameters under the reference's names so that `Yolact.load_weights()` accepts
the reference checkpoints unchanged (backbone.conv1/bn1, backbone.layers.{s}.{i}.{conv,bn}{1,2,3}
[.conv_offset_mask], ...downsample.{0,1}, fpn.{lat,pred,downsample}_layers.{i}, proto_net.{0,2,4,8,10},
prediction_layers.0.{upfeature.0,bbox_layer,conf_layer,mask_layer}, semantic_seg_conv,
maskiou_net.maskiou_net.{0,2,..,10}).  They never compute: the arithmetic is done by the HIP engine
(yolact_amd/engine.py), which reads these tensors once, folds BatchNorm and re-lays the filters.
"""
from __future__ import annotations

import torch
import torch.nn as nn


class _NoCompute(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError('yolact_amd parameter containers do not compute; call Yolact.forward (HIP engine)')


class InterpolateModule(_NoCompute):
    """Placeholder for layers/interpolate.py:4-17 inside proto_net (owns no parameters)."""

    def __init__(self, scale_factor=2):
        super().__init__()
        self.scale_factor = scale_factor


# external/DCNv2/dcn_v2.py:97-128.  Unlike the other containers this one computes when called on its own (the reference's DCNv2
# known-answer test does that): yolact_amd/dcn_v2.py runs the HIP kernels behind the reference's module API.
from .dcn_v2 import DCN          # noqa: E402,F401


class Bottleneck(_NoCompute):
    """backbone.py:13-35 layout (stride on the 3x3)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, use_dcn=False):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        if use_dcn:
            self.conv2 = DCN(planes, planes, 3, stride=stride, padding=1)
        else:
            self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample
        self.stride = stride
        self.use_dcn = use_dcn


class ResNetBackbone(_NoCompute):
    """backbone.py:60-124 layout: conv1/bn1 + layers[4] of Bottlenecks; dcn placement per _make_layer."""

    def __init__(self, layers, dcn_layers=(0, 0, 0, 0), dcn_interval=1):
        super().__init__()
        self.layers = nn.ModuleList()
        self.channels = []
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        for i, (planes, n) in enumerate(zip((64, 128, 256, 512), layers)):
            self._make_layer(planes, n, 1 if i == 0 else 2, dcn_layers[i], dcn_interval)
        self.backbone_modules = [m for m in self.modules() if isinstance(m, nn.Conv2d)]

    def _make_layer(self, planes, blocks, stride, dcn_layers, dcn_interval):
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                                       nn.BatchNorm2d(planes * 4))
        mods = [Bottleneck(self.inplanes, planes, stride, downsample, use_dcn=(dcn_layers >= blocks))]
        self.inplanes = planes * 4
        for i in range(1, blocks):
            use_dcn = ((i + dcn_layers) >= blocks) and (i % dcn_interval == 0)
            mods.append(Bottleneck(self.inplanes, planes, use_dcn=use_dcn))
        self.channels.append(planes * 4)
        self.layers.append(nn.Sequential(*mods))


def _dark_unit(cin, cout, k, stride=1, padding=0):
    return nn.Sequential(nn.Conv2d(cin, cout, k, stride=stride, padding=padding, bias=False),
                         nn.BatchNorm2d(cout), nn.LeakyReLU(0.1, inplace=True))


class DarkNetBlock(_NoCompute):
    def __init__(self, in_channels, channels):
        super().__init__()
        self.conv1 = _dark_unit(in_channels, channels, 1)
        self.conv2 = _dark_unit(channels, channels * 2, 3, padding=1)


class DarkNetBackbone(_NoCompute):
    """backbone.py:239-294 layout: _preconv + 5 stages of (stride-2 unit, n residual blocks)."""

    def __init__(self, layers=(1, 2, 8, 8, 4)):
        super().__init__()
        self.layers = nn.ModuleList()
        self.channels = []
        self._preconv = _dark_unit(3, 32, 3, padding=1)
        cin = 32
        for ch, n in zip((32, 64, 128, 256, 512), layers):
            mods = [_dark_unit(cin, ch * 2, 3, stride=2, padding=1)]
            cin = ch * 2
            mods += [DarkNetBlock(cin, ch) for _ in range(n)]
            self.channels.append(cin)
            self.layers.append(nn.Sequential(*mods))
        self.backbone_modules = [m for m in self.modules() if isinstance(m, nn.Conv2d)]


class FPN(_NoCompute):
    """yolact.py:281-308 layout (lat/pred layers stored deepest level first)."""

    def __init__(self, in_channels, num_features=256, num_downsample=2, pad=True):
        super().__init__()
        self.lat_layers = nn.ModuleList([nn.Conv2d(c, num_features, 1) for c in reversed(in_channels)])
        self.pred_layers = nn.ModuleList([nn.Conv2d(num_features, num_features, 3, padding=1 if pad else 0)
                                          for _ in in_channels])
        self.downsample_layers = nn.ModuleList([nn.Conv2d(num_features, num_features, 3, padding=1, stride=2)
                                                for _ in range(num_downsample)])
        self.num_downsample = num_downsample


def make_net(in_channels, conf, include_last_relu=True):
    """Container twin of utils/functions.py:163-213: same Sequential indices (layer, ReLU, layer, ReLU, ...)."""
    mods = []
    for ch, k, kw in conf:
        if ch is None:
            mods.append(InterpolateModule(scale_factor=-k))
        elif k > 0:
            mods.append(nn.Conv2d(in_channels, ch, k, **kw))
            in_channels = ch
        else:
            raise NotImplementedError('ConvTranspose2d layers are not used by any shipped YOLACT config')
        mods.append(nn.ReLU(inplace=True))
    if not include_last_relu:
        mods = mods[:-1]
    return nn.Sequential(*mods), in_channels


class PredictionModule(_NoCompute):
    """yolact.py:47-131 layout; only the first head owns parameters (share_prediction_module)."""

    def __init__(self, in_channels, num_priors, num_classes, mask_dim, extra_head_net, head_layer_params,
                 aspect_ratios, scales, parent=None, index=0):
        super().__init__()
        self.num_classes, self.mask_dim, self.num_priors = num_classes, mask_dim, num_priors
        self.parent = [parent]
        self.index = index
        self.aspect_ratios, self.scales = aspect_ratios, scales
        self.priors = None
        self.last_conv_size = None
        if parent is None:
            out_channels = in_channels
            if extra_head_net is not None:
                self.upfeature, out_channels = make_net(in_channels, extra_head_net)
            self.bbox_layer = nn.Conv2d(out_channels, num_priors * 4, **head_layer_params)
            self.conf_layer = nn.Conv2d(out_channels, num_priors * num_classes, **head_layer_params)
            self.mask_layer = nn.Conv2d(out_channels, num_priors * mask_dim, **head_layer_params)


class FastMaskIoUNet(_NoCompute):
    """yolact.py:363-369 layout."""

    def __init__(self, maskiou_net_cfg, num_classes):
        super().__init__()
        self.maskiou_net, _ = make_net(1, list(maskiou_net_cfg) + [(num_classes - 1, 1, {})], include_last_relu=True)
