"""Deterministic synthetic weights and inputs (there are no checkpoints or datasets offline).

`synth_state_dict(shapes, seed)` fills ANY state-dict layout — the reference's or ours — key by key from a
per-key seeded generator, so the reference model (golden generation), the CPU oracle and the HIP path all
see bit-identical parameters without shipping a 125 MB file.  Recipe follows SURVEY §8(d): non-trivial
BatchNorm statistics (so BN folding is exercised), He-uniform convs, and a gain on the class-logit layer so
that softmax scores clear the 0.05 candidate threshold.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import torch


def _gen(key: str, seed: int) -> torch.Generator:
    g = torch.Generator(device='cpu')
    g.manual_seed((zlib.crc32(key.encode()) + 1000003 * int(seed)) & 0x7FFFFFFF)
    return g


def synth_state_dict(shapes: Iterable[Tuple[str, Tuple[int, ...]]], seed: int = 0, conf_gain: float = 0.04,
                     bg_bias: float = 0.0) -> Dict[str, torch.Tensor]:
    """`bg_bias` is added to the background logit's bias of every anchor (conf_layer.bias[a * C + 0]): with a large gain it
    gives the "pretrained-like" sparse regime of SURVEY 8(d) — about 1 % of the priors over the 0.05 candidate threshold and a
    handful of confident, well separated detections."""
    # every factory call names device='cpu': eval.py / train.py run under torch.set_default_tensor_type('torch.cuda.FloatTensor')
    # (eval.py:1077-1081), where a device-less torch.rand(generator=<cpu generator>) raises
    shapes = [(k, tuple(s)) for k, s in shapes]
    keys = {k for k, _ in shapes}
    out = {}
    for k, shp in shapes:
        g = _gen(k, seed)
        if k.endswith('num_batches_tracked'):
            out[k] = torch.zeros(shp, dtype=torch.long, device='cpu')
        elif k.endswith('running_var'):
            out[k] = torch.rand(shp, generator=g, device='cpu') + 0.5
        elif k.endswith('running_mean'):
            out[k] = torch.randn(shp, generator=g, device='cpu') * 0.1
        elif len(shp) == 1 and k.endswith('.weight') and (k[:-len('weight')] + 'running_mean') in keys:
            if k.endswith('bn3.weight') or k.endswith('conv2.1.weight'):
                # last BN of a residual branch: small gamma keeps activations O(1) through 16-33 blocks
                out[k] = torch.rand(shp, generator=g, device='cpu') * 0.2 + 0.1
            else:
                out[k] = torch.rand(shp, generator=g, device='cpu') + 0.5      # BN gamma
        elif len(shp) == 1 and k.endswith('.bias') and (k[:-len('bias')] + 'running_mean') in keys:
            out[k] = torch.randn(shp, generator=g, device='cpu') * 0.1          # BN beta
        elif len(shp) == 4:
            fan_in = shp[1] * shp[2] * shp[3]
            bound = (6.0 / fan_in) ** 0.5
            w = (torch.rand(shp, generator=g, device='cpu') * 2 - 1) * bound
            if 'conv_offset_mask' in k:
                w = torch.randn(shp, generator=g, device='cpu') * 0.02
            if 'conf_layer' in k:
                w = w * conf_gain
            elif 'bbox_layer' in k:
                w = w * 0.05      # keeps exp(loc.wh * 0.2) sane
            elif 'mask_layer' in k:
                w = w * 0.04      # keeps tanh out of saturation
            elif k == 'proto_net.10.weight':
                w = w * 0.2       # keeps sigmoid(proto @ coef) away from 0/1 so thresholds are exercised
            out[k] = w
        elif len(shp) == 1:
            std = 0.3 if 'conv_offset_mask' in k else 0.02
            out[k] = torch.randn(shp, generator=g, device='cpu') * std
            if bg_bias and k.endswith('conf_layer.bias'):
                ncls = 81 if shp[0] % 81 == 0 else shp[0]
                out[k][0::ncls] += bg_bias
        else:
            out[k] = torch.randn(shp, generator=g, device='cpu') * 0.02
    return out


def synth_images(B: int, H: int, W: int, seed: int = 1234) -> torch.Tensor:
    """Zero-mean unit-variance RGB planes = the post-BaseTransform distribution."""
    g = torch.Generator(device='cpu')
    g.manual_seed(int(seed))
    return torch.randn(B, 3, H, W, generator=g, device='cpu')


def plant_outlier_channels(sd, k_exp, nch=4):
    """Equivalent re-parametrisation of the network with OUTLIER CHANNELS (what real BN-folded checkpoints have): in
    stages C3 .. C5 `nch` channels of the stage tensor are scaled by 2^k (every block's bn3 and the projection shortcut's BN), and
    every consumer of that tensor (the conv1 of the following blocks, the next stage's first conv1 / projection shortcut, the FPN
    lateral) divides its weights for those input channels by 2^k; likewise four output channels of proto_net.0 against proto_net.2.  Powers of two commute with fp32 rounding (no overflow here), so in exact
    fp32 arithmetic every product — hence every head tensor — is unchanged; what changes is the dynamic range INSIDE the activation
    tensors (2^k between channels of one tensor: the fp16x2 tiles use ONE power-of-two scale per tensor) and inside the filter rows
    (2^-k between columns of one row: one scale per row)."""
    sd = {k: v.clone() for k, v in sd.items()}
    f = float(2 ** k_exp)
    planted = []

    def scale_out(bn, ch):
        sd[bn + '.weight'][ch] *= f
        sd[bn + '.bias'][ch] *= f

    def scale_in(conv_w, ch):
        sd[conv_w][:, ch] /= f
    nblocks = {}
    for key in sd:
        if key.startswith('backbone.layers.') and key.endswith('.bn3.weight'):
            _, _, li, bi = key.split('.')[:4]
            nblocks[int(li)] = max(nblocks.get(int(li), 0), int(bi) + 1)
    nstage = len(nblocks)
    for li in range(1, nstage):                       # C3, C4, C5 (stage outputs the FPN consumes; backbone.py:126-139, yolact.py:310-341)
        n = nblocks[li]
        C = sd['backbone.layers.%d.%d.bn3.weight' % (li, n - 1)].shape[0]
        ch = torch.arange(nch) * (C // nch) + 3
        # the stage tensor runs through identity shortcuts (y = relu(bn3(conv3) + x), backbone.py:50-55): the channel must be
        # scaled in EVERY block's bn3 and in the projection shortcut's BN, and every conv1 that reads the stage tensor divides
        for bi in range(n):
            scale_out('backbone.layers.%d.%d.bn3' % (li, bi), ch)
            if bi > 0:
                scale_in('backbone.layers.%d.%d.conv1.weight' % (li, bi), ch)
        scale_out('backbone.layers.%d.0.downsample.1' % li, ch)
        if li + 1 < nstage:
            scale_in('backbone.layers.%d.0.conv1.weight' % (li + 1), ch)
            scale_in('backbone.layers.%d.0.downsample.0.weight' % (li + 1), ch)
        scale_in('fpn.lat_layers.%d.weight' % (nstage - 1 - li), ch)     # lat_layers are stored top-down (yolact.py:286-289)
        planted.append(('C%d' % (li + 2), ch.tolist()))
    ch = torch.tensor([5, 70, 131, 200])
    sd['proto_net.0.weight'][ch] *= f
    sd['proto_net.0.bias'][ch] *= f
    sd['proto_net.2.weight'][:, ch] /= f
    planted.append(('proto_net.0', ch.tolist()))
    return sd, planted
