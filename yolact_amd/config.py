"""Configuration for the YOLACT inference hot path.

Only the fields that `Yolact.forward`, `Detect` and `postprocess` READ are modelled
here (reference: data/config.py:61-100 `Config`, :417-648 `coco_base_config`,
:656-806 shipped model configs, :810-825 `cfg`/`set_cfg`).  Training, dataset and
augmentation fields are out of scope for this tier.

Two ways to get a config:
  * stand-alone: `set_cfg('yolact_resnet50_config')` on this module's global `cfg`;
  * drop-in: if the reference's own `data.config` module is importable (eval.py's
    sys.path), `active_cfg()` returns ITS global `cfg`, so flags mutated by eval.py
    (`cfg.mask_proto_debug`, `cfg.rescore_bbox`, ...) are seen at call time exactly
    like the reference's forward passes see them.
"""
from __future__ import annotations

import sys
from math import sqrt


class Cfg(dict):
    """Attribute dict with `copy(overrides)` / `replace(other)` like data/config.py:61-100."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:  # pragma: no cover
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def copy(self, new=None):
        out = Cfg(self)
        if new:
            out.update(new)
        return out

    def replace(self, other):
        self.clear()
        self.update(other)


# activation / mask-type tags (reference uses enum-ish objects + lambdas; we use strings
# and accept either when reading a reference cfg, see `act_name`).
MASK_DIRECT, MASK_LINCOMB = 0, 1

_fpn = Cfg(num_features=256, interpolation_mode='bilinear', num_downsample=2,
           use_conv_downsample=True, pad=True, relu_downsample_layers=False,
           relu_pred_layers=True)

_base_scales = [[24], [48], [96], [192], [384]]
_plus_scales = [[s * 2 ** (j / 3.0) for j in range(3)] for s in (24, 48, 96, 192, 384)]
_ars = [[[1, 1 / 2, 2]]] * 5


def _backbone(name, kind, args, selected, scales, square, transform='resnet'):
    return Cfg(name=name, kind=kind, args=args, selected_layers=selected,
               pred_scales=scales, pred_aspect_ratios=_ars, use_pixel_scales=True,
               preapply_sqrt=False, use_square_anchors=square, transform=transform)


_common = Cfg(
    num_classes=81, max_num_detections=100, nms_top_k=200, nms_conf_thresh=0.05,
    nms_thresh=0.5, eval_mask_branch=True, mask_type=MASK_LINCOMB, mask_proto_src=0,
    # (channels, kernel, kwargs) conv | (None, -2, {}) bilinear x2  (utils/functions.py:163-213)
    mask_proto_net=[(256, 3, {'padding': 1})] * 3 + [(None, -2, {}), (256, 3, {'padding': 1}), (32, 1, {})],
    mask_proto_prototype_activation='relu', mask_proto_mask_activation='sigmoid',
    mask_proto_coeff_activation='tanh', mask_proto_bias=False, mask_proto_use_grid=False,
    mask_proto_prototypes_as_features=False, mask_proto_split_prototypes_by_head=False,
    mask_proto_coeff_gate=False, mask_proto_debug=False,
    share_prediction_module=True, extra_head_net=[(256, 3, {'padding': 1})],
    extra_layers=(0, 0, 0), head_layer_params={'kernel_size': 3, 'padding': 1},
    use_prediction_module=False, use_yolo_regressors=False, use_mask_scoring=False,
    use_instance_coeff=False, use_focal_loss=False, use_sigmoid_focal_loss=False,
    use_objectness_score=False, use_class_existence_loss=False,
    use_semantic_segmentation_loss=True, freeze_bn=False,
    use_maskiou=False, maskiou_net=[], rescore_mask=False, rescore_bbox=False,
    fpn=_fpn, max_size=550,
)

_R50 = ([3, 4, 6, 3],)
_R101 = ([3, 4, 23, 3],)

CONFIGS = {
    'yolact_base_config': _common.copy(dict(
        name='yolact_base',
        backbone=_backbone('ResNet101', 'resnet', _R101, [1, 2, 3], _base_scales, True))),
    'yolact_resnet50_config': _common.copy(dict(
        name='yolact_resnet50',
        backbone=_backbone('ResNet50', 'resnet', _R50, [1, 2, 3], _base_scales, True))),
    'yolact_darknet53_config': _common.copy(dict(
        name='yolact_darknet53',
        backbone=_backbone('DarkNet53', 'darknet', ([1, 2, 8, 8, 4],), [2, 3, 4], _base_scales, True,
                           transform='darknet'))),
    'yolact_im400_config': _common.copy(dict(
        name='yolact_im400', max_size=400,
        backbone=_backbone('ResNet101', 'resnet', _R101, [1, 2, 3],
                           [[int(s[0] / 550 * 400)] for s in _base_scales], True))),
    'yolact_im700_config': _common.copy(dict(
        name='yolact_im700', max_size=700,
        backbone=_backbone('ResNet101', 'resnet', _R101, [1, 2, 3],
                           [[int(s[0] / 550 * 700)] for s in _base_scales], True))),
}
_plus = dict(use_maskiou=True, rescore_mask=True, rescore_bbox=False,
             maskiou_net=[(8, 3, {'stride': 2}), (16, 3, {'stride': 2}), (32, 3, {'stride': 2}),
                          (64, 3, {'stride': 2}), (128, 3, {'stride': 2})])
CONFIGS['yolact_plus_base_config'] = _common.copy(dict(
    name='yolact_plus_base',
    backbone=_backbone('ResNet101_DCN_Interval3', 'resnet', ([3, 4, 23, 3], [0, 4, 23, 3], 3),
                       [1, 2, 3], _plus_scales, False), **_plus))
CONFIGS['yolact_plus_resnet50_config'] = _common.copy(dict(
    name='yolact_plus_resnet50',
    backbone=_backbone('ResNet50_DCNv2', 'resnet', ([3, 4, 6, 3], [0, 4, 6, 3]),
                       [1, 2, 3], _plus_scales, False), **_plus))

cfg = CONFIGS['yolact_base_config'].copy()


def set_cfg(config_name: str):
    """Select a shipped config by the reference's name (data/config.py:812-821)."""
    if config_name not in CONFIGS:
        raise KeyError('unknown config %r (have: %s)' % (config_name, ', '.join(sorted(CONFIGS))))
    cfg.replace(CONFIGS[config_name].copy())
    ref = _reference_cfg_module()
    if ref is not None:  # keep the reference's global in step when running under eval.py
        ref.set_cfg(config_name)


def _reference_cfg_module():
    m = sys.modules.get('data.config')
    return m if (m is not None and hasattr(m, 'cfg') and hasattr(m, 'set_cfg')) else None


def active_cfg():
    """The config object the hot path must read *at call time* (SURVEY §5 'config / flags')."""
    ref = _reference_cfg_module()
    return ref.cfg if ref is not None else cfg


# ---- adapters so one code path reads either our Cfg or the reference's Config ------------

def act_name(v):
    """'relu' | 'sigmoid' | 'tanh' | 'none' from our tag or the reference's lambda
    (data/config.py activation_func: tanh/sigmoid/softmax/relu/none)."""
    if isinstance(v, str):
        return v
    import torch
    t = torch.tensor([-1.0, 0.0, 2.0])
    y = v(t)
    for name, f in (('relu', torch.relu), ('sigmoid', torch.sigmoid), ('tanh', torch.tanh),
                    ('none', lambda z: z)):
        if y.shape == t.shape and torch.equal(y, f(t)):
            return name
    raise ValueError('unsupported activation in cfg')


def is_lincomb(c):
    mt = c.mask_type
    return mt == MASK_LINCOMB  # reference: mask_type.lincomb == 1 (data/config.py:368-385)


def backbone_kind(bb):
    """'resnet' | 'darknet' for our Cfg (kind) or the reference's (type = class)."""
    k = bb.get('kind') if isinstance(bb, dict) and 'kind' in bb else None
    if k:
        return k
    n = getattr(bb.type, '__name__', '')
    if n == 'ResNetBackbone':
        return 'resnet'
    if n == 'DarkNetBackbone':
        return 'darknet'
    raise NotImplementedError('backbone %s is outside the hot-path scope (SURVEY §2)' % n)


def make_priors_host(conv_h, conv_w, scales, aspect_ratios, max_size, bb):
    """Anchor boxes [cx, cy, w, h] for one level; python doubles then fp32, cell-major
    (y outer, x inner), scale outer / ratio inner (reference yolact.py:224-246)."""
    out = []
    for j in range(conv_h):
        for i in range(conv_w):
            x = (i + 0.5) / conv_w
            y = (j + 0.5) / conv_h
            for ars in aspect_ratios:
                for scale in scales:
                    for ar in ars:
                        if not bb.preapply_sqrt:
                            ar = sqrt(ar)
                        if bb.use_pixel_scales:
                            w = scale * ar / max_size
                            h = scale / ar / max_size
                        else:
                            w = scale * ar / conv_w
                            h = scale / ar / conv_h
                        if bb.use_square_anchors:
                            h = w
                        out += [x, y, w, h]
    return out
