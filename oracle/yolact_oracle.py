"""CPU oracle for the YOLACT inference hot path.  *** TEST INFRASTRUCTURE ONLY ***

A restatement, in plain torch-CPU functional ops, of the algorithm the reference runs for
    Yolact.forward (eval)  ->  Detect  ->  postprocess
so that the HIP path can be checked on any box (the reference itself is not available on the GPU box).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product path (yolact_amd/) never does and fails loudly without its HIP library.

PINNING: this oracle is checked against outputs of the reference itself, executed in the build container
by oracle/make_golden.py (shimmed import of /root/reference) and committed under
tests/golden/ (tests/test_oracle_golden.py).  The only reference-shipped known-answer test on this path
is DCNv2's zero-offset identity (external/DCNv2/test.py:32-67), restated in tests/test_dcn_oracle.py.

Dense arithmetic (conv2d, batch_norm, interpolate, softmax, sort) is torch's own CPU implementation —
the same third-party library the reference calls (PyTorch, unpinned ">=1.0.1", environment.yml:18); this
file restates everything the reference *itself* writes around those calls.  Each function cites the
reference lines it follows.  Ties in sorts: the reference's torch.sort is unstable (order of exact ties is
arbitrary); the oracle — like the HIP path — defines lowest index first (stable sort).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ----------------------------------------------------------------------------------------------
# building blocks
def _conv(x: Tensor, sd: Dict[str, Tensor], key: str, stride: int = 1, padding: int = 0) -> Tensor:
    return F.conv2d(x, sd[key + '.weight'], sd.get(key + '.bias'), stride=stride, padding=padding)


def _bn(x: Tensor, sd: Dict[str, Tensor], key: str) -> Tensor:
    # nn.BatchNorm2d in eval mode, eps 1e-5 (backbone.py:17-33 uses the default norm layer)
    return F.batch_norm(x, sd[key + '.running_mean'], sd[key + '.running_var'], sd[key + '.weight'],
                        sd[key + '.bias'], False, 0.0, 1e-5)


def dcn_v2_forward(x: Tensor, offset: Tensor, mask: Tensor, weight: Tensor, bias: Tensor,
                   stride: int, padding: int, dilation: int = 1) -> Tensor:
    """Modulated deformable conv, one deformable group.  The reference has NO CPU implementation
    (external/DCNv2/src/cpu/dcn_v2_cpu.cpp:23); this follows the CUDA path:
      sampling grid + validity test  dcn_v2_im2col_cuda.cu:155-156,177-188
      zero-padded bilinear           dcn_v2_im2col_cuda.cu:25-54
      column order c*9 + i*3 + j     dcn_v2_im2col_cuda.cu:150,159,191  (== weight.view(Co,-1))
      bias + GEMM                    dcn_v2_cuda.cu:123-163 ;  output size  dcn_v2_cuda.cu:86-87
    x [B,C,H,W], offset [B,2*kh*kw,Ho,Wo] (ch 2k = dh, 2k+1 = dw), mask [B,kh*kw,Ho,Wo]."""
    B, C, H, W = x.shape
    Co, _, kh, kw = weight.shape
    Ho = (H + 2 * padding - (dilation * (kh - 1) + 1)) // stride + 1
    Wo = (W + 2 * padding - (dilation * (kw - 1) + 1)) // stride + 1
    oy = torch.arange(Ho, dtype=torch.float32).view(1, Ho, 1) * stride - padding
    ox = torch.arange(Wo, dtype=torch.float32).view(1, 1, Wo) * stride - padding
    cols = x.new_zeros(B, C, kh * kw, Ho, Wo)
    xf = x.reshape(B, C, H * W)
    for i in range(kh):
        for j in range(kw):
            k = i * kw + j
            h = (oy + i * dilation) + offset[:, 2 * k]          # [B,Ho,Wo]
            w = (ox + j * dilation) + offset[:, 2 * k + 1]
            valid = (h > -1) & (w > -1) & (h < H) & (w < W)
            hl = torch.floor(h)
            wl = torch.floor(w)
            lh, lw = h - hl, w - wl
            hh_, hw_ = 1 - lh, 1 - lw
            hl, wl = hl.long(), wl.long()
            hh, wh = hl + 1, wl + 1

            def tap(yy, xx, ok):
                ok = ok & valid
                idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).view(B, 1, Ho * Wo).expand(B, C, Ho * Wo)
                v = torch.gather(xf, 2, idx).view(B, C, Ho, Wo)
                return v * ok.view(B, 1, Ho, Wo).to(v.dtype)

            v1 = tap(hl, wl, (hl >= 0) & (wl >= 0))
            v2 = tap(hl, wh, (hl >= 0) & (wh <= W - 1))
            v3 = tap(hh, wl, (hh <= H - 1) & (wl >= 0))
            v4 = tap(hh, wh, (hh <= H - 1) & (wh <= W - 1))
            w1, w2, w3, w4 = (hh_ * hw_).unsqueeze(1), (hh_ * lw).unsqueeze(1), (lh * hw_).unsqueeze(1), (lh * lw).unsqueeze(1)
            val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4
            cols[:, :, k] = val * mask[:, k].unsqueeze(1)
    out = torch.einsum('ok,bkn->bon', weight.reshape(Co, C * kh * kw), cols.reshape(B, C * kh * kw, Ho * Wo))
    return out.view(B, Co, Ho, Wo) + bias.view(1, Co, 1, 1)


def _dcn_module(x: Tensor, sd: Dict[str, Tensor], key: str, stride: int) -> Tensor:
    # DCN.forward, external/DCNv2/dcn_v2.py:118-128
    om = _conv(x, sd, key + '.conv_offset_mask', stride=stride, padding=1)
    offset, mask = om[:, :18], torch.sigmoid(om[:, 18:])
    return dcn_v2_forward(x, offset, mask, sd[key + '.weight'], sd[key + '.bias'], stride, 1, 1)


# ----------------------------------------------------------------------------------------------
# backbones
def resnet_backbone(x: Tensor, sd, blocks: List[int], dcn_layers=(0, 0, 0, 0), dcn_interval: int = 1,
                    pre: str = 'backbone') -> List[Tensor]:
    """ResNetBackbone.forward + Bottleneck.forward (backbone.py:126-139, :37-57; _make_layer :92-124)."""
    x = F.relu(_bn(_conv(x, sd, pre + '.conv1', 2, 3), sd, pre + '.bn1'))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    outs = []
    for li, nb in enumerate(blocks):
        for bi in range(nb):
            p = '%s.layers.%d.%d' % (pre, li, bi)
            stride = 2 if (li > 0 and bi == 0) else 1
            if bi == 0:
                use_dcn = dcn_layers[li] >= nb
            else:
                use_dcn = (bi + dcn_layers[li]) >= nb and (bi % dcn_interval == 0)
            o = F.relu(_bn(_conv(x, sd, p + '.conv1'), sd, p + '.bn1'))
            if use_dcn:
                o = _dcn_module(o, sd, p + '.conv2', stride)
            else:
                o = _conv(o, sd, p + '.conv2', stride, 1)
            o = F.relu(_bn(o, sd, p + '.bn2'))
            o = _bn(_conv(o, sd, p + '.conv3'), sd, p + '.bn3')
            res = x
            if (p + '.downsample.0.weight') in sd:
                res = _bn(_conv(x, sd, p + '.downsample.0', stride), sd, p + '.downsample.1')
            x = F.relu(o + res)
        outs.append(x)
    return outs


def darknet_backbone(x: Tensor, sd, blocks: List[int], pre: str = 'backbone') -> List[Tensor]:
    """DarkNetBackbone.forward (backbone.py:296-306); unit = conv(no bias) -> BN -> LeakyReLU(0.1) (:222-236)."""
    def unit(t, key, stride=1, padding=0):
        return F.leaky_relu(_bn(_conv(t, sd, key + '.0', stride, padding), sd, key + '.1'), 0.1)

    x = unit(x, pre + '._preconv', 1, 1)
    outs = []
    for li, nb in enumerate(blocks):
        p = '%s.layers.%d' % (pre, li)
        x = unit(x, p + '.0', 2, 1)
        for bi in range(nb):
            q = '%s.%d' % (p, bi + 1)
            x = unit(unit(x, q + '.conv1'), q + '.conv2', 1, 1) + x
        outs.append(x)
    return outs


# ----------------------------------------------------------------------------------------------
def fpn(convouts: List[Tensor], sd, num_downsample: int = 2) -> List[Tensor]:
    """FPN.forward (yolact.py:310-361): lat/pred layers are stored deepest-first; the top-down path carries the
    pre-pred sums; pred 3x3 + ReLU; extra levels by stride-2 3x3 convs on the (post-ReLU) last level."""
    n = len(convouts)
    sums: List[Optional[Tensor]] = [None] * n
    x = None
    for i in range(n):
        j = n - 1 - i
        lat = _conv(convouts[j], sd, 'fpn.lat_layers.%d' % i)
        if x is None:
            x = lat            # reference: zeros(1) + lat
        else:
            x = F.interpolate(x, size=convouts[j].shape[2:], mode='bilinear', align_corners=False) + lat
        sums[j] = x
    out = [None] * n
    for i in range(n):
        j = n - 1 - i
        out[j] = F.relu(_conv(sums[j], sd, 'fpn.pred_layers.%d' % i, 1, 1))
    for i in range(num_downsample):
        out.append(_conv(out[-1], sd, 'fpn.downsample_layers.%d' % i, 2, 1))
    return out


def make_net_forward(x: Tensor, sd, pre: str, conf, include_last_relu: bool = True) -> Tensor:
    """utils/functions.py:163-213: (ch, k>0, kw) conv | (None, -s, {}) bilinear x s; every layer followed by
    ReLU except (optionally) the last."""
    idx = 0
    for li, (ch, k, kw) in enumerate(conf):
        last = li == len(conf) - 1
        if ch is None:
            x = F.interpolate(x, scale_factor=-k, mode='bilinear', align_corners=False)
        elif k > 0:
            x = _conv(x, sd, '%s.%d' % (pre, idx), kw.get('stride', 1), kw.get('padding', 0))
        else:
            raise NotImplementedError('deconv layers are not used by any shipped config')
        if not (last and not include_last_relu):
            x = F.relu(x)
        idx += 2
    return x


def make_priors(conv_h: int, conv_w: int, scales, aspect_ratios, max_size: int, use_pixel_scales=True,
                preapply_sqrt=False, use_square_anchors=True) -> Tensor:
    """PredictionModule.make_priors (yolact.py:224-246): python doubles, then one fp32 cast."""
    data = []
    for j in range(conv_h):
        for i in range(conv_w):
            x = (i + 0.5) / conv_w
            y = (j + 0.5) / conv_h
            for ars in aspect_ratios:
                for scale in scales:
                    for ar in ars:
                        if not preapply_sqrt:
                            ar = math.sqrt(ar)
                        if use_pixel_scales:
                            w = scale * ar / max_size
                            h = scale / ar / max_size
                        else:
                            w = scale * ar / conv_w
                            h = scale / ar / conv_h
                        if use_square_anchors:
                            h = w
                        data += [x, y, w, h]
    return torch.tensor(data, dtype=torch.float32).view(-1, 4)


def forward_raw(x: Tensor, sd: Dict[str, Tensor], cfg) -> Dict[str, Tensor]:
    """Yolact.forward up to (and including) the eval-mode softmax (yolact.py:564-674). `cfg` is a
    yolact_amd.config Cfg (plain data).  Returns loc/conf/mask/priors/proto plus the stage tensors."""
    bb = cfg.backbone
    stages = {}
    if bb.kind == 'resnet':
        args = list(bb.args)
        blocks = args[0]
        dcn_layers = args[1] if len(args) > 1 else [0, 0, 0, 0]
        dcn_interval = args[2] if len(args) > 2 else 1
        outs = resnet_backbone(x, sd, blocks, dcn_layers, dcn_interval)
    else:
        outs = darknet_backbone(x, sd, bb.args[0])
    sel = [outs[i] for i in bb.selected_layers]
    for i, t in enumerate(sel):
        stages['C%d' % (i + 3)] = t
    feats = fpn(sel, sd, cfg.fpn.num_downsample)
    for i, t in enumerate(feats):
        stages['P%d' % (i + 3)] = t
    emb = bool(getattr(cfg, 'eval_mask_branch', True))       # eval.py:1067-1068: --detect sets cfg.eval_mask_branch = False
    proto = None
    if emb:                                                  # yolact.py:579-580: no prototypes without the mask branch
        proto = make_net_forward(feats[cfg.mask_proto_src], sd, 'proto_net', cfg.mask_proto_net, include_last_relu=False)
        proto = F.relu(proto)                                # cfg.mask_proto_prototype_activation
        proto = proto.permute(0, 2, 3, 1).contiguous()       # yolact.py:599
    A = sum(len(a) for a in bb.pred_aspect_ratios[0]) * len(bb.pred_scales[0])
    C, D = cfg.num_classes, int(sd['prediction_layers.0.mask_layer.weight'].shape[0]) // A
    locs, confs, masks, priors = [], [], [], []
    B = x.shape[0]
    for lvl, f in enumerate(feats):
        u = make_net_forward(f, sd, 'prediction_layers.0.upfeature', cfg.extra_head_net)
        locs.append(_conv(u, sd, 'prediction_layers.0.bbox_layer', 1, 1).permute(0, 2, 3, 1).reshape(B, -1, 4))
        confs.append(_conv(u, sd, 'prediction_layers.0.conf_layer', 1, 1).permute(0, 2, 3, 1).reshape(B, -1, C))
        if emb:
            masks.append(torch.tanh(_conv(u, sd, 'prediction_layers.0.mask_layer', 1, 1).permute(0, 2, 3, 1).reshape(B, -1, D)))
        else:                                                # yolact.py:172-175: zero coefficients, no activation (:189)
            masks.append(torch.zeros(B, locs[-1].shape[1], D))
        priors.append(make_priors(f.shape[2], f.shape[3], bb.pred_scales[lvl], bb.pred_aspect_ratios[lvl],
                                  cfg.max_size, bb.use_pixel_scales, bb.preapply_sqrt, bb.use_square_anchors))
    out = dict(loc=torch.cat(locs, 1), conf_logits=torch.cat(confs, 1), mask=torch.cat(masks, 1),
               priors=torch.cat(priors, 0), proto=proto)
    out['conf'] = F.softmax(out['conf_logits'], -1)          # yolact.py:674
    out['stages'] = stages
    return out


# ----------------------------------------------------------------------------------------------
# Detect
def decode(loc: Tensor, priors: Tensor) -> Tensor:
    """box_utils.py:304-310, same evaluation order."""
    xy = priors[:, :2] + loc[:, :2] * 0.1 * priors[:, 2:]
    wh = priors[:, 2:] * torch.exp(loc[:, 2:] * 0.2)
    x1y1 = xy - wh / 2
    return torch.cat([x1y1, wh + x1y1], 1)


def jaccard(a: Tensor, b: Tensor) -> Tensor:
    """box_utils.py:33-80 for batched [n,A,4] x [n,B,4]."""
    mx = torch.min(a[:, :, None, 2:], b[:, None, :, 2:])
    mn = torch.max(a[:, :, None, :2], b[:, None, :, :2])
    wh = torch.clamp(mx - mn, min=0)
    inter = wh[..., 0] * wh[..., 1]
    area_a = ((a[:, :, 2] - a[:, :, 0]) * (a[:, :, 3] - a[:, :, 1]))[:, :, None]
    area_b = ((b[:, :, 2] - b[:, :, 0]) * (b[:, :, 3] - b[:, :, 1]))[:, None, :]
    return inter / (area_a + area_b - inter)


def detect_image(conf: Tensor, loc: Tensor, mask: Tensor, priors: Tensor, conf_thresh=0.05, nms_thresh=0.5,
                 top_k=200, max_det=100, cross_class=False):
    """Detect.detect + fast_nms / cc_fast_nms for one image (detection.py:81-180).
    conf [P,C] post-softmax.  Returns None or dict(box, mask, class, score, prior)."""
    boxes_all = decode(loc, priors)
    cur = conf[:, 1:].t().contiguous()                  # [C-1, P]
    conf_scores, _ = cur.max(0)
    keep = conf_scores > conf_thresh
    kidx = torch.nonzero(keep).squeeze(1)
    if kidx.numel() == 0:
        return None
    scores = cur[:, keep]
    boxes = boxes_all[keep]
    masks = mask[keep]
    if cross_class:
        sc, classes = scores.max(0)
        _, idx = sc.sort(dim=0, descending=True, stable=True)
        idx = idx[:top_k]
        bsel = boxes[idx]
        iou = jaccard(bsel[None], bsel[None])[0].triu_(diagonal=1)
        iou_max, _ = iou.max(0)
        out = idx[iou_max <= nms_thresh]
        return dict(box=boxes[out], mask=masks[out], score=sc[out], prior=kidx[out])  | {'class': classes[out]}
    sc, idx = scores.sort(dim=1, descending=True, stable=True)
    idx = idx[:, :top_k].contiguous()
    sc = sc[:, :top_k]
    ncls, nd = idx.shape
    b = boxes[idx.view(-1)].view(ncls, nd, 4)
    m = masks[idx.view(-1)].view(ncls, nd, -1)
    pri = kidx[idx.view(-1)].view(ncls, nd)
    iou = jaccard(b, b).triu_(diagonal=1)
    iou_max, _ = iou.max(1)
    kp = iou_max <= nms_thresh
    classes = torch.arange(ncls)[:, None].expand_as(kp)[kp]
    b, m, s, pri = b[kp], m[kp], sc[kp], pri[kp]
    s, order = s.sort(dim=0, descending=True, stable=True)
    order = order[:max_det]
    return {'box': b[order], 'mask': m[order], 'class': classes[order], 'score': s[:max_det], 'prior': pri[order]}


def detect(raw: Dict[str, Tensor], cfg, cross_class=False) -> List[Optional[Dict[str, Tensor]]]:
    """Detect.__call__ (detection.py:32-78) over the batch; attaches proto[b]."""
    out = []
    for b in range(raw['loc'].shape[0]):
        r = detect_image(raw['conf'][b], raw['loc'][b], raw['mask'][b], raw['priors'], cfg.nms_conf_thresh,
                         cfg.nms_thresh, cfg.nms_top_k, cfg.max_num_detections, cross_class)
        if r is not None and raw.get('proto') is not None:       # detection.py:73-74
            r['proto'] = raw['proto'][b]
        out.append(r)
    return out


# ----------------------------------------------------------------------------------------------
# postprocess
def sanitize(_x1: Tensor, _x2: Tensor, size: int, padding: int = 0):
    """box_utils.py:327-346 with cast=False."""
    a, b = _x1 * size, _x2 * size
    x1, x2 = torch.min(a, b), torch.max(a, b)
    return torch.clamp(x1 - padding, min=0), torch.clamp(x2 + padding, max=size)


def crop(masks: Tensor, boxes: Tensor, padding: int = 1) -> Tensor:
    """box_utils.py:349-373; masks [h,w,n]."""
    h, w, n = masks.shape
    x1, x2 = sanitize(boxes[:, 0], boxes[:, 2], w, padding)
    y1, y2 = sanitize(boxes[:, 1], boxes[:, 3], h, padding)
    cols = torch.arange(w, dtype=x1.dtype).view(1, -1, 1)
    rows = torch.arange(h, dtype=x1.dtype).view(-1, 1, 1)
    inside = (cols >= x1.view(1, 1, -1)) & (cols < x2.view(1, 1, -1)) & (rows >= y1.view(1, 1, -1)) & (rows < y2.view(1, 1, -1))
    return masks * inside.to(masks.dtype)


def maskiou_forward(masks: Tensor, sd, cfg) -> Tensor:
    """FastMaskIoUNet.forward (yolact.py:363-375): conv stack (config maskiou_net + 1x1 to 80), ReLU after every
    layer, global max-pool."""
    conf = list(cfg.maskiou_net) + [(cfg.num_classes - 1, 1, {})]
    x = make_net_forward(masks, sd, 'maskiou_net.maskiou_net', conf, include_last_relu=True)
    return F.max_pool2d(x, kernel_size=x.shape[2:]).squeeze(-1).squeeze(-1)


def postprocess(det: Optional[Dict[str, Tensor]], w: int, h: int, cfg, sd=None, crop_masks=True, score_threshold=0.0,
                return_soft=False):
    """output_utils.py:15-122 (lincomb branch).  Returns (classes i64, scores, boxes i64, masks f32{0,1});
    with return_soft also the pre-threshold upsampled masks (for margin-aware comparison)."""
    if det is None:
        return None
    det = dict(det)
    if score_threshold > 0:
        k = det['score'] > score_threshold
        for key in det:
            if key != 'proto':
                det[key] = det[key][k]
        if det['score'].shape[0] == 0:
            return None
    classes, boxes, scores, coef, proto = det['class'], det['box'].clone(), det['score'], det['mask'], det.get('proto')
    if not bool(getattr(cfg, 'eval_mask_branch', True)):     # output_utils.py:58,97-122: boxes only, `masks` stays the coefficient rows
        x1, x2 = sanitize(boxes[:, 0], boxes[:, 2], w)
        y1, y2 = sanitize(boxes[:, 1], boxes[:, 3], h)
        return classes, scores, torch.stack([x1, y1, x2, y2], 1).long(), coef
    masks = torch.sigmoid(proto @ coef.t())
    if crop_masks:
        masks = crop(masks, boxes)
    masks = masks.permute(2, 0, 1).contiguous()
    if cfg.use_maskiou and sd is not None:
        miou = maskiou_forward(masks.unsqueeze(1), sd, cfg)
        miou = torch.gather(miou, 1, classes.unsqueeze(1)).squeeze(1)
        if cfg.rescore_mask:
            scores = scores * miou if cfg.rescore_bbox else [scores, scores * miou]
    soft = F.interpolate(masks.unsqueeze(0), (h, w), mode='bilinear', align_corners=False).squeeze(0)
    hard = (soft > 0.5).float()
    x1, x2 = sanitize(boxes[:, 0], boxes[:, 2], w)
    y1, y2 = sanitize(boxes[:, 1], boxes[:, 3], h)
    boxes_px = torch.stack([x1, y1, x2, y2], 1).long()
    if return_soft:
        return classes, scores, boxes_px, hard, soft
    return classes, scores, boxes_px, hard


# ----------------------------------------------------------------------------------------------
# FastBaseTransform — utils/augmentations.py:630-658
def fast_base_transform(img: Tensor, cfg, means=(103.94, 116.78, 123.68), std=(57.38, 57.12, 58.40)) -> Tensor:
    """img [n,h,w,3] float BGR -> [n,3,S,S] normalised RGB.  augmentations.py:637-642 size selection, :644-645 permute +
    bilinear (align_corners=False), :647-652 transform (data/config.py:181-202), :657 BGR->RGB.  MEANS/STD are in BGR
    order and applied before the swap (data/config.py:28-29)."""
    if getattr(cfg, 'preserve_aspect_ratio', False):
        _, h, w, _ = img.shape
        ratio = math.sqrt(w / h)                      # Resize.calc_size_preserve_ar, augmentations.py:133-138
        size = (int(cfg.max_size / ratio), int(cfg.max_size * ratio))
    else:
        size = (cfg.max_size, cfg.max_size)
    x = img.permute(0, 3, 1, 2).contiguous()
    x = F.interpolate(x, size, mode='bilinear', align_corners=False)
    mean = torch.tensor(means, dtype=torch.float32).view(1, 3, 1, 1)
    sd = torch.tensor(std, dtype=torch.float32).view(1, 3, 1, 1)
    tr = cfg.backbone.transform
    name = tr if isinstance(tr, str) else ('resnet' if tr.normalize else 'vgg' if tr.subtract_means else
                                           'darknet' if tr.to_float else 'none')
    if name == 'resnet':
        x = (x - mean) / sd
    elif name == 'vgg':
        x = x - mean
    elif name == 'darknet':
        x = x / 255
    return x[:, (2, 1, 0), :, :].contiguous()
