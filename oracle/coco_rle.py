"""CPU restatement of the COCO result wire format used by the reference's evaluator -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(yolact_amd/) never does.

What the reference does (eval.py:300-340, `Detections`):
  add_bbox  : (x1,y1,x2,y2) -> [x, y, w, h], each rounded to one decimal; category through get_coco_cat (eval.py:283-286,
              data/config.py COCO_LABEL_MAP); score as float.
  add_mask  : pycocotools.mask.encode(np.asfortranarray(mask.astype(np.uint8))) and counts.decode('ascii').
  dump      : json.dump of the two lists.

`pycocotools` is a third-party dependency that is NOT vendored in /root/reference (requirement "pycocotools" of the
reference's README / environment.yml, cocoapi 2.0, common/maskApi.c).  Its published algorithm is restated here:
  rleEncode   (maskApi.c rleEncode)   : run lengths of the column-major (Fortran order) mask, starting with a run of zeros
                                        (possibly of length 0).
  rleToString (maskApi.c rleToString) : every count x (for i > 2 the difference to counts[i-2]) is written as a
                                        little-endian sequence of 5-bit groups, bit 0x20 = "more groups follow", sign
                                        extended like LEB128, each group + 48 -> ASCII '0'..'o'.
  rleFrString (maskApi.c rleFrString) : the inverse.
Pinning: the reference ships real outputs of this exact code path -- web/dets/*.json were written by
`eval.py --output_web_json` (eval.py:342-371) with the official weights and contain the RLE strings pycocotools
produced.  oracle/make_golden_rle.py samples them into tests/golden/rle_web.json and tests/test_coco_rle.py checks that
this restatement decodes every string to a mask of the declared size that lies inside the detection's own box, and
re-encodes it to the identical string.
"""
import numpy as np


def rle_encode_counts(mask):
    """mask [h,w] (any dtype; nonzero = 1, like .astype(np.uint8) of the reference's {0,1} float masks) -> list of counts."""
    m = (np.asarray(mask) != 0).astype(np.uint8)
    flat = m.flatten(order='F')
    a = flat.size
    if a == 0:
        return [0]
    change = np.flatnonzero(flat[1:] != flat[:-1]) + 1          # positions where a new run starts
    starts = np.concatenate(([0], change))
    counts = np.diff(np.concatenate((starts, [a]))).tolist()
    if flat[0] != 0:                                            # first run is by definition a run of zeros
        counts = [0] + counts
    return counts


def rle_to_string(counts):
    out = bytearray()
    for i, c in enumerate(counts):
        x = int(c)
        if i > 2:
            x -= int(counts[i - 2])
        more = True
        while more:
            ch = x & 0x1f
            x >>= 5                                             # arithmetic shift (Python ints), like C `long`
            more = (x != -1) if (ch & 0x10) else (x != 0)
            if more:
                ch |= 0x20
            out.append(ch + 48)
    return out.decode('ascii')


def rle_from_string(s):
    counts = []
    p, n = 0, len(s)
    b = s.encode('ascii') if isinstance(s, str) else bytes(s)
    while p < n:
        x, k, more = 0, 0, True
        while more:
            c = b[p] - 48
            x |= (c & 0x1f) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    return counts


def rle_decode(counts, h, w):
    flat = np.zeros(h * w, dtype=np.uint8)
    pos, v = 0, 0
    for c in counts:
        if v:
            flat[pos:pos + c] = 1
        pos += c
        v ^= 1
    assert pos == h * w, (pos, h, w)
    return flat.reshape((h, w), order='F')


def encode(mask):
    """pycocotools.mask.encode + the .decode('ascii') of eval.py:323: {'size': [h, w], 'counts': str}."""
    h, w = mask.shape
    return {'size': [int(h), int(w)], 'counts': rle_to_string(rle_encode_counts(mask))}


def bbox_record(image_id, category_id, bbox, score, label_map=None):
    """Detections.add_bbox (eval.py:306-318)."""
    bb = [bbox[0], bbox[1], bbox[2] - bbox[0], bbox[3] - bbox[1]]
    bb = [round(float(x) * 10) / 10 for x in bb]
    cat = int(category_id)
    return {'image_id': int(image_id), 'category_id': (label_map[cat] if label_map is not None else cat),
            'bbox': bb, 'score': float(score)}
