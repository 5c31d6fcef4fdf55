"""A tiny COCO-format dataset on disk for the pull_item tests (images = golden JPEG files, annotations = every form
pycocotools' annToRLE accepts).  Test infrastructure."""
import json
import os

import numpy as np

from oracle import coco_rle

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, 'golden', 'jpeg.npz'))

IMAGES = [  # (id, golden case, file name)
    (139, 'pil_50x35_420_base', '000000000139.jpg'),
    (285, 'pil_33x47_444_prog', 'COCO_val2014_000000000285.jpg'),        # 2014-style name: data/coco.py:131-134 strips the prefix
    (632, 'pil_exif6', '000000000632.jpg'),                             # stored 37x53, decodes to 53x37 (EXIF rotate)
    (724, 'pil_64x64_gray_base', '000000000724.jpg'),
    (785, 'pil_17x3_422_rst', '000000000785.jpg'),                      # has no annotations
]


def write_dataset(root):
    os.makedirs(root, exist_ok=True)
    images, anns = [], []
    for iid, case, fname in IMAGES:
        h, w = GOLD['bgr_' + case].shape[:2]
        with open(os.path.join(root, fname.split('_')[-1]), 'wb') as f:
            f.write(GOLD['jpg_' + case].tobytes())
        images.append({'id': iid, 'file_name': fname, 'height': int(h), 'width': int(w)})
    aid = [0]

    def ann(iid, cat, bbox, seg, crowd=0, **kw):
        aid[0] += 1
        d = {'id': aid[0], 'image_id': iid, 'category_id': cat, 'bbox': bbox, 'segmentation': seg, 'iscrowd': crowd, 'area': 1.0}
        d.update(kw)
        anns.append(d)

    # image 139 (50 rows x 35 cols): crowd FIRST in file order (must move to the end), two polygons, a multi-part polygon,
    # a sliver that Resize discards (narrower than 4/550 of the image)
    crowd_mask = np.zeros((50, 35), dtype=np.uint8)
    crowd_mask[5:20, 3:30] = 1
    crowd_mask[30:40, 10:12] = 1
    ann(139, 1, [3, 5, 27, 35], {'size': [50, 35], 'counts': [int(c) for c in coco_rle.rle_encode_counts(crowd_mask)]}, crowd=1)
    ann(139, 18, [2.5, 3.25, 20.0, 30.5], [[2.5, 3.25, 22.5, 4.0, 20.25, 33.75, 4.0, 30.0]])
    ann(139, 64, [10, 10, 15, 25], [[10, 10, 25, 10, 25, 20, 10, 20], [12.5, 24.5, 24.0, 26.0, 18.0, 35.0]])
    ann(139, 90, [1, 1, 0.1, 40], [[1, 1, 1.1, 1, 1.1, 41, 1, 41]])
    # image 285 (33 x 47): compressed RLE string, polygon touching the borders, repeated vertex
    m = np.zeros((33, 47), dtype=np.uint8)
    m[2:30, 40:47] = 1
    m[10:12, 0:5] = 1
    ann(285, 2, [0, 2, 47, 28], {'size': [33, 47], 'counts': coco_rle.rle_to_string(coco_rle.rle_encode_counts(m))})
    ann(285, 3, [0, 0, 47, 33], [[0, 0, 47, 0, 47, 33, 0, 33]])
    ann(285, 44, [5, 5, 30, 20], [[5, 5, 35, 5, 35, 5, 35, 25, 20.5, 30.25, 5, 25]])
    # image 632 (decodes to 53 x 37 after the EXIF rotation): only crowds
    m = np.zeros((53, 37), dtype=np.uint8)
    m[::2, 1::3] = 1
    ann(632, 1, [0, 0, 37, 53], {'size': [53, 37], 'counts': [int(c) for c in coco_rle.rle_encode_counts(m)]}, crowd=1)
    # image 724 (gray 64 x 64): one polygon partly outside the image
    ann(724, 7, [-5, 20, 40, 50], [[-5, 20, 35, 22, 30, 70, -3, 66]])
    # image 785 gets none
    with open(os.path.join(root, 'instances.json'), 'w') as f:
        json.dump({'images': images, 'annotations': anns, 'categories': [{'id': i} for i in range(1, 91)]}, f)
    return os.path.join(root, 'instances.json')
