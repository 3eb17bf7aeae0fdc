"""Index over a COCO-format annotation file: the handful of `pycocotools.coco.COCO` queries the reference's datasets
make (mmdet/datasets/coco.py:46-81, coco_pose.py:32-72), plus run-length mask decoding for the extreme-point tool.
pycocotools is not part of this stack; the orderings below are the ones its API yields, because dataset indices, group
flags and therefore sampler output depend on them:

  * images in file order (`getImgIds()` = keys of the id->image dict in insertion order);
  * annotations of an image in file order (`getAnnIds(imgIds=[i])`);
  * category ids in file order, filtered by name (`getCatIds(catNms=...)`)."""
import json
from collections import defaultdict

import numpy as np


class CocoIndex:

    def __init__(self, annotation_file=None, dataset=None):
        if dataset is None:
            with open(annotation_file, 'r') as f:
                dataset = json.load(f)
        assert isinstance(dataset, dict), f'annotation file format {type(dataset)} not supported'
        self.dataset = dataset
        self.anns, self.imgs, self.cats = {}, {}, {}
        self.img_to_anns, self.cat_img_map = defaultdict(list), defaultdict(list)
        for ann in dataset.get('annotations', []):
            self.img_to_anns[ann['image_id']].append(ann)
            self.anns[ann['id']] = ann
        for img in dataset.get('images', []):
            self.imgs[img['id']] = img
        for cat in dataset.get('categories', []):
            self.cats[cat['id']] = cat
        if 'categories' in dataset:
            for ann in dataset.get('annotations', []):
                self.cat_img_map[ann['category_id']].append(ann['image_id'])

    def get_cat_ids(self, cat_names=()):
        # a bare string is matched by substring (`name in 'person'`), exactly what the COCO api does with it
        cats = self.dataset.get('categories', [])
        if len(cat_names):
            cats = [c for c in cats if c['name'] in cat_names]
        return [c['id'] for c in cats]

    def get_img_ids(self):
        return list(self.imgs.keys())

    def get_ann_ids(self, img_ids=()):
        if isinstance(img_ids, int):
            img_ids = [img_ids]
        if len(img_ids) == 0:
            return [a['id'] for a in self.dataset.get('annotations', [])]
        return [a['id'] for i in img_ids if i in self.img_to_anns for a in self.img_to_anns[i]]

    def load_anns(self, ids):
        return [self.anns[i] for i in ids]

    def load_imgs(self, ids):
        return [self.imgs[i] for i in ids]

    def load_cats(self, ids):
        return [self.cats[i] for i in ids]

    def ann_to_mask(self, ann):
        """Binary mask (h, w) of an annotation whose segmentation is run-length encoded (crowd regions)."""
        img = self.imgs[ann['image_id']]
        seg = ann['segmentation']
        if isinstance(seg, list):
            raise ValueError('polygon segmentations are consumed as polygons on the LSNet path; no rasteriser here')
        return rle_decode(seg, img['height'], img['width'])


def rle_counts_from_string(s):
    """COCO's compressed run-length string -> run lengths: 5 payload bits + continuation bit per character
    (offset 48), sign-extended, delta-coded against the run two places back from the third run on."""
    if isinstance(s, str):
        s = s.encode('ascii')
    counts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = s[p] - 48
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


def rle_decode(rle, height=None, width=None):
    """{'size': [h, w], 'counts': list | str} -> uint8 (h, w); runs alternate 0/1 starting with 0, column-major."""
    h, w = rle.get('size', (height, width))
    counts = rle['counts']
    if not isinstance(counts, (list, tuple)):
        counts = rle_counts_from_string(counts)
    flat = np.zeros(h * w, dtype=np.uint8)
    pos, val = 0, 0
    for c in counts:
        if val:
            flat[pos:pos + c] = 1
        pos += c
        val ^= 1
    return flat.reshape(w, h).T.copy()
