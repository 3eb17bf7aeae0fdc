"""COCO detection / instance-segmentation / keypoint metrics: AP and AR over IoU (or OKS) thresholds .50:.05:.95,
object sizes and detection budgets.  Same definitions and the same numbers as the reference's vendored evaluator
(cocoapi/pycocotools/pycocotools/cocoeval.py:9-609, driven by mmdet/datasets/coco.py:330-507), organised around
arrays instead of per-annotation dicts:

  evaluate()   per (category, image): IoU matrix of score-sorted detections x ground truths (liblsnet_host.so), then
               per area range the greedy matching for all thresholds in one native call (`lsn_coco_match`);
  accumulate() per (category, area, budget): detections of all images merged by score (stable), cumulative TP / FP,
               precision envelope sampled at 101 recall points;
  summarize()  the 12 (boxes, masks) or 10 (keypoints) headline numbers.

`load_results` is `COCO.loadRes` (cocoapi/.../coco.py:367-424): result records -> an index with ids, areas and
crowd flags filled in."""
import copy
import ctypes as C

import numpy as np

from ..data.coco_index import CocoIndex
from . import mask as mask_util

KPT_OKS_SIGMAS = np.array([.26, .25, .25, .35, .35, .79, .79, .72, .72, .62, .62, 1.07, 1.07, .87, .87, .89, .89]) / 10.0


class Params:
    """cocoeval.py:546-609"""

    def __init__(self, iou_type='segm'):
        if iou_type not in ('segm', 'bbox', 'keypoints'):
            raise Exception('iouType not supported')
        self.iou_type = iou_type
        self.img_ids, self.cat_ids = [], []
        self.iou_thrs = np.linspace(.5, 0.95, int(np.round((0.95 - .5) / .05)) + 1, endpoint=True)
        self.rec_thrs = np.linspace(.0, 1.00, int(np.round((1.00 - .0) / .01)) + 1, endpoint=True)
        self.use_cats = 1
        if iou_type == 'keypoints':
            self.max_dets = [20]
            self.area_rng = [[0 ** 2, 1e5 ** 2], [32 ** 2, 96 ** 2], [96 ** 2, 1e5 ** 2]]
            self.area_lbl = ['all', 'medium', 'large']
            self.kpt_oks_sigmas = KPT_OKS_SIGMAS
        else:
            self.max_dets = [1, 10, 100]
            self.area_rng = [[0 ** 2, 1e5 ** 2], [0 ** 2, 32 ** 2], [32 ** 2, 96 ** 2], [96 ** 2, 1e5 ** 2]]
            self.area_lbl = ['all', 'small', 'medium', 'large']


def load_results(gt_index, results):
    """Result records (list of dicts, or the path of a json file holding one) -> CocoIndex over the same images."""
    import json
    if isinstance(results, str):
        with open(results) as f:
            results = json.load(f)
    assert isinstance(results, list), 'results in not an array of objects'
    anns = results
    ids = set(a['image_id'] for a in anns)
    assert ids == ids & set(gt_index.get_img_ids()), 'Results do not correspond to current coco set'
    data = {'images': list(gt_index.dataset['images'])}
    first = anns[0]            # IndexError on empty results, as the reference (caught by CocoDataset.evaluate)
    if 'bbox' in first and not first['bbox'] == []:
        data['categories'] = copy.deepcopy(gt_index.dataset['categories'])
        for i, a in enumerate(anns):
            x, y, w, h = a['bbox']
            if 'segmentation' not in a:
                a['segmentation'] = [[x, y, x, y + h, x + w, y + h, x + w, y]]
            a['area'] = w * h
            a['id'] = i + 1
            a['iscrowd'] = 0
    elif 'segmentation' in first:
        data['categories'] = copy.deepcopy(gt_index.dataset['categories'])
        for i, a in enumerate(anns):
            a['area'] = mask_util.area(a['segmentation'])
            if 'bbox' not in a:
                a['bbox'] = mask_util.toBbox(a['segmentation'])
            a['id'] = i + 1
            a['iscrowd'] = 0
    elif 'keypoints' in first:
        data['categories'] = copy.deepcopy(gt_index.dataset['categories'])
        for i, a in enumerate(anns):
            xs, ys = a['keypoints'][0::3], a['keypoints'][1::3]
            x0, x1, y0, y1 = np.min(xs), np.max(xs), np.min(ys), np.max(ys)
            a['area'] = (x1 - x0) * (y1 - y0)
            a['id'] = i + 1
            a['bbox'] = [x0, y0, x1 - x0, y1 - y0]
    data['annotations'] = anns
    return CocoIndex(dataset=data)


def ann_to_rle(index, ann):
    """polygons / uncompressed RLE / RLE of one annotation -> RLE (coco.py:486-505)"""
    img = index.imgs[ann['image_id']]
    h, w = img['height'], img['width']
    seg = ann['segmentation']
    if isinstance(seg, list):
        return mask_util.merge(mask_util.frPyObjects(seg, h, w))
    if isinstance(seg['counts'], list):
        return mask_util.frPyObjects(seg, h, w)
    return seg


class CocoEval:

    def __init__(self, gt_index, dt_index, iou_type='segm'):
        self.gt, self.dt = gt_index, dt_index
        self.params = Params(iou_type)
        self.params.img_ids = sorted(gt_index.get_img_ids())
        self.params.cat_ids = sorted(gt_index.get_cat_ids())
        self.eval_imgs, self.eval, self.stats = [], {}, []

    # ------------------------------------------------------------------------------------------- per image
    def _collect(self, index, img_ids, cat_ids):
        """(image, category) -> annotations in file order (cocoeval.py:57-83: getAnnIds(imgIds, catIds))."""
        cats = set(cat_ids)
        table = {}
        for i in img_ids:
            for a in index.img_to_anns.get(i, ()):
                if not self.params.use_cats or a['category_id'] in cats:
                    table.setdefault((i, a['category_id']), []).append(a)
        return table

    def _similarity(self, gts, dts, rle):
        """(D, G) IoU / OKS of score-sorted, budget-truncated detections x ground truths."""
        p = self.params
        if p.iou_type == 'keypoints':
            return self._oks(gts, dts)
        crowd = [int(g['iscrowd']) for g in gts]
        if p.iou_type == 'segm':
            return mask_util.iou([rle[id(d)] for d in dts], [rle[id(g)] for g in gts], crowd)
        return mask_util.iou([d['bbox'] for d in dts], [g['bbox'] for g in gts], crowd)

    def _oks(self, gts, dts):
        """Object keypoint similarity (cocoeval.py:154-197): mean over labelled keypoints of
        exp(-d^2 / (2 area (2 sigma_k)^2)); for an unlabelled ground truth the distance to its doubled box."""
        if len(gts) == 0 or len(dts) == 0:
            return []
        var = (self.params.kpt_oks_sigmas * 2) ** 2
        d = np.array([dt['keypoints'] for dt in dts], dtype=np.float64)
        xd, yd = d[:, 0::3], d[:, 1::3]
        out = np.zeros((len(dts), len(gts)))
        for j, gt in enumerate(gts):
            g = np.array(gt['keypoints'])
            xg, yg, vg = g[0::3], g[1::3], g[2::3]
            k1 = np.count_nonzero(vg > 0)
            if k1 > 0:
                dx, dy = xd - xg, yd - yg
            else:
                bb = gt['bbox']
                x0, x1, y0, y1 = bb[0] - bb[2], bb[0] + bb[2] * 2, bb[1] - bb[3], bb[1] + bb[3] * 2
                z = np.zeros_like(xd)
                dx = np.maximum(z, x0 - xd) + np.maximum(z, xd - x1)
                dy = np.maximum(z, y0 - yd) + np.maximum(z, yd - y1)
            e = (dx ** 2 + dy ** 2) / var / (gt['area'] + np.spacing(1)) / 2
            if k1 > 0:
                e = e[:, vg > 0]
            out[:, j] = np.sum(np.exp(-e), axis=1) / e.shape[1]
        return out

    def evaluate(self):
        p = self.params
        p.img_ids = list(np.unique(p.img_ids))
        if p.use_cats:
            p.cat_ids = list(np.unique(p.cat_ids))
        p.max_dets = sorted(p.max_dets)
        gts = self._collect(self.gt, p.img_ids, p.cat_ids)
        dts = self._collect(self.dt, p.img_ids, p.cat_ids)
        if not p.use_cats:                                    # one pseudo category holding everything
            def fold(table):
                out = {}
                for c in p.cat_ids:
                    for i in p.img_ids:
                        out.setdefault((i, -1), []).extend(table.get((i, c), ()))
                return out
            gts, dts = fold(gts), fold(dts)
        rle = {}
        if p.iou_type == 'segm':
            for index, table in ((self.gt, gts), (self.dt, dts)):
                for anns in table.values():
                    for a in anns:
                        rle[id(a)] = ann_to_rle(index, a)
        cat_ids = p.cat_ids if p.use_cats else [-1]
        T, budget = len(p.iou_thrs), p.max_dets[-1]
        thrs = np.ascontiguousarray(p.iou_thrs, dtype=np.float64)
        lib = mask_util.lib()
        i64p, f64p, u8p = C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_uint8)
        records = {}
        for c in cat_ids:
            for i in p.img_ids:
                g, d = gts.get((i, c), []), dts.get((i, c), [])
                if not g and not d:
                    continue
                order = np.argsort([-x['score'] for x in d], kind='mergesort')[:budget]
                d = [d[k] for k in order]
                ignore = np.array([bool(x.get('iscrowd', 0)) or
                                   (p.iou_type == 'keypoints' and x['num_keypoints'] == 0) for x in g], dtype=bool)
                sim = self._similarity(g, d, rle) if (g or d) else []
                sim = np.asarray(sim, dtype=np.float64).reshape(len(d), len(g)) if len(sim) else np.zeros((len(d), len(g)))
                g_area = np.array([x['area'] for x in g], dtype=np.float64)
                d_area = np.array([x['area'] for x in d], dtype=np.float64)
                crowd = np.array([int(x['iscrowd']) for x in g], dtype=np.uint8)
                per_area = []
                for lo, hi in p.area_rng:
                    g_ig = ignore | (g_area < lo) | (g_area > hi)
                    gorder = np.argsort(g_ig, kind='mergesort')             # real ground truths first
                    g_ig_s = np.ascontiguousarray(g_ig[gorder], dtype=np.uint8)
                    dm = np.full((T, len(d)), -1, dtype=np.int64)
                    gm = np.full((T, len(g)), -1, dtype=np.int64)
                    if len(g) and len(d):
                        s = np.ascontiguousarray(sim[:, gorder])
                        cr = np.ascontiguousarray(crowd[gorder])
                        lib.lsn_coco_match(s.ctypes.data_as(f64p), len(d), len(g), g_ig_s.ctypes.data_as(u8p),
                                           cr.ctypes.data_as(u8p), thrs.ctypes.data_as(f64p), T,
                                           dm.ctypes.data_as(i64p), gm.ctypes.data_as(i64p))
                    matched = dm >= 0
                    d_ig = np.zeros((T, len(d)), dtype=bool)
                    if len(g):
                        d_ig[matched] = g_ig_s[dm[matched]].astype(bool)
                    outside = (d_area < lo) | (d_area > hi)
                    d_ig |= (~matched) & outside[None, :]
                    per_area.append(dict(matched=matched, dt_ignore=d_ig, gt_ignore=g_ig_s.astype(bool)))
                records[(c, i)] = dict(scores=np.array([x['score'] for x in d], dtype=np.float64), areas=per_area)
        self.eval_imgs = records
        self._evaluated = copy.deepcopy(p)

    # ------------------------------------------------------------------------------------------- accumulate
    def accumulate(self):
        p = self._evaluated
        cat_ids = p.cat_ids if p.use_cats else [-1]
        T, R, K, A, M = len(p.iou_thrs), len(p.rec_thrs), len(cat_ids), len(p.area_rng), len(p.max_dets)
        precision = -np.ones((T, R, K, A, M))
        recall = -np.ones((T, K, A, M))
        scores = -np.ones((T, R, K, A, M))
        eps = np.spacing(1)
        for k, c in enumerate(cat_ids):
            recs = [self.eval_imgs[(c, i)] for i in p.img_ids if (c, i) in self.eval_imgs]
            if not recs:
                continue
            for a in range(A):
                gt_ig = np.concatenate([r['areas'][a]['gt_ignore'] for r in recs])
                n_real = np.count_nonzero(~gt_ig)
                if n_real == 0:
                    continue
                for m, budget in enumerate(p.max_dets):
                    sc = np.concatenate([r['scores'][:budget] for r in recs])
                    order = np.argsort(-sc, kind='mergesort')
                    sc = sc[order]
                    hit = np.concatenate([r['areas'][a]['matched'][:, :budget] for r in recs], axis=1)[:, order]
                    ig = np.concatenate([r['areas'][a]['dt_ignore'][:, :budget] for r in recs], axis=1)[:, order]
                    tp = np.cumsum(hit & ~ig, axis=1).astype(np.float64)
                    fp = np.cumsum(~hit & ~ig, axis=1).astype(np.float64)
                    nd = tp.shape[1]
                    for t in range(T):
                        rc = tp[t] / n_real
                        pr = tp[t] / (fp[t] + tp[t] + eps)
                        recall[t, k, a, m] = rc[-1] if nd else 0
                        pr = np.maximum.accumulate(pr[::-1])[::-1]          # precision envelope
                        at = np.searchsorted(rc, p.rec_thrs, side='left')
                        ok = at < nd
                        q, ss = np.zeros(R), np.zeros(R)
                        q[ok], ss[ok] = pr[at[ok]], sc[at[ok]]
                        precision[t, :, k, a, m] = q
                        scores[t, :, k, a, m] = ss
        self.eval = dict(params=p, counts=[T, R, K, A, M], precision=precision, recall=recall, scores=scores)

    # -------------------------------------------------------------------------------------------- summarize
    def _mean(self, ap, iou_thr=None, area='all', max_dets=100):
        p = self.params
        a = [i for i, lbl in enumerate(p.area_lbl) if lbl == area]
        m = [i for i, v in enumerate(p.max_dets) if v == max_dets]
        s = self.eval['precision'] if ap else self.eval['recall']
        if iou_thr is not None:
            s = s[np.where(iou_thr == p.iou_thrs)[0]]
        s = s[:, :, :, a, m] if ap else s[:, :, a, m]
        return -1 if len(s[s > -1]) == 0 else float(np.mean(s[s > -1]))

    def summarize(self, printer=None):
        if not self.eval:
            raise Exception('Please run accumulate() first')
        p = self.params
        if p.iou_type == 'keypoints':
            spec = [(1, None, 'all', 20), (1, .5, 'all', 20), (1, .75, 'all', 20), (1, None, 'medium', 20),
                    (1, None, 'large', 20), (0, None, 'all', 20), (0, .5, 'all', 20), (0, .75, 'all', 20),
                    (0, None, 'medium', 20), (0, None, 'large', 20)]
        else:
            md = p.max_dets
            spec = [(1, None, 'all', 100), (1, .5, 'all', md[2]), (1, .75, 'all', md[2]), (1, None, 'small', md[2]),
                    (1, None, 'medium', md[2]), (1, None, 'large', md[2]), (0, None, 'all', md[0]),
                    (0, None, 'all', md[1]), (0, None, 'all', md[2]), (0, None, 'small', md[2]),
                    (0, None, 'medium', md[2]), (0, None, 'large', md[2])]
        stats = np.zeros(len(spec))
        for n, (ap, thr, area, md_) in enumerate(spec):
            stats[n] = self._mean(ap, thr, area, md_)
            if printer is not None:
                iou = f'{p.iou_thrs[0]:0.2f}:{p.iou_thrs[-1]:0.2f}' if thr is None else f'{thr:0.2f}'
                kind = ('Average Precision', '(AP)') if ap else ('Average Recall', '(AR)')
                printer(f' {kind[0]:<18} {kind[1]} @[ IoU={iou:<9} | area={area:>6s} | maxDets={md_:>3d} ] = {stats[n]:0.3f}')
        self.stats = stats
        return stats
