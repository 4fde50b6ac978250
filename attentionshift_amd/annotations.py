"""The point-annotated COCO-style JSON the reference trains from (mmdet_plugins/datasets/
voc_coco_poi.py:13-109): every object annotation carries `"point": [x, y]` next to (or instead of) `"bbox"`.

    parse_ann_info(img_info, ann_info, cat_ids, cat2label)   VOCCocoDatasetPoi._parse_ann_info: the per-image dict
                                                             {bboxes?, labels, points, bboxes_ignore, masks, seg_map}
    PointAnnotations(path | dict, classes)                   the file-level index pycocotools gives CocoDataset
                                                             (images, category ids in id order -> labels, annotations
                                                             per image); `ann(i)` = parse_ann_info of image i
    to_head_inputs(ann, scale, flip_w, device)               the tensors the RoI head takes (gt_points, labels, boxes)
                                                             after the pipeline's resize / horizontal flip

No mmdet / pycocotools dependency: plain json + numpy.  The image pipeline itself (loading, augmentation, collation)
stays the reference's (SURVEY 2 "OUT")."""
import json

import numpy as np

VOC_CLASSES = ('aeroplane', 'bicycle', 'bird', 'boat', 'bottle', 'bus', 'car', 'cat', 'chair', 'cow', 'diningtable',
               'dog', 'horse', 'motorbike', 'person', 'pottedplant', 'sheep', 'sofa', 'train', 'tvmonitor')


def parse_ann_info(img_info, ann_info, cat_ids, cat2label):
    """voc_coco_poi.py:12-109.  Box mode when the first non-ignored annotation has a 'bbox' (degenerate / crowd /
    foreign-category boxes dropped exactly as CocoDataset does), point-only mode otherwise (only annotations with a
    two-number 'point')."""
    with_bbox = True
    for a in ann_info:
        if a.get("ignore", False):
            continue
        if "bbox" not in a:
            with_bbox = False
        break
    boxes, labels, points, ignore, masks = [], [], [], [], []
    for a in ann_info:
        if a.get("ignore", False):
            continue
        if with_bbox:
            x1, y1, w, h = a["bbox"]
            iw = max(0, min(x1 + w, img_info["width"]) - max(x1, 0))
            ih = max(0, min(y1 + h, img_info["height"]) - max(y1, 0))
            if iw * ih == 0 or a["area"] <= 0 or w < 1 or h < 1 or a["category_id"] not in cat_ids:
                continue
            if a.get("iscrowd", False):
                ignore.append([x1, y1, x1 + w, y1 + h])
                continue
            boxes.append([x1, y1, x1 + w, y1 + h])
            points.append(a.get("point", None))
        else:
            p = a.get("point", None)
            if a["category_id"] not in cat_ids or p is None or len(p) != 2:
                continue
            points.append(p)
        labels.append(cat2label[a["category_id"]])
        masks.append(a.get("segmentation", None))
    out = {}
    if with_bbox:
        out["bboxes"] = np.array(boxes, dtype=np.float32) if boxes else np.zeros((0, 4), dtype=np.float32)
    out["labels"] = np.array(labels, dtype=np.int64)
    out["points"] = np.array(points, dtype=np.float32) if points else np.zeros((0, 2), dtype=np.float32)
    out["bboxes_ignore"] = np.array(ignore, dtype=np.float32) if ignore else np.zeros((0, 4), dtype=np.float32)
    out["masks"] = masks
    out["seg_map"] = img_info["filename"].replace("jpg", "png")
    return out


class PointAnnotations:
    """What CocoDataset.load_annotations / get_ann_info provide, without pycocotools (mmdet/datasets/coco.py:45-95):
    category ids of `classes` by NAME in the order of `classes`, label = position; images in id order, each with
    `filename = file_name`."""

    def __init__(self, source, classes=VOC_CLASSES):
        data = source if isinstance(source, dict) else json.load(open(source))
        by_name = {c["name"]: c["id"] for c in data.get("categories", [])}
        self.classes = tuple(classes)
        self.cat_ids = [by_name[c] for c in self.classes if c in by_name]
        self.cat2label = {cid: i for i, cid in enumerate(self.cat_ids)}
        self.img_infos = []
        for info in sorted(data.get("images", []), key=lambda d: d["id"]):
            info = dict(info)
            info["filename"] = info["file_name"]
            self.img_infos.append(info)
        self._anns = {}
        for a in data.get("annotations", []):
            self._anns.setdefault(a["image_id"], []).append(a)

    def __len__(self):
        return len(self.img_infos)

    def ann(self, idx):
        info = self.img_infos[idx]
        return parse_ann_info(info, self._anns.get(info["id"], []), self.cat_ids, self.cat2label)


def to_head_inputs(ann, scale=1.0, flip_w=None, device="cpu"):
    """(gt_points [G,2], gt_labels [G], gt_bboxes [G,4] | None) as torch tensors after a resize by `scale` (a number or
    (sx, sy)) and, if `flip_w` is the resized image width, a horizontal flip -- the two geometric transforms of the
    reference's training pipeline that touch the annotations (configs/mae/attnshift_voc12aug.py train_pipeline)."""
    import torch
    sx, sy = (scale, scale) if np.isscalar(scale) else scale
    pts = torch.as_tensor(ann["points"], dtype=torch.float32).clone().reshape(-1, 2)
    pts[:, 0] *= sx
    pts[:, 1] *= sy
    boxes = None
    if "bboxes" in ann:
        boxes = torch.as_tensor(ann["bboxes"], dtype=torch.float32).clone().reshape(-1, 4)
        boxes[:, 0::2] *= sx
        boxes[:, 1::2] *= sy
    if flip_w is not None:
        pts[:, 0] = flip_w - pts[:, 0]
        if boxes is not None:
            x1 = flip_w - boxes[:, 2]
            boxes[:, 2] = flip_w - boxes[:, 0]
            boxes[:, 0] = x1
    labels = torch.as_tensor(ann["labels"], dtype=torch.long)
    return pts.to(device), labels.to(device), None if boxes is None else boxes.to(device)
