"""Golden vectors for the point-annotation JSON: executes the reference's own
VOCCocoDatasetPoi._parse_ann_info (extracted from mmdet_plugins/datasets/voc_coco_poi.py with ast, because importing the
module needs mmdet's CocoDataset / mmcv) on a small synthetic annotation file that hits every branch, and stores the
input file + the outputs in tests/golden/annotations.json.  Container-only (needs /root/reference)."""
import ast
import json
import os
import types

import numpy as np

REF = "/root/reference/mmdet_plugins/datasets/voc_coco_poi.py"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_parse():
    tree = ast.parse(open(REF).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef))
    fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "_parse_ann_info")
    ns = {"np": np}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), REF, "exec"), ns)
    return ns["_parse_ann_info"]


def main():
    cats = [dict(id=i + 1, name=n) for i, n in enumerate(
        ('aeroplane', 'bicycle', 'bird', 'boat', 'bottle', 'bus', 'car', 'cat', 'chair', 'cow', 'diningtable', 'dog',
         'horse', 'motorbike', 'person', 'pottedplant', 'sheep', 'sofa', 'train', 'tvmonitor'))]
    cats.append(dict(id=99, name="unicorn"))                                     # a category outside CLASSES
    images = [dict(id=10, file_name="a.jpg", width=500, height=375), dict(id=11, file_name="b.jpg", width=320, height=240),
              dict(id=12, file_name="c.jpg", width=100, height=100), dict(id=13, file_name="d.jpg", width=64, height=64)]
    sq = lambda x, y, w, h: [[x, y, x, y + h, x + w, y + h, x + w, y]]
    anns = [
        # image 10: box mode; a normal object, a crowd, an ignored one, a foreign category, a degenerate and an outside box
        dict(id=1, image_id=10, bbox=[219, 80, 250, 284], area=71000, iscrowd=0, category_id=9, ignore=0,
             point=[332.226, 241.908], segmentation=sq(219, 80, 250, 284)),
        dict(id=2, image_id=10, bbox=[10, 10, 50, 60], area=3000, iscrowd=1, category_id=15, point=[30.0, 40.0]),
        dict(id=3, image_id=10, bbox=[1, 1, 20, 20], area=400, iscrowd=0, category_id=3, ignore=1, point=[5.0, 5.0]),
        dict(id=4, image_id=10, bbox=[5, 5, 20, 20], area=400, iscrowd=0, category_id=99, point=[9.0, 9.0]),
        dict(id=5, image_id=10, bbox=[5, 5, 0.5, 20], area=10, iscrowd=0, category_id=1, point=[5.1, 9.0]),
        dict(id=6, image_id=10, bbox=[600, 5, 20, 20], area=400, iscrowd=0, category_id=1, point=[610.0, 9.0]),
        dict(id=7, image_id=10, bbox=[400, 300, 200, 200], area=40000, iscrowd=0, category_id=20, point=[450.5, 330.25]),
        # image 11: point-only mode; a bad point, a missing point, a foreign category
        dict(id=8, image_id=11, category_id=12, point=[100.5, 50.25], segmentation=sq(80, 30, 40, 40)),
        dict(id=9, image_id=11, category_id=8, point=[10.0, 20.0, 1.0]),
        dict(id=10, image_id=11, category_id=8),
        dict(id=11, image_id=11, category_id=99, point=[1.0, 2.0]),
        dict(id=12, image_id=11, category_id=1, point=[300.0, 200.0], ignore=0),
        # image 12: box mode where nothing survives;  image 13: no annotations at all
        dict(id=13, image_id=12, bbox=[0, 0, 0, 0], area=0, iscrowd=0, category_id=2, point=[0.0, 0.0]),
    ]
    data = dict(images=images, annotations=anns, categories=cats)
    parse = load_parse()
    classes = [c["name"] for c in cats[:-1]]
    cat_ids = [c["id"] for c in cats[:-1]]
    self = types.SimpleNamespace(cat_ids=cat_ids, cat2label={c: i for i, c in enumerate(cat_ids)})
    want = []
    for info in images:
        info2 = dict(info, filename=info["file_name"])
        out = parse(self, info2, [a for a in anns if a["image_id"] == info["id"]])
        want.append({k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in out.items()}
                    | {"_dtypes": {k: str(v.dtype) for k, v in out.items() if isinstance(v, np.ndarray)},
                       "_shapes": {k: list(v.shape) for k, v in out.items() if isinstance(v, np.ndarray)}})
    path = os.path.join(ROOT, "tests", "golden", "annotations.json")
    json.dump(dict(file=data, classes=classes, expected=want), open(path, "w"), indent=1)
    print("wrote", path, [list(w["_shapes"].items()) for w in want])


if __name__ == "__main__":
    main()
