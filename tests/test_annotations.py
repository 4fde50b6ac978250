"""Point-annotation JSON reader vs the reference's VOCCocoDatasetPoi._parse_ann_info (fixture produced by executing the
reference function: tools/gen_golden_annotations.py).  CPU only."""
import json
import os

import numpy as np
import torch

from attentionshift_amd import annotations as AN

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "annotations.json")


def test_parse_matches_the_reference_on_every_branch(tmp_path):
    g = json.load(open(GOLD))
    path = tmp_path / "ann.json"
    path.write_text(json.dumps(g["file"]))
    ds = AN.PointAnnotations(str(path), classes=g["classes"])
    assert len(ds) == len(g["expected"]) == 4 and ds.classes == AN.VOC_CLASSES and 99 not in ds.cat_ids
    for i, want in enumerate(g["expected"]):
        got = ds.ann(i)
        assert set(got) == set(want) - {"_dtypes", "_shapes"}, i
        for k, v in got.items():
            if isinstance(v, np.ndarray):
                assert str(v.dtype) == want["_dtypes"][k] and list(v.shape) == want["_shapes"][k], (i, k)
                assert np.array_equal(v, np.array(want[k], dtype=v.dtype).reshape(v.shape)), (i, k)
            else:
                assert v == want[k], (i, k)
    assert "bboxes" in ds.ann(0) and "bboxes" not in ds.ann(1)              # box mode / point-only mode


def test_head_inputs_follow_resize_and_flip():
    g = json.load(open(GOLD))
    ds = AN.PointAnnotations(g["file"], classes=g["classes"])
    pts, labels, boxes = AN.to_head_inputs(ds.ann(0), scale=2.0, flip_w=1000.0)
    assert labels.tolist() == [8, 19] and labels.dtype == torch.long
    assert torch.allclose(pts[0], torch.tensor([1000 - 2 * 332.226, 2 * 241.908]))
    assert torch.allclose(boxes[0], torch.tensor([1000 - 2 * 469., 160., 1000 - 2 * 219., 728.]))
    assert bool(((pts[:, 0] >= boxes[:, 0]) & (pts[:, 0] <= boxes[:, 2])).all())   # points stay inside their boxes
    pts1, _, boxes1 = AN.to_head_inputs(ds.ann(1), scale=(0.5, 2.0))
    assert boxes1 is None and torch.allclose(pts1[0], torch.tensor([50.25, 100.5]))
    e = AN.to_head_inputs(ds.ann(3))
    assert e[0].shape == (0, 2) and e[1].shape == (0,) and e[2].shape == (0, 4)
