"""Golden vectors for the pseudo-label consumers: runs the reference's own module-level functions
update_coords_with_semantic_centers / get_point_coords_wrt_box (tools/ref_import.py) on seeded ragged inputs and stores
inputs + outputs in tests/golden/consumers.npz.  Container-only (needs /root/reference)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_import  # noqa: E402


def main():
    ns = ref_import.load_roi_functions()
    gen = torch.Generator().manual_seed(77)
    store = {}
    # three images: 3 objects / 2 objects / 4 objects with NO semantic centres (the pass-through branch)
    shapes = [(3, 10), (2, 10), (4, 10)]
    coords = [torch.rand(g, p, 2, generator=gen) * 500 for g, p in shapes]
    labels = [torch.rand(g, p, generator=gen) > 0.5 for g, p in shapes]
    labels[0][1] = True                                             # an object without negative points
    centers = [[torch.rand(k, 2, generator=gen) * 500 for k in (2, 0, 4)], [torch.rand(k, 2, generator=gen) * 500 for k in (1, 3)], []]
    out_c, out_l = ns["update_coords_with_semantic_centers"](coords, labels, centers)
    for i in range(3):
        store[f"coords{i}"], store[f"labels{i}"] = coords[i].numpy(), labels[i].numpy()
        store[f"out_coords{i}"], store[f"out_labels{i}"] = out_c[i].numpy(), out_l[i].numpy()
        store[f"ncenters{i}"] = np.array([c.shape[0] for c in centers[i]], dtype=np.int64)
        for g, c in enumerate(centers[i]):
            store[f"center{i}_{g}"] = c.numpy()
    boxes = torch.tensor([[10., 20., 210., 320.], [0., 0., 500., 500.], [100., 100., 101., 400.]])
    pts = torch.rand(3, 7, 2, generator=gen) * 500
    store["boxes"], store["pts"] = boxes.numpy(), pts.numpy()
    store["pts_wrt_box"] = ns["get_point_coords_wrt_box"](boxes, pts).numpy()
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "consumers.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, {k: v.shape for k, v in store.items() if k.startswith("out_")})


if __name__ == "__main__":
    main()
