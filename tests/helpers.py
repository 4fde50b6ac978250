"""Shared helpers for the parity tests (CPU and GPU)."""
import numpy as np
import torch

from attentionshift_amd import synthetic


def t(a):
    return torch.from_numpy(np.asarray(a))


def backbone_state_dict(g):
    """Rebuild the deterministic weights a backbone fixture was generated with."""
    shapes = {}
    for name, shp in zip(g["param_names"].tolist(), g["param_shapes"].tolist()):
        shapes[name] = tuple(int(v) for v in shp.split(",")) if shp else ()
    return synthetic.det_state_dict(shapes)


def backbone_cfg(g):
    return dict(img_size=int(g["cfg_img_size"]), embed_dim=int(g["cfg_embed_dim"]), depth=int(g["cfg_depth"]),
                num_heads=int(g["cfg_num_heads"]), point_tokens_num=int(g["cfg_point_tokens_num"]),
                num_classes=int(g["cfg_num_classes"]), out_indices=tuple(int(v) for v in g["cfg_out_indices"]),
                batch=int(g["cfg_batch"]), seed=int(g["cfg_seed"]), cam_layer=int(g["cfg_cam_layer"]),
                img_hw=tuple(int(v) for v in g["img_hw"]))


def shift_case_inputs(g):
    inp = synthetic.shift_inputs(int(g["seed"]), int(g["hp"]), int(g["wp"]), int(g["C"]), int(g["G"]), int(g["Lc"]))
    return inp


def assert_close(ref, got, rtol=1e-3, atol=1e-5, what=""):
    ref = ref.detach().cpu().double() if torch.is_tensor(ref) else torch.as_tensor(np.asarray(ref)).double()
    got = got.detach().cpu().double() if torch.is_tensor(got) else torch.as_tensor(np.asarray(got)).double()
    assert ref.shape == got.shape, f"{what}: shape {tuple(ref.shape)} vs {tuple(got.shape)}"
    if ref.numel() == 0:
        return
    err = (ref - got).abs()
    tol = atol + rtol * ref.abs()
    bad = err > tol
    assert not bad.any(), (f"{what}: {int(bad.sum())}/{ref.numel()} beyond tol; max err {err.max().item():.3e} "
                           f"(max |ref| {ref.abs().max().item():.3e})")


def assert_equal(ref, got, what=""):
    ref = ref.detach().cpu().numpy() if torch.is_tensor(ref) else np.asarray(ref)
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    assert ref.shape == got.shape, f"{what}: shape {ref.shape} vs {got.shape}"
    bad = int((ref != got).sum())
    assert bad == 0, f"{what}: {bad}/{ref.size} elements differ"
