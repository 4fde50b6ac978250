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


# ---- full-size mean-shift parity: which argmax decisions are determined, and which are rounding coin flips --------
def shift_state_inputs(g, inp):
    """(feats_masked [G,Np,C], tokens [Np,C], seeds prot [G,P,C], box_patch [G,4] int32) of a shift_* fixture that stores
    `seed_coords` (the slim full-size cases)."""
    import attnshift_oracle as O
    hp, wp = int(g["hp"]), int(g["wp"])
    rois = t(g["rois"])
    tok = inp["vit_feat"].flatten(1).t()              # a transposed view, as oracle.mean_shift_prototypes builds it
    inbox = O.box_mask(rois // 16, (hp, wp)).flatten(1)
    sc = t(g["seed_coords"]).long()
    prot = inp["vit_feat"].permute(1, 2, 0)[sc[..., 0], sc[..., 1]].contiguous()
    return tok[None] * inbox[..., None], tok, prot, (rois // 16).int()


def check_shift_decisions(step, got_assign, prot, feats, tau, temp=0.1, what="", max_flips=16):
    """`got_assign` [G,Np] must equal the reference arithmetic's argmax `step['win']` (one iteration of
    cosine_shift_batch evaluated in fp32 from the SAME state) wherever that argmax is determined.  A differing patch is
    accepted only if it is a rounding coin flip of the reference itself:
      * near tie: the float64 log-weights of the two candidates differ by less than the fp32 evaluation noise of
        cos/(temp*tau) (2e-7 absolute on a cosine, amplified by 1/(temp*tau) of either prototype) + 2e-5, or
      * underflow class: both candidates' fp32 softmax weights are below the smallest normal number (the reference
        compares denormals / zeros whose value depends on libm's rounding of exp near 1e-44).
    Returns (mismatches, near_ties, underflow)."""
    import attnshift_oracle as O
    want = step["win"]
    got = got_assign.long().cpu()
    bad = (got != want)
    if not bad.any():
        return 0, 0, 0
    G, P, Np = step["w"].shape
    logw = O.shift_log_weights64(prot, feats, tau, temp)
    tt = (temp * torch.as_tensor(tau, dtype=torch.float64)).expand(G, P, 1)[..., 0] if torch.is_tensor(tau) \
        else torch.full((G, P), temp * tau, dtype=torch.float64)
    gi, ni = bad.nonzero(as_tuple=True)
    a, r = got[gi, ni], want[gi, ni]
    margin = (logw[gi, a, ni] - logw[gi, r, ni]).abs()
    noise = 2e-7 * (1.0 / tt[gi, a] + 1.0 / tt[gi, r]) + 2e-5
    near = margin <= noise
    tiny = torch.finfo(torch.float32).tiny
    under = (step["w"][gi, a, ni] < tiny) & (step["w"][gi, r, ni] < tiny)
    ok = near | under
    assert ok.all(), (f"{what}: {int((~ok).sum())} cluster assignments differ from the reference arithmetic on DETERMINED "
                      f"decisions (worst margin {float(margin[~ok].max()):.3e} vs noise {float(noise[~ok].min()):.3e})")
    # the CPU matmul-form oracle itself flips at most 16 of 12 288 decisions per iteration against the reference
    # (test_full_size_matmul_oracle_differs_from_reference_only_on_coin_flips); more than that is a regression
    assert int(bad.sum()) <= max_flips, f"{what}: {int(bad.sum())} coin-flip patches (> {max_flips}) is implausibly many"
    return int(bad.sum()), int(near.sum()), int((under & ~near).sum())
