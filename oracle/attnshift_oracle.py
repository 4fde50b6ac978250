"""CPU restatement (torch fp32 / numpy) of AttentionShift's data-parallel hot path.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline`
leg may import this module; the product path (attentionshift_amd/) never does and fails loudly
when the HIP library is missing.

Every function cites the reference lines it restates (paths relative to /root/reference;
`stdroi` = mmdet/models/roi_heads/stdroi_point_deform_attn_reppoints.py).  The restatement is
pinned by tests/golden/*.npz, which tools/gen_golden.py produced by executing the reference's
own functions in the build container.  One third-party piece is NOT
pinned by anything in the reference: `cc_torch.connected_components_labeling` (the directory
Connected_components_PyTorch/ is empty).  Its restatement here (`ccl_labels`) follows the
upstream documentation (8-connectivity) with our own numbering (1 + min raster index of the
component): parity for the *numbering* is unpinned, the *partition* is what the consumer uses.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

EPS_COS = 1e-8


# --------------------------------------------------------------------------------------
# small shared pieces
# --------------------------------------------------------------------------------------
def unit_rows(x, dim=-1, eps=EPS_COS):
    """x / max(||x||, eps): the per-operand normalisation of torch>=1.12 F.cosine_similarity."""
    return x / x.norm(dim=dim, keepdim=True).clamp_min(eps)


def cos_matrix(a, b, eps=EPS_COS):
    """cos(a_i, b_j) for a [..., P, C], b [..., M, C] -> [..., P, M] as a normalised matmul.
    Same value (to fp32 round-off) as F.cosine_similarity on the broadcast pair, without the
    [P, M, C] temporary the reference allocates (stdroi:832)."""
    return unit_rows(a, eps=eps) @ unit_rows(b, eps=eps).transpose(-1, -2)


def cos_broadcast(a, b, eps=EPS_COS, chunk=4):
    """cos(a_i, b_j) for a [G,P,C], b [G,M,C] or [M,C] -> [G,P,M] with the reference's OWN arithmetic:
    F.cosine_similarity on the broadcast pair (stdroi:832, :848, :883), i.e. both operands normalised, multiplied
    element-wise into a [P,M,C] temporary and summed over C by torch's reduction.  The rounding of that sum differs
    from a matmul's in the last bits, which decides near-tied argmaxes at full size, so the fixtures are
    pinned through this form.  Evaluated `chunk` prototypes at a time: each output element's reduction is independent
    of the others, the chunking only bounds the temporary (755 MB at 64x64x768 otherwise)."""
    if b.dim() == 2:
        b = b[None].expand(a.shape[0], -1, -1)
    out = []
    for g in range(a.shape[0]):
        # all-zero rows of b (patches outside the object's box, stdroi:1819) give exactly 0 (0 / eps * ... = 0): only the
        # non-zero rows are evaluated, the result is bit-identical to the full broadcast
        nz = (b[g] != 0).any(-1)
        # keep the operand's memory order: the reference's features are a transposed VIEW of the [C, Np] map
        # (stdroi:1816-1824), torch then reduces over C as the OUTER dimension (a sequential sum per patch), and a
        # contiguous [n, C] copy would be reduced in a different order
        bn = b[g][nz] if b[g].stride(-1) == 1 else b[g].t()[:, nz].t()
        rows = [F.cosine_similarity(a[g, p0:p0 + chunk, None, :], bn[None, :, :], dim=-1, eps=eps)
                for p0 in range(0, a.shape[1], chunk)]
        full = torch.zeros(a.shape[1], b.shape[1], dtype=a.dtype)
        full[:, nz] = torch.cat(rows, dim=0)
        out.append(full)
    return torch.stack(out)


def box_mask(boxes, size, default_val=0.0):
    """stdroi:303-309 box2mask: inclusive integer box [x0..x1] x [y0..y1] set to 1."""
    n = boxes.shape[0]
    m = torch.full((n, size[0], size[1]), float(default_val), dtype=boxes.dtype)
    for i in range(n):
        x0, y0, x1, y1 = (int(v) for v in boxes[i])
        m[i, y0:y1 + 1, x0:x1 + 1] = 1.0
    return m


def fill_in(idx, want):
    """stdroi:1147-1155 fill_in_idx: cyclic repeat of the rows of `idx` up to `want` rows."""
    assert idx.shape[0] != 0
    while idx.shape[0] < want / 2:
        rep = want // idx.shape[0]
        idx = idx.repeat(rep, *([1] * (idx.dim() - 1))) if idx.dim() > 1 else idx.repeat(rep, 1)
    return torch.cat((idx, idx[: want - idx.shape[0]]), dim=0)


def erode(x, k):
    """stdroi:145-146 / 1182-1187 corrosion: min-pool k x k, stride 1, pad k//2 (implicit pad
    never wins because max_pool2d pads with -inf on the negated map)."""
    shp = x.shape
    y = -F.max_pool2d(-x.reshape(1, -1, shp[-2], shp[-1]), k, 1, k // 2)
    return y.reshape(shp)


def upsample_bilinear(x, out_h, out_w):
    """F.interpolate(size=..., mode='bilinear', align_corners=False) on the trailing two dims:
    the op the reference calls at stdroi:2279, 1010-1011."""
    lead = x.shape[:-2]
    y = F.interpolate(x.reshape(1, -1, *x.shape[-2:]), (out_h, out_w), mode="bilinear")
    return y.reshape(*lead, out_h, out_w)


def upsample_bilinear_explicit(x, out_h, out_w):
    """The same op spelled out step by step: this is the arithmetic the HIP kernels implement, and
    tests assert it is BIT-identical to `upsample_bilinear` on CPU.  Per output pixel (ATen
    UpSampleKernel, align_corners=False): src = max(scale*(dst+0.5)-0.5, 0) with scale = in/out in
    fp32; i0 = int(src), i1 = min(i0+1, in-1); l1 = src-i0, l0 = 1-l1;
    row(r) = fma(lx0, v[r][x0], lx1*v[r][x1]);  out = fma(ly0, row(y0), ly1*row(y1))."""
    h, w = x.shape[-2:]

    def axis(n_in, n_out):
        scale = np.float32(n_in) / np.float32(n_out)
        dst = np.arange(n_out, dtype=np.float32)
        src = np.maximum(scale * (dst + np.float32(0.5)) - np.float32(0.5), np.float32(0))
        i0 = np.minimum(src.astype(np.int64), n_in - 1)
        i1 = np.minimum(i0 + 1, n_in - 1)
        l1 = (src - i0.astype(np.float32)).astype(np.float32)
        l0 = (np.float32(1) - l1).astype(np.float32)
        return i0, i1, l0, l1

    def fma(a, b, c):        # fp32 fma emulated through an exact fp64 product
        return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)

    y0, y1, wy0, wy1 = axis(h, out_h)
    x0, x1, wx0, wx1 = axis(w, out_w)
    v = x.detach().numpy().astype(np.float32)
    top, bot = v[..., y0, :], v[..., y1, :]
    r_top = fma(np.broadcast_to(wx0, top[..., x0].shape), top[..., x0], wx1 * top[..., x1])
    r_bot = fma(np.broadcast_to(wx0, bot[..., x0].shape), bot[..., x0], wx1 * bot[..., x1])
    wy0b = np.broadcast_to(wy0[:, None], r_top.shape)
    return torch.from_numpy(fma(wy0b, r_top, wy1[:, None] * r_bot))


# --------------------------------------------------------------------------------------
# Part A: backbone attention  (models/vision_transformer.py, visual_transformer_det.py)
# --------------------------------------------------------------------------------------
def attention(x, w_qkv, b_qkv, w_proj, b_proj, num_heads):
    """models/vision_transformer.py:74-86 Attention.forward -> (out [B,N,D], P [B,h,N,N])."""
    B, N, D = x.shape
    d = D // num_heads
    qkv = F.linear(x, w_qkv, b_qkv).reshape(B, N, 3, num_heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    p = ((q @ k.transpose(-2, -1)) * (d ** -0.5)).softmax(dim=-1)
    o = (p @ v).transpose(1, 2).reshape(B, N, D)
    return F.linear(o, w_proj, b_proj), p


def attention_head_mean(x, w_qkv, b_qkv, num_heads):
    """The `attn.mean(1)` the backbone keeps per layer (mmdet/models/backbones/visual_transformer_det.py:236,242) of
    Attention.forward's softmax (models/vision_transformer.py:75-81), accumulated head by head so that the
    [B,h,N,N] tensor of `attention` is never held (N = 4197: 845 MB per layer) -> [B,N,N]."""
    B, N, D = x.shape
    d = D // num_heads
    qkv = F.linear(x, w_qkv, b_qkv).reshape(B, N, 3, num_heads, d).permute(2, 0, 3, 1, 4)
    mean = torch.zeros(B, N, N, dtype=x.dtype)
    for hd in range(num_heads):
        mean += ((qkv[0][:, hd] @ qkv[1][:, hd].transpose(-2, -1)) * (d ** -0.5)).softmax(dim=-1)
    return mean / num_heads


def block(x, sd, prefix, num_heads, ln_eps=1e-6):
    """models/vision_transformer.py:109-124 Block.forward (gamma_1/2 None, drop_path identity in
    eval) -> (x, P)."""
    g = lambda k: sd[prefix + k]
    D = x.shape[-1]
    y, p = attention(F.layer_norm(x, (D,), g("norm1.weight"), g("norm1.bias"), ln_eps),
                     g("attn.qkv.weight"), sd.get(prefix + "attn.qkv.bias"),
                     g("attn.proj.weight"), g("attn.proj.bias"), num_heads)
    x = x + y
    z = F.layer_norm(x, (D,), g("norm2.weight"), g("norm2.bias"), ln_eps)
    z = F.linear(F.gelu(F.linear(z, g("mlp.fc1.weight"), g("mlp.fc1.bias"))),
                 g("mlp.fc2.weight"), g("mlp.fc2.bias"))
    return x + z, p


def pos_encoding(pos_embed, n_patch_tokens, w, h, patch_size):
    """models/vision_transformer.py:187-207 interpolate_pos_encoding (bicubic, +0.1 trick)."""
    n0 = pos_embed.shape[1] - 1
    if n_patch_tokens == n0 and w == h:
        return pos_embed
    dim = pos_embed.shape[-1]
    w0, h0 = w // patch_size + 0.1, h // patch_size + 0.1
    s = int(math.sqrt(n0))
    grid = pos_embed[:, 1:].reshape(1, s, s, dim).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, scale_factor=(w0 / math.sqrt(n0), h0 / math.sqrt(n0)), mode="bicubic")
    assert int(w0) == grid.shape[-2] and int(h0) == grid.shape[-1]
    return torch.cat((pos_embed[:, :1], grid.permute(0, 2, 3, 1).reshape(1, -1, dim)), dim=1)


def prepare_tokens(img, sd, patch_size):
    """visual_transformer_det.py:192-214: conv patch embed, cls, pos-embed, point tokens."""
    B, _, w, h = img.shape
    x = F.conv2d(img, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=patch_size)
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat((sd["cls_token"].expand(B, -1, -1), x), dim=1)
    x = x + pos_encoding(sd["pos_embed"], x.shape[1] - 1, w, h, patch_size)
    pt = (sd["point_token"] + sd["point_pos_embed"]).expand(B, -1, -1)
    return torch.cat((x, pt), dim=1)


def backbone_forward(img, sd, *, patch_size, depth, num_heads, out_indices, point_tokens_num,
                     with_fpn=True, bn_eps=1e-5, trace=None):
    """visual_transformer_det.py:221-275 VisionTransformerDet.forward in eval mode
    (return_attention=True, last_feat=True, with_point_head=True, patch_size 16 FPN).
    trace (a list): receives the token tensor in front of every block and behind the last one (depth + 1 entries), for
    per-block ("teacher-forced") comparisons."""
    B, _, H, W = img.shape
    Hp, Wp = H // patch_size, W // patch_size
    T = point_tokens_num
    x = prepare_tokens(img, sd, patch_size)
    feats, attns = [], []
    for i in range(depth):
        if trace is not None:
            trace.append(x.clone())
        x, p = block(x, sd, f"blocks.{i}.", num_heads)
        attns.append(p.mean(1))
        if i in out_indices:
            feats.append(x[:, 1:, :][:, :-T].permute(0, 2, 1).reshape(B, -1, Hp, Wp).contiguous())
    if trace is not None:
        trace.append(x.clone())
    last_feat = x[:, :-T]
    org = torch.stack(feats, dim=1)
    if with_fpn:
        def fpn1(f):
            f = F.conv_transpose2d(f, sd["fpn1.0.weight"], sd["fpn1.0.bias"], stride=2)
            f = F.batch_norm(f, sd["fpn1.1.running_mean"], sd["fpn1.1.running_var"],
                             sd["fpn1.1.weight"], sd["fpn1.1.bias"], False, 0.0, bn_eps)
            return F.conv_transpose2d(F.gelu(f), sd["fpn1.3.weight"], sd["fpn1.3.bias"], stride=2)

        ops = [fpn1, lambda f: F.conv_transpose2d(f, sd["fpn2.0.weight"], sd["fpn2.0.bias"], stride=2),
               lambda f: f, lambda f: F.max_pool2d(f, 2, 2)]
        feats = [ops[i](f) for i, f in enumerate(feats)]
    pt = x[:, -T:]

    def mlp(z, name):
        for j in range(3):
            z = F.linear(z, sd[f"{name}.layers.{j}.weight"], sd[f"{name}.layers.{j}.bias"])
            if j < 2:
                z = F.relu(z)
        return z

    return dict(org_feats=org, feature=tuple(feats), point_tokens=pt,
                outputs_class=mlp(pt, "class_embed"), outputs_coord=mlp(pt, "bbox_embed").sigmoid(),
                attns=attns, last_feat=last_feat)


def rollout_full(attn_list):
    """stdroi:1257-1272 attns_project_to_feature.  attn_list: Lc x [B,N,N] head-mean softmax.
    Returns [B, Lc, N, N]; index k along dim 1 = product of the TOP k+1 layers."""
    a = torch.stack(attn_list)
    n = a.shape[-1]
    aug = a + torch.eye(n, dtype=a.dtype)
    aug = aug / aug.sum(-1, keepdim=True)
    out, run = [], None
    for l in range(len(attn_list) - 1, -1, -1):
        run = aug[l] if run is None else run @ aug[l]
        out.append(run)
    return torch.stack(out, dim=1)


def rollout_rows(attn_list, num_point_tokens):
    """Row-sliced equivalent used by the HIP path: only rows [-T:] of every partial product are
    ever consumed (stdroi:2272), and row slicing commutes with the left-multiplication chain.
    Returns [B, Lc, T, N]."""
    T = num_point_tokens
    a = torch.stack(attn_list)
    n = a.shape[-1]
    aug = a + torch.eye(n, dtype=a.dtype)
    aug = aug / aug.sum(-1, keepdim=True)
    out, run = [], None
    for l in range(len(attn_list) - 1, -1, -1):
        run = aug[l][:, -T:, :] if run is None else run @ aug[l]
        out.append(run)
    return torch.stack(out, dim=1)


# --------------------------------------------------------------------------------------
# Part B1: CAM -> connected components -> box   (stdroi:60-116, 2272-2294)
# --------------------------------------------------------------------------------------
def ccl_labels(binary):
    """8-connectivity connected components of a [H,W] (or [M,H,W]) 0/1 array.
    label = 1 + min raster index of the component, background 0, int32.
    (cc_torch call site stdroi:68; numbering is ours, see module docstring.)"""
    from scipy import ndimage

    a = np.asarray(binary).astype(np.uint8)
    if a.ndim == 3:
        return np.stack([ccl_labels(m) for m in a])
    lab, n = ndimage.label(a, structure=np.ones((3, 3), dtype=np.int32))
    if n == 0:
        return np.zeros(a.shape, dtype=np.int32)
    flat = lab.ravel()
    first = np.full(n + 1, flat.size, dtype=np.int64)
    np.minimum.at(first, flat, np.arange(flat.size, dtype=np.int64))
    return np.where(lab > 0, first[lab] + 1, 0).astype(np.int32)


def cam_box(cam, point, cam_thr=0.2, area_ratio=0.5, img_size=None):
    """stdroi:60-116 get_bbox_from_cam_fast, box_method='expand'.
    cam [H,W] fp32 (not modified), point (x,y).  Returns (box [4] fp32, kept mask [H,W] bool).
    A CAM with no foreground makes the reference raise (torch.stack of an empty list,
    stdroi:80); here that case returns the unreachable fallback box [0,0,1,1] and an empty mask
    so callers can decide (the product path reports it through a status word)."""
    img_h, img_w = img_size
    cam = (cam - cam.min()) / (cam.max() - cam.min()).clamp(1e-6)
    binary = (cam >= cam_thr)
    lab = torch.from_numpy(ccl_labels(binary.numpy()))
    ids, areas = torch.unique(lab[lab > 0], return_counts=True)
    if ids.numel() == 0:
        return cam.new_tensor([0, 0, 1, 1]), torch.zeros_like(binary)
    keep_ids = ids[areas.float() >= area_ratio * areas.max().float()]
    kept = torch.isin(lab, keep_ids)
    ys, xs = torch.nonzero(kept, as_tuple=True)
    xmin, xmax = xs.min().float(), xs.max().float()
    ymin, ymax = ys.min().float(), ys.max().float()
    xc, yc = point[0].float(), point[1].float()

    def grow(c, lo, hi, limit):
        if (c - lo).abs() > (c - hi).abs():      # low side is the farther one: mirror it
            new_hi = c * 2 - lo
            return lo, (new_hi if new_hi < limit else torch.tensor(float(limit)))
        new_lo = c * 2 - hi
        return (new_lo if new_lo > 0 else torch.tensor(0.0)), hi

    bx0, bx1 = grow(xc, xmin, xmax, img_w)
    by0, by1 = grow(yc, ymin, ymax, img_h)
    return torch.stack([bx0, by0, bx1, by1]).float(), kept


def cam_boxes_from_rollout(cams_lowres, points, cam_thr, area_ratio, up=16):
    """stdroi:2272-2294 for one image: cams_lowres [Lc,G,Hp,Wp] (rollout rows of the matched
    point tokens), points [G,2].  Returns (boxes [G,Lc,4], upsampled cams [Lc,G,H,W])."""
    Lc, G, Hp, Wp = cams_lowres.shape
    H, W = Hp * up, Wp * up
    cams = upsample_bilinear(cams_lowres, H, W)
    boxes = torch.zeros(G, Lc, 4)
    for l in range(Lc):
        for g in range(G):
            boxes[g, l], _ = cam_box(cams[l, g], points[g], cam_thr, area_ratio, (H, W))
    return boxes, cams


# --------------------------------------------------------------------------------------
# Part B2: cosine-affinity refinement   (stdroi:1000-1019 and callees)
# --------------------------------------------------------------------------------------
def minmax_maps(a):
    """stdroi:329-333 norm_attns: per-map min-max (no clamp)."""
    flat = a.flatten(1)
    lo, hi = flat.min(1)[0][:, None, None], flat.max(1)[0][:, None, None]
    return (a - lo) / (hi - lo)


def sample_points(maps, num_points, thr, is_pos, gt_points=None):
    """stdroi:343-371 sample_point_grid.  Draws from torch's global CPU generator exactly as the
    reference does (`torch.randint(n, shape)`), so equal seeds give equal samples.
    Returns [G, num_points, 2] (x, y) long."""
    out = []
    for g, m in enumerate(maps):
        factor = 1.0
        coords = ((m >= thr) if is_pos else (m < thr)).nonzero()
        n = coords.shape[0]
        if n < num_points:
            if is_pos:
                out.append(torch.cat((coords, gt_points[g].repeat(num_points - n, 1)), dim=0))
                continue
            while n < num_points:
                factor *= 2
                coords = (m < thr * factor).nonzero()
                n = coords.shape[0]
        n_draw = torch.arange(0, n, step=n // num_points).shape[0]
        pick = torch.randint(n, (n_draw,)) % n
        out.append(coords[pick][:num_points])
    return torch.stack(out).flip(-1)


def seed_features(point_xy, feat, stride=16):
    """stdroi:335-338: mean ViT feature at the patches under the sampled pixels.
    point_xy [G', K, 2] (x,y); feat [C,Hp,Wp] -> [G', C]."""
    C, Hp, Wp = feat.shape
    py = (point_xy[..., 1].long() // stride).clamp(0, Hp)
    px = (point_xy[..., 0].long() // stride).clamp(0, Wp)
    f = feat.permute(1, 2, 0)[py, px]            # [G', K, C]
    return f.mean(dim=1)


def refined_similarity(point_xy, feat, boxes, refine_times, tau, is_select):
    """stdroi:668-707 get_refined_similarity.  feat [C,Hp,Wp], boxes [G,4] pixel coords.
    Returns (maps [R+1, G', Hp, Wp], refined seed features [G', C])."""
    C, Hp, Wp = feat.shape
    G = boxes.shape[0]
    tokens = feat.flatten(1).t()                                    # [Np, C]
    seeds = seed_features(point_xy, feat)
    cur = cos_matrix(seeds, tokens).reshape(-1, Hp, Wp)
    inbox = box_mask(boxes // 16, (Hp, Wp), 0.0)

    def select(m):
        m = m.clone()
        m[:G] = m[:G] * inbox
        win = m.argmax(0, keepdim=True)
        own = torch.arange(m.shape[0])[:, None, None]
        return torch.where(win == own, m, torch.zeros_like(m))

    maps = [select(cur) if is_select else cur.clone()]
    work = cur.clone()                      # NB reference refines the UNMASKED first map
    for _ in range(refine_times):
        peak = work.flatten(1).max(1)[0][:, None, None]
        w = torch.where(work < peak * tau, torch.zeros_like(work), work)
        seeds = (w.flatten(1) @ tokens) / w.flatten(1).sum(1, keepdim=True).clamp(1e-8)
        work = cos_matrix(seeds, tokens).reshape(-1, Hp, Wp)
        if is_select:
            work = work.clone()
            work[:G] = work[:G] * inbox
            win = work.argmax(0, keepdim=True)
            own = torch.arange(work.shape[0])[:, None, None]
            maps.append(torch.where(win == own, work, torch.zeros_like(work)))
        else:
            maps.append(work.clone())
    return torch.stack(maps), seeds


def cosine_refined_maps(attn_maps, feat, boxes, points_fg, points_bg, refine_times, obj_tau):
    """stdroi:1000-1019 get_cosine_similarity_refined_map AFTER sampling (the sampled points are
    inputs so both implementations see the same draws).  attn_maps [G,H,W] only gives the output
    size.  points_fg [G+1,K,2] (objects + shared background group), points_bg [G,K,2].
    Returns map_fg [R+1,G,H,W], map_bg [R+1,G,H,W], fg_feat [G+1,C], bg_feat [G,C]."""
    G, H, W = attn_maps.shape
    sim_fg, fg_feat = refined_similarity(points_fg, feat, boxes, refine_times, obj_tau, True)
    sim_bg, bg_feat = refined_similarity(points_bg, feat, boxes, refine_times, obj_tau, False)
    up_fg = upsample_bilinear(sim_fg, H, W)[:, :G]
    up_bg = upsample_bilinear(sim_bg, H, W)
    ret = (1 - up_bg) * up_fg
    peak = ret.flatten(-2).max(-1)[0][..., None, None].clamp(1e-8)

    def unit_peak(m):                       # stdroi:1037-1040 normalize_map
        return m / (m.flatten(-2).max(-1)[0][..., None, None] + 1e-8)

    nb, nf = unit_peak(up_bg), unit_peak(ret)
    bg = nb + (1 - (nf * 0.5 + nb * 0.5))   # stdroi:1042-1046 decouple_instance
    peak_bg = bg.flatten(-2).max(-1)[0][..., None, None].clamp(1e-8)
    return ret / peak, bg / peak_bg, fg_feat, bg_feat


def sample_refine_inputs(attn_maps, gt_points, num_points=20, thr_pos=0.2, thr_neg=0.1):
    """stdroi:1002-1007: the three sample_point_grid calls in the reference's order."""
    nm = minmax_maps(attn_maps)
    bg = sample_points(nm, num_points, thr_neg, False)
    fg = sample_points(nm, num_points, thr_pos, True, gt_points)
    supp = sample_points(nm.mean(0, keepdim=True), num_points, thr_neg, False)
    return torch.cat((fg, supp), dim=0), bg


def mask_points_fg_bg(map_fg, map_bg, pos_thr, neg_thr, num_gt, corr_size):
    """stdroi:433-461 on one box crop; draws torch.randperm from the global CPU generator.
    Returns (coords [num_gt,2] (y,x) long or -1 float, labels [num_gt] bool)."""
    pos = erode((map_fg > map_fg.max() * pos_thr).float(), corr_size).nonzero()
    neg = (map_bg > map_bg.max() * neg_thr).nonzero()
    both = torch.cat((pos, neg), dim=0)
    lab = torch.cat((torch.ones(pos.shape[0], dtype=torch.bool), torch.zeros(neg.shape[0], dtype=torch.bool)))
    pick = torch.randperm(both.shape[0])[:num_gt]
    if pick.shape[0] < num_gt:
        if pick.shape[0] == 0:
            return -torch.ones(num_gt, 2), torch.zeros(num_gt, dtype=torch.bool)
        pick = fill_in(pick, num_gt)
    return both[pick], lab[pick]


def mask_sample_points(map_fg_last, map_bg_last, rois, pos_thr, neg_thr, num_gt, corr_size):
    """stdroi:1980-1993: per-object crop + point sampling; returns coords [G,num_gt,2] (x,y) float,
    labels [G,num_gt]."""
    cs, ls = [], []
    for g in range(map_fg_last.shape[0]):
        x0, y0, x1, y1 = rois[g].int().tolist()
        c, l = mask_points_fg_bg(map_fg_last[g][y0:y1, x0:x1], map_bg_last[g][y0:y1, x0:x1],
                                 pos_thr, neg_thr, num_gt, corr_size)
        c = c.clone()
        c[:, 0] += y0
        c[:, 1] += x0
        cs.append(c.flip(1))
        ls.append(l)
    return torch.stack(cs).float(), torch.stack(ls)


# --------------------------------------------------------------------------------------
# Part B3/B4: semantic centres: seeds, mean-shift token clustering  (stdroi:1995-2031, 1778-1840)
# --------------------------------------------------------------------------------------
def semantic_prestage(map_fg, map_bg, patch_hw, pos_thr):
    """stdroi:2011-2020: erode(11) the thresholded fg map at full res, bilinear DOWN to the patch
    grid.  Returns (fg_inter [G,Hp,Wp], bg_inter [1,Hp,Wp], map_fg_bin [G,Hp,Wp])."""
    core = erode((map_fg > pos_thr).float(), 11)
    fg_inter = F.interpolate(core[None], patch_hw, mode="bilinear")[0]
    bg_inter = F.interpolate(map_bg[None].max(dim=1, keepdim=True)[0], patch_hw, mode="bilinear")[0]
    return fg_inter, bg_inter, (fg_inter > pos_thr).float()


def grid_seed_coords(maps, rois, thr=0.35, n_points=20):
    """stdroi:1784-1810: n_points grid-strided positive patch coords (y,x) per object."""
    out = []
    for g, m in enumerate(maps):
        pos = (m >= thr).nonzero()
        n = pos.shape[0]
        if n >= n_points:
            c = pos[torch.arange(0, n, step=n // n_points)[:n_points]]
        elif n > 0:
            c = fill_in(pos, n_points)
        else:
            c = ((rois[g][:2] + rois[g][2:]) // (2 * 16)).long().view(1, 2).flip(1).repeat(n_points, 1)
        out.append(c)
    return torch.stack(out)


def update_density(prot, feats, onehot, faithful=False):
    """stdroi:882-908 update_density_batch -> tau [G,P,1]."""
    sim = cos_broadcast(prot, feats) if faithful else cos_matrix(prot, feats)
    cnt = onehot.sum(-1)
    dens = (sim * onehot).sum(-1)
    dens = 1 - torch.where(cnt >= 1, dens / cnt, torch.zeros_like(dens))
    return dens.clamp(1e-10).unsqueeze(-1)


def cosine_shift(prot, feats, feats_org, tau=0.1, temp=0.1, n_shift=5, trace=None, faithful=False):
    """stdroi:830-854 cosine_shift_batch.  prot [G,P,C]; feats [G,Np,C] (zero outside each
    object's box); feats_org [Np,C].  Returns (prot [G*P,C], sim [G*P,Np]).
    `trace`, if a list, receives (assign [G,Np] long, tau [G,P]) per iteration.
    faithful=True evaluates every cosine in the reference's broadcast form (cos_broadcast): bit-identical to
    the reference on the same host, ~6 s at 64x64x768; the default normalised matmul agrees to fp32 round-off."""
    G, P, _ = prot.shape
    cosf = cos_broadcast if faithful else cos_matrix
    for _ in range(n_shift):
        sim = cosf(prot, feats)
        w = F.softmax(sim / (temp * tau), dim=-1)
        win = w.argmax(1, keepdim=True)                               # [G,1,Np], ties -> lowest p
        onehot = (torch.arange(P)[None, :, None] == win).to(w.dtype)
        prot = (w * onehot) @ feats
        tau = update_density(prot, feats, onehot, faithful)
        if trace is not None:
            trace.append((win[:, 0].clone(), tau[..., 0].clone()))
    sim = cosf(prot, feats_org)
    return prot.flatten(0, 1), sim.flatten(0, 1)


def cosine_shift_step(prot, feats, tau, temp=0.1, faithful=False):
    """ONE iteration of cosine_shift_batch (stdroi:833-841) from a given state: prot [G,P,C], tau scalar or [G,P,1].
    Returns dict(sim, w, win [G,Np], prot [G,P,C], tau [G,P,1]).  Used by the full-size parity tests to check every
    iteration of the HIP kernel from the kernel's OWN state, so that one flipped near-tie cannot compound."""
    G, P, _ = prot.shape
    cosf = cos_broadcast if faithful else cos_matrix
    sim = cosf(prot, feats)
    w = F.softmax(sim / (temp * tau), dim=-1)
    win = w.argmax(1, keepdim=True)
    onehot = (torch.arange(P)[None, :, None] == win).to(w.dtype)
    new = (w * onehot) @ feats
    return dict(sim=sim, w=w, win=win[:, 0], prot=new, tau=update_density(new, feats, onehot, faithful))


def shift_log_weights64(prot, feats, tau, temp=0.1):
    """float64 log softmax weights log w[g,p,n] of one iteration (stdroi:833-834) -- the exact-arithmetic value of what
    the fp32 reference evaluates; |log w[a] - log w[b]| is the margin by which prototype a beats b for patch n."""
    prot, feats = prot.double(), feats.double()
    tau = torch.as_tensor(tau, dtype=torch.float64)
    sim = cos_matrix(prot, feats)
    logit = sim / (temp * tau)
    return logit - torch.logsumexp(logit, dim=-1, keepdim=True)


def mean_shift_prototypes(maps, feat, rois, n_shift, thr=0.35, tau=0.1, temp=0.1, n_points=20, trace=None,
                          faithful=False):
    """stdroi:1778-1840 mean_shift_grid_prototype (rois given).  maps [G,Hp,Wp] binary,
    feat [C,Hp,Wp].  Returns (prot [G*P,C], sim [G*P,Hp,Wp] clamped at 0, seed coords [G,P,2])."""
    C, Hp, Wp = feat.shape
    coords = grid_seed_coords(maps, rois, thr, n_points)
    tokens = feat.flatten(1).t()
    prot = feat.permute(1, 2, 0)[coords[..., 0], coords[..., 1]].clone()          # [G,P,C]
    inbox = box_mask(rois // 16, (Hp, Wp), 0.0).flatten(1)                        # [G,Np]
    prot, sim = cosine_shift(prot, tokens[None] * inbox[..., None], tokens, tau, temp, n_shift, trace, faithful)
    return prot, sim.reshape(-1, Hp, Wp).clamp(0), coords


def filter_parts(sim, fg_inter, pos_thr=0.85):
    """stdroi:265-275 filter_maps: keep prototypes whose (sim>0.8) support overlaps fg >= 0.85.
    sim [G,P,Hp,Wp].  Returns (list of kept maps per object, keep mask [G,P])."""
    support = (sim > 0.8).to(sim.dtype)
    score = (fg_inter[:, None] * support).sum(dim=[-2, -1]) / support.sum(dim=[-2, -1]).clamp(1e-6)
    keep = score >= pos_thr
    kept = [sim[g][keep[g]] for g in range(sim.shape[0])]
    return kept, keep


def merge_parts(prot_list, thr):
    """stdroi:278-294 merge_maps: greedy upper-triangular merge of near-duplicate prototypes."""
    out = []
    for prot in prot_list:
        if prot.shape[0] == 0:
            out.append([])
            continue
        sim = cos_matrix(prot, prot).t()          # reference layout: sim[i,j] = cos(prot[i], prot[j])
        link = (torch.triu(sim, diagonal=0) >= thr).to(prot.dtype)
        merged = []
        for i in range(link.shape[0]):
            wgt = link[i].clone()
            if wgt.sum() > 0:
                merged.append((wgt @ prot) / (wgt.sum() + 1e-8))
            link[wgt > 0] *= 0
        out.append(torch.stack(merged))
    return out


def part_similarity(prot, feat_hwc):
    """stdroi:297-301 cal_similarity: [k,C] x [Hp,Wp,C] -> [k,Hp,Wp]; [] -> zeros(0,0)."""
    if isinstance(prot, list):
        return torch.zeros(0, 0)
    Hp, Wp, C = feat_hwc.shape
    return cos_matrix(prot, feat_hwc.reshape(-1, C)).reshape(-1, Hp, Wp)


def part_centers(maps, rois, obj_label, feat, num_max_keep=50, num_max_obj=3):
    """stdroi:222-262 get_center_coord_with_feat.  Returns dict with the eight reference outputs."""
    coords, labels, feats, owner = [], [], [], []
    split = [0] * len(maps)
    for g, m in enumerate(maps):
        if m.shape[0] == 0:
            continue
        peak = m.flatten(1).topk(dim=1, k=1)[0][:, -1, None, None]
        at_peak = (m >= peak).nonzero().float()
        x0, y0, x1, y1 = rois[g]
        order = (m > 0.9).sum(dim=[-2, -1]).argsort(descending=True, dim=0)
        for i in range(m.shape[0]):
            if i > num_max_obj:
                break
            yx = at_peak[at_peak[:, 0] == order[i]].mean(dim=0)[1:]
            xy = yx.flip(0)
            c = (xy + 0.5) * 16
            if (c[0] >= x0) & (c[0] <= x1) & (c[1] >= y0) & (c[1] <= y1):
                coords.append(c)
                labels.append(obj_label[g])
                owner.append(g)
                feats.append(feat[:, xy[1].long(), xy[0].long()])
                split[g] += 1
    if not coords:
        z2 = torch.zeros(0, 2, dtype=rois[0].dtype)
        zl = torch.zeros(0, dtype=obj_label[0].dtype)
        return dict(centers=[z2, zl], split=[], feat_split=[], feats=[], num_parts=split,
                    coords_org=z2.clone(), labels_org=zl.clone(), corres_gt=torch.zeros(0, dtype=torch.long))
    coords, labels, feats = torch.stack(coords), torch.stack(labels), torch.stack(feats)
    res = dict(coords_org=coords.clone(), labels_org=labels.clone(), feats=feats,
               split=list(coords.split(split, dim=0)), feat_split=list(feats.split(split, dim=0)),
               num_parts=split, corres_gt=torch.tensor(owner, dtype=torch.long))
    if coords.shape[0] > num_max_keep:
        pick = torch.randperm(coords.shape[0])[:num_max_keep]
        coords, labels = coords[pick], labels[pick]
    res["centers"] = [coords, labels]
    return res


def semantic_centers(map_fg, map_bg, rois, feat, pos_thr, n_shift, gt_labels, merge_thr=0.85,
                     num_semantic_points=3, trace=None, faithful=False):
    """stdroi:1995-2031 get_semantic_centers for one image."""
    C, Hp, Wp = feat.shape
    fg_inter, bg_inter, fg_bin = semantic_prestage(map_fg, map_bg, (Hp, Wp), pos_thr)
    prot, sim, seeds = mean_shift_prototypes(fg_bin, feat, rois, n_shift, trace=trace, faithful=faithful)
    G = map_fg.shape[0]
    P = sim.shape[0] // G
    kept_maps, keep = filter_parts(sim.reshape(G, P, Hp, Wp), fg_inter)
    counts = keep.sum(-1).tolist()
    merged = merge_parts(list(prot[keep.flatten()].split(counts, dim=0)), merge_thr)
    sim_parts = [part_similarity(p, feat.permute(1, 2, 0)) for p in merged]
    res = part_centers(sim_parts, rois, gt_labels, feat, num_max_obj=num_semantic_points)
    res.update(sim_parts=sim_parts, prot=prot, sim=sim, seeds=seeds, keep=keep,
               fg_inter=fg_inter, bg_inter=bg_inter)
    return res


def pseudo_masks(map_fg_last, pos_thr):
    """stdroi:2356-2358: (map > rowmax * thr) as uint8 numpy."""
    peak = map_fg_last.flatten(1).max(1)[0][:, None, None]
    return (map_fg_last > peak * pos_thr).to(torch.uint8).numpy()


# ---------------------------------------------------------------------------------------------------------
# A6 (BASELINE config 5): Swin window attention.  Restates models/swin_transformer.py
# ---------------------------------------------------------------------------------------------------------
def swin_window_partition(x, ws):
    """models/swin_transformer.py:45-57: [B,H,W,C] -> [nW*B, ws, ws, C]."""
    B, H, W, C = x.shape
    return x.view(B, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C)


def swin_window_reverse(windows, ws, H, W):
    """models/swin_transformer.py:60-74."""
    B = int(windows.shape[0] / (H * W / ws / ws))
    return windows.view(B, H // ws, W // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H, W, -1)


def swin_relative_position_index(ws):
    """models/swin_transformer.py:120-130."""
    coords = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def swin_attn_mask(H, W, ws, shift):
    """SwinTransformerBlock.create_attn_mask, models/swin_transformer.py:233-256: [nW, ws*ws, ws*ws] of 0 / -100."""
    Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
    img = torch.zeros(1, Hp, Wp, 1)
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[:, hs, wsl, :] = cnt
            cnt += 1
    mw = swin_window_partition(img, ws).view(-1, ws * ws)
    m = mw.unsqueeze(1) - mw.unsqueeze(2)
    return m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0)


def swin_window_attention(xw, p, num_heads, ws, mask=None):
    """WindowAttention.forward, models/swin_transformer.py:125-157 -> (out [B_,N,C], attn [B_,h,N,N]).
    p: dict with qkv.weight/bias, proj.weight/bias, relative_position_bias_table."""
    B_, N, C = xw.shape
    d = C // num_heads
    qkv = F.linear(xw, p["qkv.weight"], p.get("qkv.bias")).reshape(B_, N, 3, num_heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * d ** -0.5, qkv[1], qkv[2]
    attn = q @ k.transpose(-2, -1)
    bias = p["relative_position_bias_table"][swin_relative_position_index(ws).view(-1)].view(N, N, -1).permute(2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        attn = (attn.view(B_ // nW, nW, num_heads, N, N) + mask.unsqueeze(1).unsqueeze(0)).view(-1, num_heads, N, N)
    attn = attn.softmax(dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(B_, N, C)
    return F.linear(out, p["proj.weight"], p["proj.bias"]), attn


def swin_block(x, p, num_heads, ws, shift, ln_eps=1e-5):
    """SwinTransformerBlock.forward, models/swin_transformer.py:259-314 (drop_path identity) -> (x [B,L,C], attn).
    p: norm1.*, attn.qkv.*, attn.proj.*, attn.relative_position_bias_table, norm2.*, mlp.fc1.*, mlp.fc2.*"""
    B, L, C = x.shape
    H = W = int(math.sqrt(L))
    shortcut = x
    y = F.layer_norm(x, (C,), p["norm1.weight"], p["norm1.bias"], ln_eps).view(B, H, W, C)
    pad_r, pad_b = (ws - W % ws) % ws, (ws - H % ws) % ws
    y = F.pad(y, (0, 0, 0, pad_r, 0, pad_b))
    Hp, Wp = y.shape[1], y.shape[2]
    mask = None
    if shift > 0:
        y = torch.roll(y, shifts=(-shift, -shift), dims=(1, 2))
        mask = swin_attn_mask(H, W, ws, shift)
    xw = swin_window_partition(y, ws).view(-1, ws * ws, C)
    ap = {k[len("attn."):]: v for k, v in p.items() if k.startswith("attn.")}
    aw, attn = swin_window_attention(xw, ap, num_heads, ws, mask)
    y = swin_window_reverse(aw.view(-1, ws, ws, C), ws, Hp, Wp)
    if shift > 0:
        y = torch.roll(y, shifts=(shift, shift), dims=(1, 2))
    y = y[:, :H, :W, :].contiguous().view(B, H * W, C)
    x = shortcut + y
    z = F.layer_norm(x, (C,), p["norm2.weight"], p["norm2.bias"], ln_eps)
    z = F.linear(F.gelu(F.linear(z, p["mlp.fc1.weight"], p["mlp.fc1.bias"])), p["mlp.fc2.weight"], p["mlp.fc2.bias"])
    return x + z, attn


# ---------------------------------------------------------------------------------------------------------
# mmcv-full 1.3.8 compiled ops used next to the path: RoIAlign and NMS.
#
# mmcv is a third-party dependency that is ABSENT from /root/reference (install.sh pins `mmcv-full==1.3.8`); its ops
# cannot be executed here, so these are restatements of its PUBLISHED algorithm (mmcv/ops/csrc/pytorch/cpu/
# roi_align.cpp `ROIAlignForward` / `ROIAlignBackward` + `pre_calc_for_bilinear_interpolate`, pool_mode 'avg';
# mmcv/ops/csrc/pytorch/cpu/nms.cpp `nms_cpu`; mmcv/ops/nms.py `batched_nms`; mmdet/core/post_processing/bbox_nms.py
# `multiclass_nms`), written as the scalar loops of that source -- deliberately NOT sharing a line of arithmetic with
# attentionshift_amd/mil_head.roi_align (separable tensor ops) or attentionshift_amd/inference.nms (IoU matrix).
# They are anchored on the reference's call sites: configs/mae/attnshift_voc12aug.py:64-68,123-127 (RoIAlign 7x7 /
# 14x14, sampling_ratio=0, featmap stride 16; mmdet's SingleRoIExtractor builds the layer with aligned=True, its
# default), stdroi:2958 (MIL RoI features), stdroi:3192-3221 + attnshift_voc12aug.py:200-204 (test-time NMS:
# score_thr 0.05, IoU 0.5, 100 per image), and on closed-form cases in tests/test_oracle_roi_nms.py.  "parity
# unpinned" in the sense of the task statement: no golden vector from mmcv itself exists in the reference tree.
# ---------------------------------------------------------------------------------------------------------
def _mmcv_bilinear_terms(height, width, y, x):
    """One sample point -> ((y_low, x_low, y_high, x_high), (w1, w2, w3, w4)) or None when the point lies outside
    [-1, size] (roi_align.cpp bilinear_interpolate / pre_calc_for_bilinear_interpolate: contributes 0)."""
    if y < -1.0 or y > height or x < -1.0 or x > width:
        return None
    y = max(y, 0.0)
    x = max(x, 0.0)
    y_low, x_low = int(y), int(x)
    if y_low >= height - 1:
        y_high = y_low = height - 1
        y = float(y_low)
    else:
        y_high = y_low + 1
    if x_low >= width - 1:
        x_high = x_low = width - 1
        x = float(x_low)
    else:
        x_high = x_low + 1
    ly, lx = y - y_low, x - x_low
    hy, hx = 1.0 - ly, 1.0 - lx
    return (y_low, x_low, y_high, x_high), (hy * hx, hy * lx, ly * hx, ly * lx)


def _mmcv_roi_geometry(roi, spatial_scale, pooled, sampling_ratio, aligned):
    offset = 0.5 if aligned else 0.0
    x1, y1, x2, y2 = (float(v) * spatial_scale - offset for v in roi[1:5])
    roi_w, roi_h = x2 - x1, y2 - y1
    if not aligned:                                      # legacy: force a minimum size of one pixel
        roi_w, roi_h = max(roi_w, 1.0), max(roi_h, 1.0)
    bin_h, bin_w = roi_h / pooled, roi_w / pooled
    grid_h = sampling_ratio if sampling_ratio > 0 else int(math.ceil(roi_h / pooled))
    grid_w = sampling_ratio if sampling_ratio > 0 else int(math.ceil(roi_w / pooled))
    count = max(grid_h * grid_w, 1)
    return x1, y1, bin_h, bin_w, grid_h, grid_w, count


def roi_align_mmcv(feat, rois, output_size, spatial_scale, sampling_ratio=0, aligned=True):
    """mmcv.ops.roi_align (pool_mode='avg').  feat [B,C,H,W] float, rois [R,5] = (batch index, x1, y1, x2, y2) in image
    coordinates -> [R,C,out,out] float64.  Scalar loops over RoIs, bins and sample points; vectorised over channels."""
    f = np.asarray(feat, dtype=np.float64)
    r = np.asarray(rois, dtype=np.float64)
    B, C, H, W = f.shape
    out = np.zeros((r.shape[0], C, output_size, output_size), dtype=np.float64)
    for n in range(r.shape[0]):
        b = int(r[n, 0])
        x1, y1, bin_h, bin_w, grid_h, grid_w, count = _mmcv_roi_geometry(r[n], spatial_scale, output_size, sampling_ratio, aligned)
        for ph in range(output_size):
            for pw in range(output_size):
                acc = np.zeros(C, dtype=np.float64)
                for iy in range(grid_h):
                    y = y1 + ph * bin_h + (iy + 0.5) * bin_h / grid_h
                    for ix in range(grid_w):
                        x = x1 + pw * bin_w + (ix + 0.5) * bin_w / grid_w
                        terms = _mmcv_bilinear_terms(H, W, y, x)
                        if terms is None:
                            continue
                        (yl, xl, yh, xh), (w1, w2, w3, w4) = terms
                        acc += w1 * f[b, :, yl, xl] + w2 * f[b, :, yl, xh] + w3 * f[b, :, yh, xl] + w4 * f[b, :, yh, xh]
                out[n, :, ph, pw] = acc / count
    return out


def roi_align_mmcv_backward(grad_out, rois, feat_shape, spatial_scale, sampling_ratio=0, aligned=True):
    """ROIAlignBackward (avg): every sample point scatters grad / count with its four bilinear weights.
    grad_out [R,C,out,out] -> grad_feat [B,C,H,W] float64."""
    g = np.asarray(grad_out, dtype=np.float64)
    r = np.asarray(rois, dtype=np.float64)
    B, C, H, W = feat_shape
    pooled = g.shape[-1]
    df = np.zeros((B, C, H, W), dtype=np.float64)
    for n in range(r.shape[0]):
        b = int(r[n, 0])
        x1, y1, bin_h, bin_w, grid_h, grid_w, count = _mmcv_roi_geometry(r[n], spatial_scale, pooled, sampling_ratio, aligned)
        for ph in range(pooled):
            for pw in range(pooled):
                go = g[n, :, ph, pw] / count
                for iy in range(grid_h):
                    y = y1 + ph * bin_h + (iy + 0.5) * bin_h / grid_h
                    for ix in range(grid_w):
                        x = x1 + pw * bin_w + (ix + 0.5) * bin_w / grid_w
                        terms = _mmcv_bilinear_terms(H, W, y, x)
                        if terms is None:
                            continue
                        (yl, xl, yh, xh), (w1, w2, w3, w4) = terms
                        df[b, :, yl, xl] += w1 * go
                        df[b, :, yl, xh] += w2 * go
                        df[b, :, yh, xl] += w3 * go
                        df[b, :, yh, xh] += w4 * go
    return df


def nms_mmcv(boxes, scores, iou_threshold, offset=0):
    """mmcv.ops.nms (nms_cpu): visit boxes by decreasing score (stable), keep a box unless suppressed, suppress every
    later box whose IoU with it EXCEEDS the threshold (`ovr > iou_threshold`; areas and intersections with `offset`,
    0 by default).  Returns the kept indices in decreasing-score order."""
    bx = np.asarray(boxes, dtype=np.float64).reshape(-1, 4)
    sc = np.asarray(scores, dtype=np.float64).reshape(-1)
    order = np.argsort(-sc, kind="stable")
    areas = (bx[:, 2] - bx[:, 0] + offset) * (bx[:, 3] - bx[:, 1] + offset)
    suppressed = np.zeros(bx.shape[0], dtype=bool)
    keep = []
    for _i in range(order.size):
        i = int(order[_i])
        if suppressed[i]:
            continue
        keep.append(i)
        for _j in range(_i + 1, order.size):
            j = int(order[_j])
            if suppressed[j]:
                continue
            w = max(0.0, min(bx[i, 2], bx[j, 2]) - max(bx[i, 0], bx[j, 0]) + offset)
            h = max(0.0, min(bx[i, 3], bx[j, 3]) - max(bx[i, 1], bx[j, 1]) + offset)
            inter = w * h
            ovr = inter / (areas[i] + areas[j] - inter) if inter > 0 else 0.0
            if ovr > iou_threshold:
                suppressed[j] = True
    return np.asarray(keep, dtype=np.int64)


def multiclass_nms_mmdet(multi_bboxes, multi_scores, score_thr, iou_threshold, max_num=-1):
    """mmdet/core/post_processing/bbox_nms.py `multiclass_nms` + mmcv.ops.batched_nms (class_agnostic=False): drop the
    background column, keep (box, class) pairs with score > score_thr, shift the boxes of class c by c * (max coordinate
    + 1) so that classes never overlap, one greedy NMS over all pairs, the best `max_num`.  Done here the direct way --
    per class, then merged by score -- which is what the offset trick computes.  -> (dets [m,5], labels [m])."""
    mb = np.asarray(multi_bboxes, dtype=np.float64)
    ms = np.asarray(multi_scores, dtype=np.float64)
    n, K = ms.shape[0], ms.shape[1] - 1
    dets, labels, flat_pos = [], [], []
    for c in range(K):
        bc = mb[:, 4 * c:4 * c + 4] if mb.shape[1] > 4 else mb
        idx = np.nonzero(ms[:, c] > score_thr)[0]
        if idx.size == 0:
            continue
        keep = nms_mmcv(bc[idx], ms[idx, c], iou_threshold)
        for k in keep:
            i = int(idx[k])
            dets.append(np.concatenate((bc[i], ms[i:i + 1, c])))
            labels.append(c)
            flat_pos.append(i * K + c)                   # position in mmdet's flattened (box-major) candidate list
    if not dets:
        return np.zeros((0, 5)), np.zeros(0, dtype=np.int64)
    dets, labels, flat_pos = np.stack(dets), np.asarray(labels, dtype=np.int64), np.asarray(flat_pos)
    # one global order by decreasing score; equal scores keep the candidate list's order (stable sort in nms)
    order = np.lexsort((flat_pos, -dets[:, 4]))
    if max_num > 0:
        order = order[:max_num]
    return dets[order], labels[order]
