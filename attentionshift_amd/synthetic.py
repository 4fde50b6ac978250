"""Seeded synthetic inputs for the hot path.

Everything is generated on the CPU with an explicit torch.Generator so the golden generator
(build container), the parity tests and bench.py (GPU box) all see the same tensors.
No compute of the path lives here: only input construction.
"""
import zlib

import torch


def det_tensor(name, shape, scale=0.02, offset=0.0):
    """Deterministic parameter tensor keyed by its state-dict name (no storage needed in fixtures)."""
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    return torch.randn(tuple(shape), generator=g) * scale + offset


def det_state_dict(shapes, scale=0.02):
    """shapes: {name: shape}.  Norm weights get 1+noise, running_var positive, BN counters zero."""
    sd = {}
    for name, shape in shapes.items():
        if name.endswith("num_batches_tracked"):
            sd[name] = torch.zeros((), dtype=torch.long)
        elif name.endswith("running_var"):
            sd[name] = det_tensor(name, shape, 0.1).abs() + 0.5
        elif (".norm" in name or name.startswith("fpn1.1")) and name.endswith("weight"):
            sd[name] = det_tensor(name, shape, 0.05, 1.0)
        elif name.endswith("bias") or name.endswith("running_mean"):
            sd[name] = det_tensor(name, shape, 0.05)
        elif "patch_embed" in name:
            sd[name] = det_tensor(name, shape, 0.05)
        elif "attn.qkv.weight" in name:
            sd[name] = det_tensor(name, shape, 0.08)     # non-uniform attention with random weights
        else:
            sd[name] = det_tensor(name, shape, scale)
    return sd


def images(batch, height, width, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(batch, 3, height, width, generator=g)


def lattice_boxes(num_obj, hp, wp):
    """Deterministic NON-overlapping patch-grid boxes [G,4] = (x0,y0,x1,y1) inclusive patch indices,
    side by side with a one-patch gap (objects that share features make the reference's argmax
    selection a rounding-noise coin flip, which no parity test can pin)."""
    out = []
    span = max((wp - 2) // max(num_obj, 1), 2)
    for g in range(num_obj):
        x0 = 1 + g * span
        y0 = int(0.2 * hp) + (g % 2)
        w, h = max(span - 1, 2), max(hp // 2, 2)
        out.append([x0, y0, min(x0 + w - 1, wp - 1), min(y0 + h - 1, hp - 1)])
    return torch.tensor(out, dtype=torch.long)


def shift_inputs(seed, hp, wp, channels, num_obj, cam_layers, stride=16, parts=2):
    """Inputs of the attention-shift stage for ONE image.

    vit_feat  [C,Hp,Wp]   1.5*mu_bg + N(0,1) background; inside object g's box `parts` vertical stripes, each
                          3*mu_{g,part} + 0.5*N(0,1)  (so the mean-shift finds >1 part per object)
    cams      [Lc,G,Hp,Wp] 0.1*U(0,1) + 1 inside a per-layer jittered copy of the box
    points    [G,2]       (x,y) pixel coords of the box centres
    boxes     [G,4]       pixel boxes (float), the box the MIL head would have selected
    labels    [G]         class ids
    """
    g = torch.Generator().manual_seed(seed)
    pb = lattice_boxes(num_obj, hp, wp)
    mu_bg = torch.randn(channels, generator=g)          # coherent background, like a real scene
    feat = 1.5 * mu_bg[:, None, None] + torch.randn(channels, hp, wp, generator=g)
    for o in range(num_obj):
        x0, y0, x1, y1 = pb[o].tolist()
        width = x1 - x0 + 1
        for p in range(parts):
            mu = torch.randn(channels, generator=g)
            xa = x0 + (width * p) // parts
            xb = x0 + (width * (p + 1)) // parts
            if xb <= xa:
                continue
            noise = torch.randn(channels, y1 - y0 + 1, xb - xa, generator=g)
            feat[:, y0:y1 + 1, xa:xb] = 3.0 * mu[:, None, None] + 0.5 * noise
    cams = 0.1 * torch.rand(cam_layers, num_obj, hp, wp, generator=g)
    for l in range(cam_layers):
        for o in range(num_obj):
            x0, y0, x1, y1 = pb[o].tolist()
            j = l % 3                                   # layer-dependent jitter so boxes differ
            xa, xb = min(x0 + j, x1), max(x1 - (j // 2), x0)
            cams[l, o, y0:y1 + 1, xa:xb + 1] += 1.0
            if l % 2 == 1 and y1 + 3 < hp:              # a detached small blob (area filter)
                cams[l, o, y1 + 2:y1 + 3, x0:x0 + 1] += 1.0
    boxes = torch.stack([pb[:, 0] * stride, pb[:, 1] * stride,
                         (pb[:, 2] + 1) * stride - 1, (pb[:, 3] + 1) * stride - 1], dim=1).float()
    points = torch.stack([(boxes[:, 0] + boxes[:, 2]) * 0.5, (boxes[:, 1] + boxes[:, 3]) * 0.5], dim=1)
    labels = torch.arange(num_obj) % 20
    return dict(vit_feat=feat, cams=cams, points=points, boxes=boxes, labels=labels, patch_boxes=pb)
