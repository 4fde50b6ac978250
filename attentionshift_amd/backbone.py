"""VisionTransformerDet -- MI355X-native MAE-ViT backbone with point tokens.

Mirror of the reference plugin (same registry name, constructor kwargs, state-dict keys and output
dict) so configs/mae/*.py build it unchanged:
    reference mmdet/models/backbones/visual_transformer_det.py:60-275 (VisionTransformerDet)
    reference models/vision_transformer.py:62-124 (Attention, Block), :187-207 (pos-embed resize)

What is different underneath:
  * Attention runs through the C ABI (QKV GEMM with fused head split -> flash SDPA with wave64 online
    softmax -> proj GEMM, attentionshift_amd/csrc/{gemm,sdpa}.hip).  The [B,h,N,N] softmax the reference
    returns from every block is never materialised; `attns` is a list of AttnLayerState handles
    (q, k, log-sum-exp) from which any head-mean attention row is recomputed on demand
    (ops.attn_mean_rows / ops.rollout_rows) -- the only consumer reads 100 rows of a product of 7 of them.
  * bf16 operands / fp32 accumulate by default (`compute_dtype`), fp32 master parameters; the reference
    runs apex O1 fp16.  `compute_dtype=torch.float32` is the exact-fp32 parity path.
  * LayerNorm / residuals are torch elementwise ops; the GELU-MLP, the patch embedding and the 2x2/2 FPN
    deconvolutions (one GEMM over tokens each) go through the same GEMM kernel (ops.linear).
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .registry import BACKBONES


def trunc_normal_(tensor, std=0.02):
    # reference utils.py:572 (timm-style truncated normal, +-2 std)
    return nn.init.trunc_normal_(tensor, std=std, a=-2 * std, b=2 * std)


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class Attention(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias):
        super().__init__()
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)


class Block(nn.Module):
    """Parameter container with the reference's key names (norm1, attn.qkv, attn.proj, norm2, mlp.fc1/fc2)."""

    def __init__(self, dim, num_heads, mlp_ratio, qkv_bias, eps):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = Attention(dim, num_heads, qkv_bias)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))


class PatchEmbed(nn.Module):
    def __init__(self, img_size, patch_size, in_chans, embed_dim):
        super().__init__()
        if isinstance(img_size, int):
            img_size = [img_size, img_size]
        self.num_patches = (img_size[0] // patch_size) * (img_size[1] // patch_size)
        self.patch_size = patch_size
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)


class MLP(nn.Module):
    """3-layer FFN of the point head (visual_transformer_det.py:26-38)."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))

    def forward(self, x):
        for i, layer in enumerate(self.layers):
            x = F.relu(layer(x)) if i < self.num_layers - 1 else layer(x)
        return x


class DeferredFPN:
    """The FPN maps of a no-grad forward whose kernels have not been queued yet (`VisionTransformerDet.defer_fpn`).

    launch()  queues the FPN of every tap on a side stream that first waits for the caller's stream (the taps);
    result()  makes the caller's stream wait for that side stream and returns the tuple of maps;
    len / iteration / indexing go through result(), so a consumer that does not know about the deferral reads correct maps
    (without the overlap).  The visual_transformer_det.py:246-256 arithmetic is `_fpn`, unchanged.

    Measured on the headline step (round 5): NOT a win on MI355X -- the 256 x 256 GEMM workgroups fill every CU's
    register file, so each small launch of the RoI head's chain waits for a GEMM workgroup to retire (7.43 -> 7.79 ms per
    step; with a fifth stream the four default hardware queues are over-subscribed and the two image chains serialise;
    with GPU_MAX_HW_QUEUES=8 the overlap is real and the step takes 8.4 ms -- also with 128 x 128 GEMM tiles that leave a
    third of every register file free, and whatever the stream priorities; a CU-masked stream is slower still).  It stays opt-in for callers whose head has
    real work to hide it under."""

    def __init__(self, backbone, features, taps):
        self._bb, self._features, self._taps = backbone, list(features), list(taps)
        self._out, self._event, self._joined = None, None, False

    def launch(self):
        if self._out is not None:
            return self
        bb = self._bb
        main = torch.cuda.current_stream()
        if bb._fpn_stream is None:
            bb._fpn_stream = torch.cuda.Stream()
        side = bb._fpn_stream
        side.wait_stream(main)
        with torch.cuda.stream(side):
            for t in self._features + self._taps:
                t.record_stream(side)                       # the taps were allocated on the caller's stream
            self._out = tuple(bb._fpn(i, self._features[i], self._taps[i]) for i in range(len(self._features)))
            self._event = side.record_event()
        return self

    def result(self):
        self.launch()
        if not self._joined:
            main = torch.cuda.current_stream()
            main.wait_event(self._event)
            for t in self._out:
                t.record_stream(main)
            self._joined = True
            self._features, self._taps = [], []
        return self._out

    def __len__(self):
        return len(self._features) if self._out is None else len(self._out)

    def __iter__(self):
        return iter(self.result())

    def __getitem__(self, i):
        return self.result()[i]


@BACKBONES.register_module()
class VisionTransformerDet(nn.Module):
    def __init__(self, img_size, patch_size, embed_dim, in_chans=3, with_fpn=True, frozen_stages=-1,
                 out_indices=(3, 5, 7, 11), use_checkpoint=False, learnable_pos_embed=True, last_feat=False,
                 recompute_last_feat=False, point_tokens_num=100, num_classes=20, return_attention=False,
                 with_point_head=True, depth=12, num_heads=12, mlp_ratio=4., qkv_bias=False, qk_scale=None,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0., norm_eps=1e-6, init_values=0,
                 compute_dtype=torch.bfloat16, **unused):
        super().__init__()
        assert not with_fpn or patch_size in (8, 16)
        assert not recompute_last_feat or last_feat
        if qk_scale is not None or attn_drop_rate or drop_rate or init_values:
            raise NotImplementedError("qk_scale / dropout / layer-scale are unused by every reference config")
        if embed_dim != num_heads * ops.HEAD_DIM:
            raise NotImplementedError(f"head dim must be {ops.HEAD_DIM}")
        self.embed_dim = self.num_features = embed_dim
        self.patch_size, self.depth, self.num_heads = patch_size, depth, num_heads
        self.last_feat, self.recompute_last_feat = last_feat, recompute_last_feat
        self.with_fpn, self.frozen_stages = with_fpn, frozen_stages
        self.out_indices = tuple(out_indices)
        self.use_checkpoint = use_checkpoint            # accepted; activations here are already O(N)
        self.drop_path_rate = drop_path_rate
        self.return_attention = return_attention
        self.point_tokens_num = point_tokens_num
        self.with_point_head = with_point_head
        self.compute_dtype = compute_dtype
        # no-grad forward: hand the FPN to the caller as a DeferredFPN (a schedule switch, same arithmetic; opt-in because
        # the maps of such a forward become valid on the caller's stream only through DeferredFPN.result())
        self.defer_fpn = bool(unused.pop("defer_fpn", False))
        self._fpn_stream = None
        self._point_pack = None
        # no-grad forward: the point head's seven small launches (200 tokens) on a side stream, next to whatever the caller
        # queues after the forward; opt-in for the same reason: outputs_class / outputs_coord of such a forward are valid on
        # the caller's stream only after it waits for out["point_head_ready"] (seed_pseudo_gt(point_ready=...) does).
        # MEASURED SLOWER (round 5, same box, three alternations): 6.50-6.59 ms per step in line, 6.81-6.95 ms on the side
        # stream -- like DeferredFPN, a second queue next to the RoI head's latency-bound chain costs more than the ~90 us
        # of device time it takes off the caller's stream.  bench.py keeps it off (AS_POINT_HEAD_STREAM=1 turns it on).
        self.point_head_stream = bool(unused.pop("point_head_stream", False))
        self._point_stream = None

        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        n_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n_patches + 1, embed_dim), requires_grad=learnable_pos_embed)
        self.blocks = nn.ModuleList(Block(embed_dim, num_heads, mlp_ratio, qkv_bias, norm_eps) for _ in range(depth))
        if with_fpn and patch_size == 16:
            self.fpn1 = nn.Sequential(nn.ConvTranspose2d(embed_dim, embed_dim, 2, 2), nn.BatchNorm2d(embed_dim), nn.GELU(),
                                      nn.ConvTranspose2d(embed_dim, embed_dim, 2, 2))
            self.fpn2 = nn.Sequential(nn.ConvTranspose2d(embed_dim, embed_dim, 2, 2))
            self.fpn3 = nn.Identity()
            self.fpn4 = nn.MaxPool2d(2, 2)
        elif with_fpn and patch_size == 8:
            self.fpn1 = nn.Sequential(nn.ConvTranspose2d(embed_dim, embed_dim, 2, 2))
            self.fpn2 = nn.Identity()
            self.fpn3 = nn.Sequential(nn.MaxPool2d(2, 2))
            self.fpn4 = nn.Sequential(nn.MaxPool2d(4, 4))
        self.point_token = nn.Parameter(torch.zeros(1, point_tokens_num, embed_dim))
        self.point_pos_embed = nn.Parameter(torch.zeros(1, point_tokens_num, embed_dim))
        if with_point_head:
            self.class_embed = MLP(embed_dim, embed_dim, num_classes, 3)
            self.bbox_embed = MLP(embed_dim, embed_dim, 2, 3)
        trunc_normal_(self.pos_embed)
        trunc_normal_(self.cls_token)
        trunc_normal_(self.point_token)
        trunc_normal_(self.point_pos_embed)
        self.apply(self._init_weights)
        self._wcache = {}

    # ---- reference API ---------------------------------------------------------------------------
    @staticmethod
    def _init_weights(m):
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def init_weights(self, pretrained=None):
        """visual_transformer_det.py:179-190: re-init, then (if a file) non-strict load."""
        if pretrained is not None and not isinstance(pretrained, str):
            raise TypeError("pretrained must be a str or None")
        self.apply(self._init_weights)
        if isinstance(pretrained, str):
            import os
            if os.path.isfile(pretrained):
                from .checkpoint import load_checkpoint
                load_checkpoint(self, pretrained, strict=False)
            else:
                print(f"checkpoint path {pretrained} is invalid, we skip it and initialize net randomly")
        self.invalidate_cache()

    def train(self, mode=True):
        # the reference's override forgets to return self (visual_transformer_det.py:153-156); nn.Module
        # semantics are kept here so `.eval()` chains.
        super().train(mode)
        self._freeze_stages()
        return self

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            self.patch_embed.eval()
            for p in self.patch_embed.parameters():
                p.requires_grad = False
            self.cls_token.requires_grad = False
            self.pos_embed.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            m = self.blocks[i - 1]
            m.eval()
            for p in m.parameters():
                p.requires_grad = False

    def invalidate_cache(self):
        self._wcache = {}

    # ---- helpers -----------------------------------------------------------------------------------
    def _w(self, p):
        """compute-dtype copy of a weight matrix, cached while parameters are frozen (eval / no-grad)."""
        if p.dtype == self.compute_dtype:
            return p
        if torch.is_grad_enabled() and p.requires_grad:
            return p.to(self.compute_dtype)
        key = id(p)
        hit = self._wcache.get(key)
        if hit is None or hit[0] != p._version or hit[1].device != p.device:
            hit = (p._version, p.detach().to(self.compute_dtype).contiguous())
            self._wcache[key] = hit
        return hit[1]

    def _derived(self, p, tag, fn):
        """cached compute-dtype matrix derived from parameter p (re-derived when p changes or needs grad)."""
        if torch.is_grad_enabled() and p.requires_grad:
            return fn(p).to(self.compute_dtype).contiguous()
        key = (id(p), tag)
        hit = self._wcache.get(key)
        if hit is None or hit[0] != p._version or hit[1].device != p.device:
            hit = (p._version, fn(p.detach()).to(self.compute_dtype).contiguous())
            self._wcache[key] = hit
        return hit[1]

    def _deconv2x2(self, x_nhwc, conv, bn=None, act="none"):
        """nn.ConvTranspose2d(k=2, s=2) on a channels-last map as ONE GEMM over its pixels:
        y[b, 2i+di, 2j+dj, co] = act(BN(sum_ci x[b,i,j,ci] W[ci,co,di,dj] + bias[co]))  (visual_transformer_det.py:107-117).
        An eval-mode BatchNorm `bn` is a per-channel affine map: it is folded into the cached GEMM weight / bias, the GELU
        runs in the GEMM epilogue, and the epilogue scatters to the interleaved output pixel (as_deconv2x2_fwd)."""
        B, h, w, cin = x_nhwc.shape
        cout = conv.weight.shape[1]
        srcs = [conv.weight, conv.bias] + ([bn.weight, bn.bias, bn.running_mean, bn.running_var] if bn is not None else [])
        stamp = tuple(-1 if t is None else t._version for t in srcs) + (x_nhwc.device,)
        key = (id(conv), "deconv")
        hit = self._wcache.get(key)
        if hit is None or hit[0] != stamp:
            with torch.no_grad():
                wmat = conv.weight.detach().float().permute(2, 3, 1, 0).reshape(4 * cout, cin)
                bias = torch.zeros(cout, device=wmat.device) if conv.bias is None else conv.bias.detach().float()
                if bn is not None:
                    scale = bn.weight.detach().float() * torch.rsqrt(bn.running_var.float() + bn.eps)
                    wmat = wmat * scale.repeat(4)[:, None]
                    bias = (bias - bn.running_mean.float()) * scale + bn.bias.detach().float()
                hit = (stamp, wmat.to(self.compute_dtype).contiguous(), bias.repeat(4).contiguous())
            self._wcache[key] = hit
        _, wmat, bias4 = hit
        if x_nhwc.is_cuda and self.compute_dtype == torch.bfloat16 and cin % 32 == 0 and cout % 8 == 0:
            return ops.deconv2x2(x_nhwc.contiguous(), wmat, bias4, act=act)
        y = ops.linear(x_nhwc.reshape(B * h * w, cin), wmat, bias4, act=act)
        return y.reshape(B, h, w, 2, 2, cout).permute(0, 1, 3, 2, 4, 5).reshape(B, 2 * h, 2 * w, cout)

    def _deconv2x2_train(self, x_nhwc, conv):
        """_deconv2x2 under autograd: the same token GEMM through autograd.LinearFn (HIP GEMM forward, split-K dW backward;
        the library GEMM when the sizes do not fit) instead of MIOpen's fp32 transposed convolution (whose backward
        dominated the training step: 17 ms of 55)."""
        B, h, w, cin = x_nhwc.shape
        cout = conv.weight.shape[1]
        cd = self.compute_dtype
        from .autograd import linear_in
        wmat = conv.weight.permute(2, 3, 1, 0).reshape(4 * cout, cin)
        bias = None if conv.bias is None else conv.bias.repeat(4)
        y = linear_in(x_nhwc.reshape(B * h * w, cin), wmat, bias, cd)         # HIP GEMM fwd / bwd (autograd.LinearFn)
        return y.reshape(B, h, w, 2, 2, cout).permute(0, 1, 3, 2, 4, 5).reshape(B, 2 * h, 2 * w, cout)

    def _fpn_train(self, i, feat_nchw, tok):
        B, D, hp, wp = feat_nchw.shape
        op = [self.fpn1, self.fpn2, self.fpn3, self.fpn4][i]
        if isinstance(op, nn.Sequential) and isinstance(op[0], nn.ConvTranspose2d):
            y = self._deconv2x2_train(tok.to(self.compute_dtype).reshape(B, hp, wp, D), op[0]).permute(0, 3, 1, 2)
            if len(op) == 4:
                y = op[2](op[1](y.float())).to(self.compute_dtype)
                y = self._deconv2x2_train(y.permute(0, 2, 3, 1), op[3]).permute(0, 3, 1, 2)
            return y
        return op(feat_nchw)

    def _fpn(self, i, feat_nchw, tok):
        """FPN tap i (visual_transformer_det.py:246-256).  feat_nchw: the fp32 [B,D,hp,wp] tap; tok: the same tokens
        as a token-major view [B,Np,D].  Deconvolved taps come back NCHW-shaped with channels-last strides in the
        compute dtype (no layout copy); identity / max-pool taps stay fp32 NCHW."""
        B, D, hp, wp = feat_nchw.shape
        ops_ = [self.fpn1, self.fpn2, self.fpn3, self.fpn4]
        op = ops_[i]
        if isinstance(op, nn.Sequential) and isinstance(op[0], nn.ConvTranspose2d):
            x0 = tok.to(self.compute_dtype).reshape(B, hp, wp, D)
            if len(op) == 4:                                             # ConvT -> BN -> GELU -> ConvT (fpn1, patch 16)
                bn = op[1]
                if isinstance(bn, nn.modules.batchnorm._BatchNorm) and not bn.training and bn.running_mean is not None \
                        and isinstance(op[2], nn.GELU) and getattr(op[2], "approximate", "none") == "none":
                    y = self._deconv2x2(x0, op[0], bn=bn, act="gelu")
                else:
                    y = self._deconv2x2(x0, op[0]).permute(0, 3, 1, 2)
                    y = op[2](op[1](y.float())).to(self.compute_dtype).permute(0, 2, 3, 1)
                return self._deconv2x2(y, op[3]).permute(0, 3, 1, 2)
            return self._deconv2x2(x0, op[0]).permute(0, 3, 1, 2)
        pool = op[0] if isinstance(op, nn.Sequential) and len(op) == 1 else op
        if isinstance(pool, nn.MaxPool2d) and tok.is_cuda and tok.dtype == torch.float32 and tok.stride()[1:] == (D, 1):
            k = pool.kernel_size if isinstance(pool.kernel_size, int) else pool.kernel_size[0]
            s_ = pool.stride if isinstance(pool.stride, int) else pool.stride[0]
            if k == s_ and pool.padding in (0, (0, 0)) and not pool.ceil_mode and hp % k == 0 and wp % k == 0 and D % 4 == 0:
                # the tap is token-major: pool it there (NCHW-shaped view of channels-last storage, like the other taps)
                return ops.maxpool_nhwc(tok.reshape(B, hp, wp, D), k).permute(0, 3, 1, 2)
        return op(feat_nchw)

    # ---- point head (visual_transformer_det.py:26-38, 152-153, 268-269) on the inference path -------------------------
    # class_embed and bbox_embed are two 3-layer FFNs over the SAME B*T = 200 point tokens: fifteen launches as written.
    # Their first layers are packed side by side (shared input) once per weight version and bias + ReLU ride in the GEMM
    # epilogue: five library GEMMs + the sigmoid.  (fp32, 200 rows: a plain small-M library GEMM -- the hand-written fp32
    # MFMA kernel's 128 x 128 tiles put this shape on 24 CUs, 170 us per layer; measured, round 5.)
    def _point_head_packable(self, x):
        ls = list(self.class_embed.layers) + list(self.bbox_embed.layers)
        return (len(self.class_embed.layers) == 3 and len(self.bbox_embed.layers) == 3 and hasattr(torch, "_addmm_activation")
                and all(l.weight.dtype == x.dtype and l.bias is not None for l in ls)
                and self.class_embed.layers[0].in_features == self.bbox_embed.layers[0].in_features)

    def _point_head_packed(self, x):
        c, b = self.class_embed.layers, self.bbox_embed.layers
        key = tuple((t.data_ptr(), t._version) for t in (c[0].weight, c[0].bias, b[0].weight, b[0].bias))
        pk = self._point_pack
        if pk is None or pk["key"] != key:
            with torch.no_grad():
                w1 = torch.cat((c[0].weight, b[0].weight)).contiguous()
                pk = dict(key=key, w1=w1, w1t=w1.t().contiguous(), b1=torch.cat((c[0].bias, b[0].bias)))
            self._point_pack = pk
        B, T, D = x.shape
        hc = c[0].out_features
        x2 = x.reshape(B * T, D)
        if ops.linear_small_applies(x2, pk["w1"]) and hc % 16 == 0 and hc >= 64:
            # round 6: five launches of the hand-written small-M fp32 kernel (as_linear_small_fwd) -- no vendor GEMM is left in
            # the forward; the two heads' second layers read their column slice of the packed first layer in place
            h1 = ops.linear_small(x2, pk["w1"], pk["b1"], act="relu")
            h2c = ops.linear_small(h1[:, :hc], c[1].weight, c[1].bias, act="relu")
            h2b = ops.linear_small(h1[:, hc:], b[1].weight, b[1].bias, act="relu")
            cls = ops.linear_small(h2c, c[2].weight, c[2].bias).reshape(B, T, -1)
            reg = ops.linear_small(h2b, b[2].weight, b[2].bias, act="sigmoid").reshape(B, T, -1)
            return cls, reg
        h1 = torch._addmm_activation(pk["b1"], x2, pk["w1t"])              # ReLU epilogue, both heads
        h2c = torch._addmm_activation(c[1].bias, h1[:, :hc], c[1].weight.t())
        h2b = torch._addmm_activation(b[1].bias, h1[:, hc:], b[1].weight.t())
        cls = torch.addmm(c[2].bias, h2c, c[2].weight.t()).reshape(B, T, -1)
        reg = torch.addmm(b[2].bias, h2b, b[2].weight.t()).reshape(B, T, -1).sigmoid()
        return cls, reg

    def interpolate_pos_encoding(self, n_patch_tokens, w, h):
        """models/vision_transformer.py:187-207 (bicubic, scale_factor with the +0.1 trick)."""
        n0 = self.pos_embed.shape[1] - 1
        if n_patch_tokens == n0 and w == h:
            return self.pos_embed
        dim = self.pos_embed.shape[-1]
        w0, h0 = w // self.patch_size + 0.1, h // self.patch_size + 0.1
        s = int(math.sqrt(n0))
        grid = self.pos_embed[:, 1:].reshape(1, s, s, dim).permute(0, 3, 1, 2)
        grid = F.interpolate(grid, scale_factor=(w0 / math.sqrt(n0), h0 / math.sqrt(n0)), mode="bicubic")
        assert int(w0) == grid.shape[-2] and int(h0) == grid.shape[-1]
        return torch.cat((self.pos_embed[:, :1], grid.permute(0, 2, 3, 1).reshape(1, -1, dim)), dim=1)

    def prepare_tokens(self, img):
        """visual_transformer_det.py:192-214.  The 16x16/16 conv is a [B*Np, 3*16*16] x [D, 768]^T GEMM."""
        B, C, w, h = img.shape
        ps = self.patch_size
        hp, wp = w // ps, h // ps
        # unfold + cast in one pass, then cls / position / point tokens are written straight into the token buffer
        patches = torch.empty(B, hp, wp, C, ps, ps, device=img.device, dtype=self.compute_dtype)
        patches.copy_(img.reshape(B, C, hp, ps, wp, ps).permute(0, 2, 4, 1, 3, 5))
        wmat = self._derived(self.patch_embed.proj.weight, "patch", lambda t: t.reshape(self.embed_dim, -1))
        emb = ops.linear(patches.view(B, hp * wp, C * ps * ps), wmat, self.patch_embed.proj.bias.float())
        # cls / position / point tokens: one constant table, added to the projection in one pass over the token tensor
        return ops.assemble_tokens(emb, self._token_table(hp * wp, w, h))

    def _token_table(self, Np, w, h):
        """[1 + Np + T, D] fp32: cls_token + pos_0, the (interpolated) position embedding of the patches, point_token +
        point_pos_embed -- what prepare_tokens adds around / to the patch projection; cached until a parameter changes."""
        srcs = (self.pos_embed, self.cls_token, self.point_token, self.point_pos_embed)
        stamp = tuple(t._version for t in srcs) + (Np, w, h, self.pos_embed.device)
        hit = self._wcache.get("token_table")
        if hit is None or hit[0] != stamp:
            with torch.no_grad():
                pos = self.interpolate_pos_encoding(Np, w, h)
                table = torch.cat((self.cls_token + pos[:, :1], pos[:, 1:], self.point_token + self.point_pos_embed), dim=1)
            hit = (stamp, table[0].float().contiguous())
            self._wcache["token_table"] = hit
        return hit[1]

    def _block(self, blk, x, delta, keep_state, need_x, on_full=None, full_out=None, x_out=None):
        """models/vision_transformer.py:109-124 with the residual stream kept in fp32 and every residual add fused into
        the LayerNorm that follows it (ops.add_layernorm).  `delta` is the previous block's MLP output that has not
        been added to `x` yet (None for the first block): the first fused add+LayerNorm completes the PREVIOUS block's
        output, which `on_full` (a feature tap) may look at; returns (x, pending MLP output, attention state); with
        `need_x` the block's own MLP output is added before returning (last block).  `full_out` / `x_out`: the caller's
        buffers for the completed previous output / this block's own full output (a tap's slot of the org_feats storage)."""
        cd = self.compute_dtype
        x, y = ops.add_layernorm(x, delta, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps, cd,
                                 x_out=full_out if delta is not None else None)
        if on_full is not None:
            on_full(x)
        a, st = ops.attention_fwd(y, self._w(blk.attn.qkv.weight),
                                  None if blk.attn.qkv.bias is None else blk.attn.qkv.bias.float(),
                                  self._w(blk.attn.proj.weight), blk.attn.proj.bias.float(), self.num_heads,
                                  keep_state=keep_state)
        x, z = ops.add_layernorm(x, a, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps, cd)
        z = ops.linear(z, self._w(blk.mlp.fc1.weight), blk.mlp.fc1.bias.float(), act="gelu")
        z = ops.linear(z, self._w(blk.mlp.fc2.weight), blk.mlp.fc2.bias.float())
        if need_x:
            x, _ = ops.add_layernorm(x, z, None, None, 0.0, cd, want_y=False, x_out=x_out)
            z = None
        return x, z, st

    # ---- trainable path (autograd): HIP attention forward + backward, library GEMMs for the MLP -------------------
    def _grad_path(self):
        return torch.is_grad_enabled() and self.training and any(p.requires_grad for p in self.blocks.parameters())

    def _drop_path(self, x, i):
        """Stochastic depth per sample, rate rising linearly with depth (models/vision_transformer.py:21-40, :160-164)."""
        rate = self.drop_path_rate * i / max(self.depth - 1, 1)
        if rate == 0.0 or not self.training:
            return x
        keep = 1.0 - rate
        mask = torch.rand(x.shape[0], 1, 1, device=x.device, dtype=x.dtype).add_(keep).floor_()
        return x / keep * mask

    def _drop_scale(self, batch, i, device):
        """The per-sample factor mask_b / keep of block i's DropPath as an fp32 vector [B] (None when inactive): the fused
        residual add applies it in fp32 (autograd.AddLayerNormFn delta_scale) -- no pass over the tokens, forward or
        backward.  Same draw as `_drop_path` (rand + keep, floor)."""
        rate = self.drop_path_rate * i / max(self.depth - 1, 1)
        if rate == 0.0 or not self.training:
            return None
        keep = 1.0 - rate
        return torch.rand(batch, device=device, dtype=torch.float32).add_(keep).floor_().div_(keep)

    def _prepare_tokens_train(self, img):
        B, C, w, h = img.shape
        ps = self.patch_size
        hp, wp = w // ps, h // ps
        cd = self.compute_dtype
        patches = img.reshape(B, C, hp, ps, wp, ps).permute(0, 2, 4, 1, 3, 5).reshape(B, hp * wp, C * ps * ps)
        from .autograd import linear_in
        x = linear_in(patches.to(cd), self.patch_embed.proj.weight.reshape(self.embed_dim, -1), self.patch_embed.proj.bias,
                      cd).float()
        x = torch.cat((self.cls_token.expand(B, -1, -1), x), dim=1)
        x = x + self.interpolate_pos_encoding(x.shape[1] - 1, w, h)
        pt = (self.point_token + self.point_pos_embed).expand(B, -1, -1)
        return torch.cat((x, pt), dim=1)

    def _block_train(self, blk, x, delta, i, sink, delta_scale=None):
        """Block.forward under autograd with the residual stream in fp32 and the residual adds fused into the LayerNorms in
        BOTH directions (autograd.AddLayerNormFn: as_add_layernorm / as_add_layernorm_bwd).  Attention =
        autograd.AttentionFn (as_attn_fwd / as_attn_bwd); the MLP = autograd.MlpFn (GELU and its derivative in the GEMM
        epilogues; library GEMMs + F.gelu only outside a bf16 region).  `delta` = the previous block's MLP output not yet added; returns (x, this block's pending MLP output)."""
        from . import autograd as AG
        cd = self.compute_dtype
        sh = self._train_shadow                          # compute-dtype copies of the blocks' GEMM parameters (forward())
        w = lambda p: sh[id(p)] if id(p) in sh else p.to(cd)                      # noqa: E731
        # `delta_scale` belongs to `delta` (the PREVIOUS block's MLP output and its DropPath factor); this block's two
        # DropPaths (vision_transformer.py:114-118) draw their own factors: independent draws, as in the reference
        x, y = AG.add_layernorm(x, delta, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps, cd, delta_scale)
        a = AG.attention(y, w(blk.attn.qkv.weight), None if blk.attn.qkv.bias is None else blk.attn.qkv.bias.float(),
                         w(blk.attn.proj.weight), blk.attn.proj.bias.float(), self.num_heads, sink)
        x, z = AG.add_layernorm(x, a, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps, cd, self._drop_scale(x.shape[0], i, x.device))
        if AG.mlp_applies(z, w(blk.mlp.fc1.weight), w(blk.mlp.fc2.weight)):
            # one autograd node, GELU / GELU' in the GEMM epilogues (autograd.MlpFn); fp32 master biases: db stays fp32
            z = AG.mlp(z, w(blk.mlp.fc1.weight), blk.mlp.fc1.bias, w(blk.mlp.fc2.weight), blk.mlp.fc2.bias)
        elif AG.linear_applies(z, w(blk.mlp.fc1.weight)):
            z = F.gelu(AG.linear(z, w(blk.mlp.fc1.weight), blk.mlp.fc1.bias))
            z = AG.linear(z, w(blk.mlp.fc2.weight), blk.mlp.fc2.bias)
        else:
            z = F.gelu(F.linear(z, w(blk.mlp.fc1.weight), w(blk.mlp.fc1.bias)))
            z = F.linear(z, w(blk.mlp.fc2.weight), w(blk.mlp.fc2.bias))
        return x, z, self._drop_scale(x.shape[0], i, x.device)

    def forward(self, x):
        """visual_transformer_det.py:221-275.  Under no-grad / eval every GEMM and the attention run on the HIP
        inference kernels; with grad enabled in train() mode the blocks run the autograd path above."""
        B, _, H, W = x.shape
        hp, wp = H // self.patch_size, W // self.patch_size
        T = self.point_tokens_num
        grad_path = self._grad_path()
        x = self._prepare_tokens_train(x) if grad_path else self.prepare_tokens(x)
        if self.recompute_last_feat:
            last_feat = x
        features, taps, attns = [], [], []
        store = []                                         # no-grad: token-major storage behind org_feats
        delta = None                                       # inference path: MLP output not yet added to x
        dscale = None                                      # training path: the DropPath factor [B] that goes with `delta`
        if not grad_path:
            x = x.contiguous()

        def take_tap(xf):
            """xf = a tapped block's full output [B, N, D]: feature map view + its slot of org_feats."""
            taps.append(xf[:, 1:-T])                                     # token-major view [B, Np, D]
            tap = xf[:, 1:, :][:, :-T].permute(0, 2, 1).unflatten(2, (hp, wp))
            if grad_path:
                # an NCHW-shaped VIEW of the token-major residual stream (no transposing clone: the consumers -- the FPN's token
                # GEMMs, RoIAlign -- read channels-last, and the gradient comes back in the same layout)
                features.append(tap if os.environ.get("AS_TAP_VIEW", "1") != "0" else tap.contiguous())
                return
            # no-grad: org_feats [B, L, D, hp, wp] is a VIEW of the token-major storage the tapped blocks' outputs were
            # WRITTEN into (tap_slot: the fused residual add's output buffer is the tap's slot of [L, B, N, D]), every tap
            # a channels-last map: no copy, no transpose, and the channels-last consumers (FPN GEMMs, RoIAlign, the
            # attention-shift kernels) read it in place
            if xf.data_ptr() != tap_slot(len(features)).data_ptr():
                tap_slot(len(features)).copy_(xf)              # (a block output that could not be directed into its slot)
            features.append(store[0][len(features)][:, 1:-T].unflatten(1, (hp, wp)).permute(0, 3, 1, 2))

        def tap_slot(k):
            if not store:
                store.append(torch.empty(len(self.out_indices), *x.shape, device=x.device, dtype=torch.float32))
            return store[0][k]

        tap_due = False                                    # the previous block is a tap whose MLP output is still pending
        nblk = len(self.blocks)
        self._train_shadow = {}
        if grad_path and self.compute_dtype != torch.float32 and not os.environ.get("AS_NO_PARAM_SHADOW"):
            # every GEMM parameter of the blocks in the compute dtype by ONE fused, differentiable cast (autograd.ParamCastFn)
            from .autograd import cast_params
            from . import autograd as _AG
            # the HIP linear takes the fp32 master biases; only the F.linear fallback consumes compute-dtype bias copies
            lib_mlp = _AG._LIBRARY_LINEAR or self.embed_dim % 32 != 0
            ps = []
            for blk in self.blocks:
                ps += [blk.attn.qkv.weight, blk.attn.proj.weight, blk.mlp.fc1.weight, blk.mlp.fc2.weight]
                if lib_mlp:
                    ps += [blk.mlp.fc1.bias, blk.mlp.fc2.bias]
            self._train_shadow = cast_params([p for p in ps if p.dtype != self.compute_dtype], self.compute_dtype)
        for i, blk in enumerate(self.blocks):
            if grad_path:
                sink = [] if self.return_attention else None
                x, delta, dscale = self._block_train(blk, x.float() if x.dtype != torch.float32 else x, delta, i, sink, dscale)
                if i in self.out_indices or i == nblk - 1:                  # taps / output need the block's full result
                    # (mixed-dtype operands are promoted inside the element-wise kernels: no separate fp32 copy of delta)
                    x = x + (delta if dscale is None else delta * dscale[:, None, None])
                    delta, dscale = None, None
                st = sink[0] if sink else None
                if i in self.out_indices:
                    take_tap(x)
            else:
                # a tapped block's full output appears inside the NEXT block's first fused add+LayerNorm: the tap is taken
                # there (no add-only pass); only the last block adds its own MLP output before returning
                last_tap = i in self.out_indices and i == nblk - 1
                x, delta, st = self._block(blk, x, delta, self.return_attention, i == nblk - 1,
                                           take_tap if tap_due else None,
                                           full_out=tap_slot(len(features)) if tap_due and x.dtype == torch.float32 else None,
                                           x_out=tap_slot(len(features) + int(tap_due)) if last_tap and x.dtype == torch.float32
                                           else None)
                tap_due = i in self.out_indices and i != nblk - 1
                if i in self.out_indices and i == nblk - 1:
                    take_tap(x)
            if self.return_attention:
                attns.append(st)
            if self.last_feat and not self.recompute_last_feat and i == nblk - 1:
                last_feat = x[:, :-T]
        self._train_shadow = {}                            # (the consumers' autograd nodes hold what backward needs)
        org_features = store[0][:, :, 1:-T].unflatten(2, (hp, wp)).permute(1, 0, 4, 2, 3) if store else None
        if org_features is None:
            org_features = torch.stack(features, dim=1)
        if self.with_fpn and grad_path:
            features = [self._fpn_train(i, features[i], taps[i]) for i in range(len(features))]
        elif self.with_fpn and self.defer_fpn and x.is_cuda:
            # nothing between here and the RoI head's feature extraction reads the FPN maps (two_stage_point_align.py:75-88:
            # seed_pseudo_gt works on attns / last_feat), so their GEMMs are handed to the caller as a DeferredFPN: the RoI
            # head launches them on a side stream when its own latency-bound chain of small launches starts
            features = DeferredFPN(self, features, taps)
        elif self.with_fpn:
            # the taps are token-major already: run the 2x2/2 deconvolutions as GEMMs over tokens (channels-last)
            features = [self._fpn(i, features[i], taps[i]) for i in range(len(features))]
        point_tokens = x[:, -T:]
        out = dict(org_feats=org_features, feature=features if isinstance(features, DeferredFPN) else tuple(features),
                   point_tokens=point_tokens)
        if self.with_point_head and not grad_path and point_tokens.is_cuda and self._point_head_packable(point_tokens):
            if self.point_head_stream:
                if self._point_stream is None:
                    self._point_stream = torch.cuda.Stream()
                main = torch.cuda.current_stream()
                self._point_stream.wait_stream(main)
                with torch.cuda.stream(self._point_stream):
                    cls, reg = self._point_head_packed(point_tokens)
                    out["point_head_ready"] = self._point_stream.record_event()
                for t in (point_tokens, cls, reg):               # (allocator: both streams use these blocks)
                    t.record_stream(main)
                    t.record_stream(self._point_stream)
            else:
                cls, reg = self._point_head_packed(point_tokens)
            out.update(outputs_class=cls, outputs_coord=reg)
        elif self.with_point_head:
            out.update(outputs_class=self.class_embed(point_tokens), outputs_coord=self.bbox_embed(point_tokens).sigmoid())
        if self.return_attention and self.last_feat:
            out.update(attns=attns)
        if self.last_feat:
            out.update(last_feat=last_feat)
        return out
