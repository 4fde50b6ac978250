"""A3 on the kernel that actually runs on the BASELINE ViT configurations.

`as_rollout_step` takes `rollout_step4_kernel` for bf16 tensors with h % 4 == 0 and a workspace (csrc/rollout.hip) --
every ViT-B (h = 12) and ViT-L (h = 16) step.  These tests hold THAT kernel, in both its forms (all T point-token rows;
the <= 32-row form behind `rows=`), to the oracle's `attns_project_to_feature` restatement (stdroi:1257-1272 + the row
slice of :2272): the oracle builds the dense head-mean attention of every layer from the same bf16-rounded operands and
multiplies the augmented matrices; the device recomputes attention tiles from (q, k, lse).  bf16 tolerance as the
existing 3-head test (3e-2 of the range; observed ~1e-2: q/k are bf16, the roll-out operand R is carried in bf16).
"""
import pytest
import torch

import attnshift_oracle as O

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.fixture(scope="module")
def ops():
    from attentionshift_amd import ops as _ops
    _ops._lib.load()
    return _ops


def dev(x):
    return x.cuda().contiguous()


def rel_to_range(ref, got):
    ref, got = ref.double().cpu(), got.double().cpu()
    scale = ref.abs().max().item() + 1e-30
    err = (ref - got).abs()
    return err.max().item() / scale, err.mean().item() / scale


def _layers(ops, B, N, h, L, seed, scale=3.0):
    """L attention layers on seeded inputs: device states (q, k, lse) and the oracle's head-mean matrices [B,N,N] of the
    same bf16-rounded operands."""
    D = 64 * h
    states, means = [], []
    for l in range(L):
        g = torch.Generator().manual_seed(seed + l)
        x = torch.randn(B, N, D, generator=g).bfloat16()
        wqkv = (torch.randn(3 * D, D, generator=g) * (scale / D ** 0.5)).bfloat16()
        bqkv = torch.randn(3 * D, generator=g) * 0.1
        wproj = (torch.randn(D, D, generator=g) / D ** 0.5).bfloat16()
        bproj = torch.zeros(D)
        means.append(O.attention_head_mean(x.float(), wqkv.float(), bqkv, h))
        _, st = ops.attention_fwd(dev(x), dev(wqkv), dev(bqkv), dev(wproj), dev(bproj), h)
        states.append(st)
    return states, means


def _uses_step4(ops, st, T):
    """the dispatch condition of as_rollout_step (csrc/rollout.hip): bf16, h % 4 == 0, workspace supplied"""
    return st.q.dtype == torch.bfloat16 and st.h % 4 == 0 and ops._lib.load().as_rollout_step_workspace_bytes(st.B, st.N, T) > 0


@pytest.mark.parametrize("B,N,h,T,L", [(2, 457, 4, 40, 4), (1, 1090, 12, 100, 3), (1, 300, 16, 100, 3)])
def test_rollout_step4_all_rows_match_oracle(ops, B, N, h, T, L):
    """full-T form (rollout_step4_kernel<4>): every point-token row of every partial product."""
    states, means = _layers(ops, B, N, h, L, 700 + N)
    assert _uses_step4(ops, states[0], T)
    ref = O.rollout_rows(means, T)
    got = ops.rollout_rows(states, T)
    assert got.shape == ref.shape
    mx, mean = rel_to_range(ref, got)
    assert mx < 3e-2 and mean < 2e-3, (mx, mean)
    # the rows are probability distributions over the tokens (row-stochastic factors): a dropped partial product or a
    # mis-weighted head group shows up here independently of the element-wise tolerance
    s = got.sum(-1)
    assert (s - 1).abs().max().item() < 2e-2, (s.min().item(), s.max().item())


@pytest.mark.parametrize("B,N,h,T,L", [(2, 457, 4, 40, 4), (1, 1090, 12, 100, 3)])
def test_rollout_step4_row_subset_matches_oracle(ops, B, N, h, T, L):
    """<= 32-row form (rollout_step4_kernel<1> behind `rows=`, the form the RoI head uses for the matched tokens):
    against the SAME rows of the oracle's roll-out, not against the device's own full roll-out."""
    states, means = _layers(ops, B, N, h, L, 900 + N)
    ref = O.rollout_rows(means, T)                                        # [B, L, T, N]
    sel = torch.stack([torch.randperm(T, generator=torch.Generator().manual_seed(b))[:7] for b in range(B)])
    got = ops.rollout_rows(states, T, rows=dev(sel))
    want = torch.gather(ref, 2, sel[:, None, :, None].expand(-1, L, -1, N))
    assert got.shape == want.shape
    mx, mean = rel_to_range(want, got)
    assert mx < 3e-2 and mean < 2e-3, (mx, mean)


def test_rollout_step4_config2_size_matched_rows_match_oracle(ops):
    """BASELINE config-2 token count (N = 4197, 12 heads), 3 layers, the first three point tokens matched (the rows the
    RoI head reads at G = 3): the device's recomputed-tile roll-out against the oracle's DENSE head-mean product
    (12 x 4197^2 softmax per layer on the host).  This is the launch geometry of the bench step: 33 column blocks x 5 head
    groups x 3 contraction splits, ragged last block (4197 = 131 * 32 + 5)."""
    B, N, h, T, L = 1, 4197, 12, 100, 3
    states, means = _layers(ops, B, N, h, L, 4197)
    sel = torch.tensor([[0, 1, 2]])
    got = ops.rollout_rows(states, T, rows=dev(sel)).cpu()                # [1, L, 3, N]
    # oracle, row-sliced by hand so the [N,N] products are [3,N] x [N,N] (stdroi:1257-1272 on rows N-T+{0,1,2})
    eye = torch.eye(N)
    run, outs = None, []
    for l in range(L - 1, -1, -1):
        aug = means[l][0] + eye
        aug = aug / aug.sum(-1, keepdim=True)
        run = aug[N - T + sel[0]] if run is None else run @ aug
        outs.append(run)
    want = torch.stack(outs)[None]
    mx, mean = rel_to_range(want, got)
    assert mx < 3e-2 and mean < 2e-3, (mx, mean)
    # what the consumer reads: columns [1:-T] as 64x64 CAMs; compare after its own min-max normalisation too
    cam_w = O.minmax_maps(want[0, :, :, 1:-T].reshape(-1, 64, 64))
    cam_g = O.minmax_maps(got[0, :, :, 1:-T].reshape(-1, 64, 64))
    assert (cam_w - cam_g).abs().max().item() < 5e-2
