"""mmdet.ops.chamfer_2d on the HIP kernels (the reference's second native op).  Same module / function
names and return values as mmdet/ops/chamfer_2d/dist_chamfer_2d.py:11-58."""
import torch
from torch import nn
from torch.autograd import Function

from . import ops


class ChamferFunction2D(Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        dist1, dist2, idx1, idx2 = ops.chamfer_2d_fwd(xyz1, xyz2)
        ctx.save_for_backward(xyz1, xyz2, idx1, idx2)
        ctx.mark_non_differentiable(idx1, idx2)
        return dist1, dist2, idx1, idx2

    @staticmethod
    def backward(ctx, graddist1, graddist2, gradidx1, gradidx2):
        xyz1, xyz2, idx1, idx2 = ctx.saved_tensors
        return ops.chamfer_2d_bwd(xyz1, xyz2, graddist1, graddist2, idx1, idx2)


class Chamfer2D(nn.Module):
    def forward(self, input1, input2):
        return ChamferFunction2D.apply(input1.contiguous(), input2.contiguous())
