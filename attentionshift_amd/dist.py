"""One-process-per-GPU plumbing for the data-parallel path (images shard over ranks; the forward / attention-shift
path has no data-path collective).  RCCL (`backend="nccl"` on ROCm) on GPU boxes, gloo in the CPU tests."""
import os

import torch


class Ranks:
    def __init__(self, backend=None, device=None):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = device
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
            kw = {}
            if backend == "nccl" and device is not None:
                kw["device_id"] = device
            dist.init_process_group(backend=backend, **kw)
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, value):
        """max of a python float over all ranks (the slowest rank defines the step time)."""
        if self.dist is None:
            return float(value)
        t = torch.tensor([value], dtype=torch.float64, device=self.device if self.device is not None else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value):
        if self.dist is None:
            return float(value)
        t = torch.tensor([value], dtype=torch.float64, device=self.device if self.device is not None else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def shard(self, n_items):
        """contiguous shard [lo, hi) of n_items for this rank (image sharding of a global batch)."""
        per, rem = divmod(n_items, self.world)
        lo = self.rank * per + min(self.rank, rem)
        return lo, lo + per + (1 if self.rank < rem else 0)

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
