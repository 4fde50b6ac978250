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


class GradAllReducer:
    """Data-parallel gradient averaging for the trainable backbone (reference: mmdet's MMDistributedDataParallel around
    the detector, tools/train.py; NCCL all-reduce of every parameter gradient once per step).

    Gradients are packed into a few large flat buckets (default 64 MiB of bf16: xGMI is point-to-point, 7 links of
    ~153 GB/s per GPU, so a handful of big RCCL all-reduces beats hundreds of per-tensor ones) in REVERSE parameter order
    (the order backward produces them).  A bucket's all-reduce is launched asynchronously from the autograd hook of its
    last gradient, so communication of late layers overlaps the backward of early ones; `finish()` waits, averages and
    writes the result back into `p.grad`.  With one rank everything is a no-op."""

    def __init__(self, params, ranks, bucket_mb=64, comm_dtype=torch.bfloat16):
        self.ranks = ranks
        self.comm_dtype = comm_dtype
        self.params = [p for p in params if p.requires_grad]
        self.buckets = []
        self._where = {}
        if ranks.world == 1:
            return
        limit = int(bucket_mb * (1 << 20)) // torch.empty((), dtype=comm_dtype).element_size()
        cur, cur_n = [], 0
        for p in reversed(self.params):
            if cur and cur_n + p.numel() > limit:
                self._seal(cur, cur_n)
                cur, cur_n = [], 0
            cur.append((p, cur_n))
            cur_n += p.numel()
        if cur:
            self._seal(cur, cur_n)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    def _seal(self, items, numel):
        dev = items[0][0].device
        b = dict(items=items, flat=torch.zeros(numel, dtype=self.comm_dtype, device=dev), ready=0, work=None)
        for p, off in items:
            self._where[id(p)] = (b, off)
        self.buckets.append(b)

    def _on_grad(self, p):
        b, off = self._where[id(p)]
        b["flat"][off:off + p.numel()].copy_(p.grad.reshape(-1))
        b["ready"] += 1
        if b["ready"] == len(b["items"]):
            b["work"] = self.ranks.dist.all_reduce(b["flat"], op=self.ranks.dist.ReduceOp.SUM, async_op=True)

    def finish(self):
        """Call after loss.backward(): completes every bucket and leaves the rank-averaged gradient in p.grad."""
        if self.ranks.world == 1:
            return
        inv = 1.0 / self.ranks.world
        for b in self.buckets:
            if b["work"] is None:              # some parameter got no gradient this step: contribute zeros for it
                for p, off in b["items"]:
                    if p.grad is None:
                        b["flat"][off:off + p.numel()].zero_()
                b["work"] = self.ranks.dist.all_reduce(b["flat"], op=self.ranks.dist.ReduceOp.SUM, async_op=True)
            b["work"].wait()
            for p, off in b["items"]:
                avg = b["flat"][off:off + p.numel()].reshape(p.shape)
                if p.grad is None:
                    p.grad = (avg * inv).to(p.dtype)
                else:
                    p.grad.copy_(avg).mul_(inv)
            b["ready"], b["work"] = 0, None

    def close(self):
        for h in getattr(self, "_hooks", []):
            h.remove()
