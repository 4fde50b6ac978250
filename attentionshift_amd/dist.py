"""One-process-per-GPU plumbing for the data-parallel path (images shard over ranks; the forward / attention-shift
path has no data-path collective).  RCCL (`backend="nccl"` on ROCm) on GPU boxes, gloo in the CPU tests."""
import os

import torch


class Ranks:
    """WORLD_SIZE / RANK / LOCAL_RANK from the launcher.  On GPU boxes the process group carries TWO backends
    ("cpu:gloo,cuda:nccl"): the timing barriers and the max / sum of host scalars go through gloo on CPU tensors, so the
    forward / attention-shift throughput (which has no data-path collective) never depends on RCCL; RCCL communicators are
    created lazily by the first collective on a device tensor -- the gradient all-reduce of the training step.

    A process group that already exists (the reference's `mmcv.runner.init_dist` creates an nccl-only one before the model
    is built, tools/train.py) is REUSED, not re-created; host scalars then travel on whatever the group offers: CPU
    tensors when it has a CPU backend, tensors on `device` otherwise (so `Ranks(backend="nccl", device=...)` works too).

    `force=True` (or AS_FORCE_DIST=1) creates a real ONE-rank group when the launcher gave none: `Ranks.dist` is live, the
    reducer builds its buckets and every collective really runs (RCCL on a single GPU) -- the N > 1 code path made
    executable on a one-GPU box."""

    def __init__(self, backend=None, device=None, force=None):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = device
        self.dist = None
        self._own_group = False
        self._tmp = None
        if force is None:
            force = os.environ.get("AS_FORCE_DIST", "0") == "1"
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            self.dist = dist
            self.world, self.rank = dist.get_world_size(), dist.get_rank()
        elif self.world > 1 or force:
            backend = backend or ("cpu:gloo,cuda:nccl" if torch.cuda.is_available() else "gloo")
            if self.world > 1:
                dist.init_process_group(backend=backend)
            else:                      # forced one-rank group: no launcher, no port -- a file store in a private directory
                import tempfile
                self._tmp = tempfile.mkdtemp(prefix="as_dist_")
                dist.init_process_group(backend=backend, init_method="file://" + os.path.join(self._tmp, "store"),
                                        rank=0, world_size=1)
            self.dist = dist
            self._own_group = True
        self._host_dev = torch.device("cpu")
        if self.dist is not None and not self._has_cpu_backend():
            if device is None:
                raise RuntimeError("Ranks: the process group has no CPU backend (e.g. backend='nccl'); pass device= so that "
                                   "barriers and host-scalar reductions can run on device tensors")
            self._host_dev = torch.device(device)

    def _has_cpu_backend(self):
        try:
            cfg = str(self.dist.get_backend_config())
            return "cpu:" in cfg
        except Exception:                     # noqa: BLE001 -- older torch: fall back to the backend's name
            return "gloo" in str(self.dist.get_backend())

    @property
    def active(self):
        """True when collectives are live (N > 1, or a forced one-rank group)."""
        return self.dist is not None

    def _host_reduce(self, value, op):
        t = torch.tensor([value], dtype=torch.float64, device=self._host_dev)   # CPU tensor: gloo (when the group has it)
        self.dist.all_reduce(t, op=op)
        return float(t.item())

    def barrier(self):
        if self.dist is not None:
            self._host_reduce(0.0, self.dist.ReduceOp.SUM)                # an all-reduce every rank must enter

    def max_over_ranks(self, value):
        """max of a python float over all ranks (the slowest rank defines the step time)."""
        if self.dist is None:
            return float(value)
        return self._host_reduce(value, self.dist.ReduceOp.MAX)

    def sum_over_ranks(self, value):
        if self.dist is None:
            return float(value)
        return self._host_reduce(value, self.dist.ReduceOp.SUM)

    def shard(self, n_items):
        """contiguous shard [lo, hi) of n_items for this rank (image sharding of a global batch)."""
        per, rem = divmod(n_items, self.world)
        lo = self.rank * per + min(self.rank, rem)
        return lo, lo + per + (1 if self.rank < rem else 0)

    def close(self):
        if self.dist is not None and self._own_group:     # a group the caller (mmcv's init_dist) made is the caller's
            self.dist.destroy_process_group()
        self.dist = None
        if self._tmp is not None:
            import shutil
            shutil.rmtree(self._tmp, ignore_errors=True)
            self._tmp = None


def broadcast_state(tensors, ranks, src=0, chunk_mb=256):
    """Rank `src`'s values into every rank's `tensors` (parameters AND buffers), in place: what
    MMDistributedDataParallel does at construction (mmdet/apis/train.py:96-100 -> torch DDP's `_sync_params_and_buffers`),
    so that ranks which initialised unseeded, or of which only rank 0 loaded the checkpoint, start from identical
    weights.  Tensors are packed by dtype into flat chunks (a few large broadcasts instead of one per tensor: the ViT-B
    detector has ~420 tensors) and unpacked on arrival.  Returns the number of collectives issued."""
    if ranks is None or ranks.dist is None:
        return 0
    groups = {}
    for t in tensors:
        groups.setdefault((t.dtype, t.device), []).append(t)
    n_coll = 0
    with torch.no_grad():
        for (dtype, dev), ts in groups.items():
            limit = max(1, int(chunk_mb * (1 << 20)) // max(torch.empty((), dtype=dtype).element_size(), 1))
            i = 0
            while i < len(ts):
                j, n = i, 0
                while j < len(ts) and (j == i or n + ts[j].numel() <= limit):
                    n += ts[j].numel()
                    j += 1
                wire = torch.uint8 if dtype == torch.bool else dtype
                flat = torch.cat([t.detach().reshape(-1).to(wire) for t in ts[i:j]]) if n else torch.empty(0, dtype=wire, device=dev)
                ranks.dist.broadcast(flat, src=src)
                n_coll += 1
                off = 0
                for t in ts[i:j]:
                    t.detach().copy_(flat[off:off + t.numel()].reshape(t.shape).to(dtype))
                    off += t.numel()
                i = j
    return n_coll


class GradAllReducer:
    """Data-parallel gradient averaging for the trainable backbone (reference: mmdet's MMDistributedDataParallel around
    the detector, mmdet/apis/train.py:95-100; NCCL all-reduce of every parameter gradient once per optimizer step).

    Construction broadcasts rank 0's parameters (and `buffers`, when given) to every rank, as the reference's DDP wrapper
    does -- the ranks need not share a seed or a checkpoint load.

    Gradients are packed into a few large flat buckets (default 64 MiB: xGMI is point-to-point, 7 links of
    ~153 GB/s per GPU, so a handful of big RCCL all-reduces beats hundreds of per-tensor ones) in REVERSE parameter order
    (the order backward produces them).  A bucket's all-reduce is launched asynchronously from the autograd hook of its
    last gradient, so communication of late layers overlaps the backward of early ones -- but always in BUCKET ORDER on
    every rank: bucket k is only launched once buckets < k are, and whatever is left (buckets holding a parameter that
    got no gradient on this rank, e.g. the mask head on a rank without positives) is launched in order by `finish()`,
    which waits and writes the rank-averaged result back into `p.grad`.  Ranks may therefore differ in which parameters
    receive gradients without mis-pairing collectives.  `no_sync()` skips the exchange for the micro-steps of a gradient
    accumulation (the reference's update_interval=2, mmdet/utils/optimizer.py:23-32): gradients accumulate in p.grad and
    the step that leaves the context reduces the accumulated values.

    Precision of the exchange: the default wire type is fp32 -- the sum over ranks is then the reference's (torch DDP
    all-reduces fp32 gradients), and at 8 ranks the 0.36 GB of the detector's gradients are ~2.6 ms of ring time hidden
    under a ~25 ms backward.  `comm_dtype=torch.bfloat16` halves the bytes; gradients are then
    PRE-SCALED by 1/world while they are packed, so the running sum stays in the range of one rank's gradient instead of
    growing by log2(world) binades before the scale (each partial sum still rounds to bf16's 8 bits -- use it only when
    the exchange is exposed).

    The reducer is live whenever the process group is (`ranks.dist is not None`): N > 1, or a forced one-rank group
    (`Ranks(force=True)` / AS_FORCE_DIST=1), which runs the identical bucket / hook / RCCL / write-back path on one GPU.
    Without a group everything is a no-op."""

    def __init__(self, params, ranks, bucket_mb=64, comm_dtype=torch.float32, buffers=None, broadcast=True,
                 find_unused_parameters=False):
        self.ranks = ranks
        self.comm_dtype = comm_dtype
        # A parameter that received no gradient on ANY rank keeps p.grad = None after finish() (torch DDP leaves such
        # gradients undefined, so weight decay / momentum skip the parameter) -- known locally in a one-rank group; with
        # more ranks it takes one extra small all-reduce per step, so it is opt-in there (off: zeros, as before)
        self.find_unused_parameters = find_unused_parameters
        self.params = [p for p in params if p.requires_grad]
        self.buckets = []
        self._where = {}
        self._sync = True
        self._next = 0                       # first bucket whose all-reduce has not been launched yet
        self.active = ranks is not None and ranks.dist is not None
        self.broadcasts = 0
        if not self.active:
            return
        self._inv = 1.0 / ranks.world
        # fp32 wire: sum, then scale on write-back (exact w.r.t. the reference's order of operations);
        # 16-bit wire: scale while packing so the running sum cannot leave the addends' range
        self._prescale = comm_dtype != torch.float32
        if broadcast:
            self.broadcasts = broadcast_state(list(self.params) + list(buffers or []), ranks)
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
        flat = torch.zeros(numel, dtype=self.comm_dtype, device=dev)
        # the parameters' windows into the bucket, made ONCE (300 slice + view calls per step were ~2.5 ms of host time)
        views = [flat[off:off + p.numel()].view(p.shape) for p, off in items]
        b = dict(items=items, flat=flat, views=views, seen=set(), work=None, index=len(self.buckets))
        for p, off in items:
            self._where[id(p)] = (b, off)
        self.buckets.append(b)

    def no_sync(self):
        """Context manager: backward passes inside it only accumulate into p.grad (no copy, no collective)."""
        reducer = self

        class _NoSync:
            def __enter__(self):
                reducer._sync = False

            def __exit__(self, *exc):
                reducer._sync = True
                return False

        return _NoSync()

    def _pack_many(self, b, items):
        """p.grad -> the bucket slices of `items`, ONE multi-tensor launch (a bucket holds ~40 tensors: per-tensor copies
        from the hooks were ~3 ms of host time per step)."""
        if not items:
            return
        if items is b["items"]:
            dst = b["views"]
        else:
            where = {id(p): v for (p, _), v in zip(b["items"], b["views"])}
            dst = [where[id(p)] for p, _ in items]
        src = [p.grad for p, _ in items]
        keep = [i for i, (d, g) in enumerate(zip(dst, src)) if d.data_ptr() != g.data_ptr()]   # already a bucket view
        if len(keep) < len(dst):
            dst, src = [dst[i] for i in keep], [src[i] for i in keep]
            if not dst:
                return
        torch._foreach_copy_(dst, src)
        if self._prescale:
            torch._foreach_mul_(dst, self._inv)

    def _launch_ready(self):
        while self._next < len(self.buckets):
            b = self.buckets[self._next]
            if len(b["seen"]) < len(b["items"]):
                return
            self._pack_many(b, b["items"])
            b["work"] = self.ranks.dist.all_reduce(b["flat"], op=self.ranks.dist.ReduceOp.SUM, async_op=True)
            self._next += 1

    def _on_grad(self, p):
        if not self._sync:
            return
        b, off = self._where[id(p)]
        if b["work"] is not None:            # a second backward before finish(): this bucket is already in flight
            raise RuntimeError("GradAllReducer: backward ran again before finish(); use no_sync() for accumulation steps")
        b["seen"].add(id(p))                 # (the copy happens once per BUCKET, when its last gradient has arrived)
        if b["index"] == self._next:
            self._launch_ready()

    def finish(self):
        """Call after the (last) loss.backward() of a step: completes every bucket in order and leaves the rank-averaged
        gradient in p.grad.  fp32 wire: p.grad becomes a VIEW of the reduced bucket (no copy back; the bucket is scaled
        by 1/world in one pass) -- `zero_grad(set_to_none=True)` drops the views, an in-place zero_grad clears the
        bucket, both are fine.  16-bit wire: one multi-tensor conversion per bucket."""
        if not self.active:
            return
        unused = set()
        missing = [i for i, p in enumerate(self.params) if p.grad is None]
        if self.ranks.world == 1:
            unused = {id(self.params[i]) for i in missing}
        elif missing and not self.find_unused_parameters and not getattr(self, "_warned_unused", False):
            # one GPU leaves such gradients None (AdamW skips the parameter); N > 1 without find_unused_parameters writes the
            # rank-averaged zeros (weight decay and momentum keep running): say so once instead of training differently in silence
            import warnings
            self._warned_unused = True
            warnings.warn(f"GradAllReducer: {len(missing)} parameter(s) received no gradient on rank {self.ranks.rank}; with "
                          f"world={self.ranks.world} and find_unused_parameters=False they get zero gradients (a one-rank run "
                          "leaves them None).  Pass find_unused_parameters=True to keep None on every world size.", RuntimeWarning)
        elif self.find_unused_parameters:
            used = torch.ones(len(self.params), dtype=torch.float32)
            used[missing] = 0.0
            used = used.to(self.params[0].device)
            self.ranks.dist.all_reduce(used, op=self.ranks.dist.ReduceOp.MAX)
            unused = {id(p) for p, u in zip(self.params, used.tolist()) if u == 0.0}
        for b in self.buckets[self._next:]:   # buckets with a parameter that got no gradient: zeros for it, in order
            have = [(p, off) for p, off in b["items"] if p.grad is not None]   # incl. grads accumulated under no_sync()
            for (p, _), v in zip(b["items"], b["views"]):
                if p.grad is None:
                    v.zero_()
            self._pack_many(b, have)
            b["work"] = self.ranks.dist.all_reduce(b["flat"], op=self.ranks.dist.ReduceOp.SUM, async_op=True)
        for b in self.buckets:
            b["work"].wait()
            views = b["views"]
            if not self._prescale and all(p.dtype == self.comm_dtype for p, _ in b["items"]):
                if self._inv != 1.0:
                    b["flat"].mul_(self._inv)
                for (p, _), v in zip(b["items"], views):
                    p.grad = None if id(p) in unused else v
            else:
                post = 1.0 if self._prescale else self._inv
                live = [(p, v) for (p, _), v in zip(b["items"], views) if id(p) not in unused]
                for p, _ in live:
                    if p.grad is None:
                        p.grad = torch.empty_like(p)
                if live:
                    torch._foreach_copy_([p.grad for p, _ in live], [v for _, v in live])
                    if post != 1.0:
                        torch._foreach_mul_([p.grad for p, _ in live], post)
            b["seen"], b["work"] = set(), None
        self._next = 0

    def close(self):
        for h in getattr(self, "_hooks", []):
            h.remove()


class PendingLogVars:
    """The logged scalars of parse_losses(lazy=True): the stacked device tensor is on its way to pinned host memory;
    `resolve()` waits for that copy (the step's only loss-related sync) and returns the OrderedDict of python floats."""

    def __init__(self, keys, flat):
        self.keys = keys
        if flat.is_cuda:
            self.host = torch.empty(flat.shape, dtype=flat.dtype, pin_memory=True)
            self.host.copy_(flat, non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record()
        else:
            self.host, self.event = flat, None

    def resolve(self):
        from collections import OrderedDict
        if self.event is not None:
            self.event.synchronize()
        return OrderedDict(zip(self.keys, self.host.tolist()))


def parse_losses(losses, ranks=None, lazy=False):
    """mmdet/models/detectors/base.py:185-218 `_parse_losses`: per-key means, `loss` = sum of the keys containing 'loss',
    and the logged values averaged over ranks -- with ONE all-reduce of the stacked scalars instead of one blocking
    all-reduce per key (the reference issues len(losses)+1 of them per step).  Returns (loss tensor, log_vars dict of
    python floats).  lazy=True: the second element is a PendingLogVars -- the reference reads the scalars back (`.item()`)
    BEFORE backward, which makes the host wait for the whole forward and only then start queueing the backward; resolving
    them after the optimizer step has been queued keeps the host ahead of the device through the backward."""
    from collections import OrderedDict
    log_vars = OrderedDict()
    for name, value in losses.items():
        if isinstance(value, torch.Tensor):
            log_vars[name] = value.mean()
        elif isinstance(value, list):
            log_vars[name] = sum(v.mean() for v in value)
        else:
            raise TypeError(f"{name} is not a tensor or list of tensors")
    loss = sum(v for k, v in log_vars.items() if "loss" in k)
    log_vars["loss"] = loss
    keys = list(log_vars.keys())
    flat = torch.stack([log_vars[k].detach().float().reshape(()) for k in keys])
    if ranks is not None and ranks.dist is not None:
        flat = flat / ranks.world
        ranks.dist.all_reduce(flat)
    if lazy:
        return loss, PendingLogVars(keys, flat)
    vals = flat.tolist()
    return loss, OrderedDict((k, v) for k, v in zip(keys, vals))


class _SyncBNFn(torch.autograd.Function):
    """Batch norm over the GLOBAL batch: one all-reduce of [sum, sum of squares, count] (2C+1 floats) forward, one of
    [sum dy, sum dy*xhat] backward -- the exchange torch.nn.SyncBatchNorm does, written on plain all_reduce so that it
    also runs on gloo (CPU tests) and needs no GPU-only kernel."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, ranks):
        C = x.shape[1]
        dims = [0] + list(range(2, x.dim()))
        xf = x.float()
        # (the element count as a device FILL, not `new_tensor([...])`: a pageable host -> device copy waits for the whole
        # stream and left the host in lock-step with the device for the rest of the step -- found with the one-rank group)
        stat = torch.cat((xf.sum(dims), (xf * xf).sum(dims), torch.full((1,), xf.numel() / C, device=xf.device, dtype=xf.dtype)))
        ranks.dist.all_reduce(stat, op=ranks.dist.ReduceOp.SUM)
        n = stat[-1]
        mean = stat[:C] / n
        var = (stat[C:2 * C] / n - mean * mean).clamp_min(0)                    # biased, as batch norm normalises with
        invstd = torch.rsqrt(var + eps)
        shape = [1, C] + [1] * (x.dim() - 2)
        xhat = (xf - mean.view(shape)) * invstd.view(shape)
        ctx.save_for_backward(xhat, weight, invstd)
        ctx.ranks, ctx.n, ctx.dims, ctx.shape = ranks, n, dims, shape
        ctx.mark_non_differentiable(mean, var, n)
        y = xhat * weight.float().view(shape) + bias.float().view(shape)
        return y.to(x.dtype), mean, var, n

    @staticmethod
    def backward(ctx, dy, _dm, _dv, _dn):
        xhat, weight, invstd = ctx.saved_tensors
        dyf = dy.float()
        C = xhat.shape[1]
        s_dy, s_dyx = dyf.sum(ctx.dims), (dyf * xhat).sum(ctx.dims)
        both = torch.cat((s_dy, s_dyx))
        ctx.ranks.dist.all_reduce(both, op=ctx.ranks.dist.ReduceOp.SUM)
        m_dy, m_dyx = (both[:C] / ctx.n).view(ctx.shape), (both[C:] / ctx.n).view(ctx.shape)
        dx = (weight.float() * invstd).view(ctx.shape) * (dyf - m_dy - xhat * m_dyx)
        # weight / bias gradients stay LOCAL sums: the data-parallel gradient average reduces them like every parameter
        return dx.to(dy.dtype), s_dyx.to(weight.dtype), s_dy.to(weight.dtype), None, None


class SyncBatchNorm2d(torch.nn.BatchNorm2d):
    """nn.BatchNorm2d whose training statistics span all ranks (the reference converts every BatchNorm of the detector
    with torch.nn.SyncBatchNorm.convert_sync_batchnorm before wrapping it in DDP, mmdet/apis/train.py:95; on this path
    that is the one BatchNorm of the FPN's stride-4 branch, visual_transformer_det.py:109).  Same parameters, buffers and
    state-dict keys as nn.BatchNorm2d; with one rank, or in eval mode, it IS nn.BatchNorm2d."""

    def __init__(self, num_features, ranks=None, **kw):
        super().__init__(num_features, **kw)
        self.ranks = ranks

    def forward(self, x):
        r = self.ranks
        if not self.training or r is None or r.dist is None:
            return super().forward(x)
        y, mean, var, n = _SyncBNFn.apply(x, self.weight, self.bias, self.eps, r)
        if self.track_running_stats:
            with torch.no_grad():
                self.num_batches_tracked += 1
                mom = self.momentum if self.momentum is not None else 1.0 / float(self.num_batches_tracked)
                self.running_mean.mul_(1 - mom).add_(mean.to(self.running_mean.dtype), alpha=mom)
                self.running_var.mul_(1 - mom).add_((var * (n / (n - 1).clamp_min(1))).to(self.running_var.dtype), alpha=mom)
        return y


def convert_sync_batchnorm(module, ranks):
    """Replace every nn.BatchNorm2d under `module` by SyncBatchNorm2d sharing its parameters and buffers."""
    for name, child in list(module.named_children()):
        if isinstance(child, torch.nn.BatchNorm2d) and not isinstance(child, SyncBatchNorm2d):
            new = SyncBatchNorm2d(child.num_features, ranks, eps=child.eps, momentum=child.momentum, affine=child.affine,
                                  track_running_stats=child.track_running_stats)
            if child.affine:
                new.weight, new.bias = child.weight, child.bias
            if child.track_running_stats:
                new.running_mean, new.running_var, new.num_batches_tracked = \
                    child.running_mean, child.running_var, child.num_batches_tracked
            new.train(child.training)
            setattr(module, name, new)
        else:
            convert_sync_batchnorm(child, ranks)
    return module
