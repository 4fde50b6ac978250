"""N>1 plumbing of the data-parallel path with gloo, world size 2, on CPU (the GPU path uses the same Ranks class
with RCCL).  Spawns two real processes that rendezvous on 127.0.0.1."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, json, time
    sys.path.insert(0, %r)
    from attentionshift_amd.dist import Ranks
    r = Ranks(backend="gloo")
    assert r.world == 2
    lo, hi = r.shard(5)                       # 5 images over 2 ranks -> 3 + 2
    r.barrier()
    t = 0.010 * (r.rank + 1)                  # rank 1 is the slow one
    tmax = r.max_over_ranks(t)
    n = r.sum_over_ranks(hi - lo)
    r.barrier()
    print(json.dumps(dict(rank=r.rank, lo=lo, hi=hi, tmax=tmax, n=n)), flush=True)
    r.close()
""") % ROOT


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo_shard_barrier_and_max():
    import json
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=120)
        assert p.returncode == 0, e[-2000:]
        outs.append(json.loads(o.strip().splitlines()[-1]))
    outs.sort(key=lambda d: d["rank"])
    assert (outs[0]["lo"], outs[0]["hi"], outs[1]["lo"], outs[1]["hi"]) == (0, 3, 3, 5)
    assert all(abs(o["tmax"] - 0.020) < 1e-9 for o in outs)          # the slowest rank defines the step time
    assert all(o["n"] == 5 for o in outs)


def test_single_process_is_a_no_op_world():
    from attentionshift_amd.dist import Ranks
    env = {k: os.environ.pop(k, None) for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    try:
        r = Ranks()
        assert r.world == 1 and r.shard(7) == (0, 7) and r.max_over_ranks(1.5) == 1.5
        r.barrier()
        r.close()
    finally:
        for k, v in env.items():
            if v is not None:
                os.environ[k] = v


GRAD_WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, %r)
    import torch
    from attentionshift_amd.dist import Ranks, GradAllReducer
    r = Ranks(backend="gloo")
    torch.manual_seed(0)                                   # identical replicas
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.GELU(), torch.nn.Linear(16, 4), torch.nn.Linear(4, 3))
    unused = torch.nn.Parameter(torch.ones(5))             # never receives a gradient: must average to zeros
    params = list(net.parameters()) + [unused]
    red = GradAllReducer(params, r, bucket_mb=0.0005, comm_dtype=torch.float32)   # ~128 floats per bucket -> several buckets
    nb = len(red.buckets)
    res = []
    for step in range(2):                                  # two steps: bucket state must reset
        for p in params:
            p.grad = None
        x = torch.randn(6, 8, generator=torch.Generator().manual_seed(100 + 10 * step + r.rank))   # each rank: own data
        net(x).square().sum().backward()
        local = [p.grad.clone() for p in net.parameters()]
        red.finish()
        res.append(dict(avg=[p.grad.flatten().tolist() for p in net.parameters()],
                        local=[g.flatten().tolist() for g in local], unused=unused.grad.tolist()))
    print(json.dumps(dict(rank=r.rank, nb=nb, res=res)), flush=True)
    red.close()
    r.close()
""") % ROOT


def test_two_rank_gloo_bucketed_gradient_allreduce():
    """GradAllReducer (the DDP gradient exchange of the trainable backbone): bucketed, hook-launched async all-reduce
    averages the per-rank gradients; parameters without a gradient contribute zeros; state resets between steps."""
    import json
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", GRAD_WORKER], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=180)
        assert p.returncode == 0, e[-2000:]
        outs.append(json.loads(o.strip().splitlines()[-1]))
    outs.sort(key=lambda d: d["rank"])
    assert outs[0]["nb"] >= 2
    for step in range(2):
        a, b = outs[0]["res"][step], outs[1]["res"][step]
        for i in range(len(a["avg"])):
            want = [(x + y) / 2 for x, y in zip(a["local"][i], b["local"][i])]
            for got in (a["avg"][i], b["avg"][i]):
                assert max(abs(g - w) for g, w in zip(got, want)) < 1e-5
        assert a["unused"] == [0.0] * 5 and b["unused"] == [0.0] * 5


SYNCBN_WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, %r)
    import torch
    from attentionshift_amd.dist import Ranks, convert_sync_batchnorm, SyncBatchNorm2d
    r = Ranks(backend="gloo")
    gen = torch.Generator().manual_seed(5)
    x_all = torch.randn(6, 4, 5, 3, generator=gen) * 2 + 1            # global batch: rank 0 takes 4 images, rank 1 two
    w_out = torch.randn(6, 4, 5, 3, generator=gen)
    torch.manual_seed(0)
    ref = torch.nn.Sequential(torch.nn.Conv2d(4, 4, 1), torch.nn.BatchNorm2d(4))
    with torch.no_grad():
        ref[1].weight.copy_(torch.tensor([1.5, 0.5, -1.0, 2.0])); ref[1].bias.copy_(torch.tensor([0.1, -0.2, 0.3, 0.0]))
    import copy
    net = convert_sync_batchnorm(copy.deepcopy(ref), r)
    assert isinstance(net[1], SyncBatchNorm2d) and list(net.state_dict()) == list(ref.state_dict())
    lo, hi = (0, 4) if r.rank == 0 else (4, 6)
    x = x_all[lo:hi].clone().requires_grad_(True)
    y = net(x)
    (y * w_out[lo:hi]).sum().backward()
    xr = x_all.clone().requires_grad_(True)                             # the single-process reference on the whole batch
    yr = ref(xr)
    (yr * w_out).sum().backward()
    gw = net[1].weight.grad.clone(); gb = net[1].bias.grad.clone(); gc = net[0].weight.grad.clone()
    for g in (gw, gb, gc):
        r.dist.all_reduce(g)                                            # what the gradient all-reduce would sum
    ok = dict(
        y=bool(torch.allclose(y, yr[lo:hi], atol=1e-5)), dx=bool(torch.allclose(x.grad, xr.grad[lo:hi], atol=1e-5)),
        gw=bool(torch.allclose(gw, ref[1].weight.grad, atol=1e-4)), gb=bool(torch.allclose(gb, ref[1].bias.grad, atol=1e-4)),
        gc=bool(torch.allclose(gc, ref[0].weight.grad, atol=1e-4)),
        rm=bool(torch.allclose(net[1].running_mean, ref[1].running_mean, atol=1e-6)),
        rv=bool(torch.allclose(net[1].running_var, ref[1].running_var, atol=1e-5)),
        nb=int(net[1].num_batches_tracked) == 1)
    net.eval()
    ok["eval"] = bool(torch.allclose(net(x_all[lo:hi]), ref.eval()(x_all[lo:hi]), atol=1e-5))
    print(json.dumps(dict(rank=r.rank, **ok)), flush=True)
    r.close()
""") % ROOT


def test_sync_batchnorm_matches_single_process_batchnorm_on_the_global_batch():
    import json
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", SYNCBN_WORKER], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    for p in procs:
        o, e = p.communicate(timeout=180)
        assert p.returncode == 0, e[-3000:]
        rec = json.loads(o.strip().splitlines()[-1])
        assert all(v is True for k, v in rec.items() if k != "rank"), rec


def test_sync_batchnorm_single_rank_is_plain_batchnorm():
    import torch
    from attentionshift_amd.dist import Ranks, SyncBatchNorm2d
    env = {k: os.environ.pop(k, None) for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    try:
        torch.manual_seed(0)
        a, b = torch.nn.BatchNorm2d(3), SyncBatchNorm2d(3, Ranks())
        x = torch.randn(4, 3, 5, 5)
        assert torch.equal(a(x), b(x)) and torch.equal(a.running_var, b.running_var)
    finally:
        for k, v in env.items():
            if v is not None:
                os.environ[k] = v


TRAIN_WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
    import torch
    import test_forward_train as TF
    from attentionshift_amd import mae_heads
    from attentionshift_amd.dist import Ranks, GradAllReducer, parse_losses
    mae_heads._Attention.forward = TF._torch_attention          # CPU stand-in for the HIP small-N attention
    r = Ranks(backend="gloo")
    torch.manual_seed(0)                                         # identical replicas
    stem = torch.nn.Conv2d(3, 48, 16, 16)                        # stands in for the backbone: image -> stride-16 map
    head = TF._head()
    params = [p for p in list(stem.parameters()) + list(head.parameters()) if p.requires_grad]
    d = TF._inputs(torch.Generator().manual_seed(1))
    img = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(7))
    empty_rank = int(os.environ.get("EMPTY_RANK", "-1"))         # this rank's image has no object: no mask-head gradient

    def losses_of(i, micro=0):
        gts, labels = d["gts"][i], d["labels"][i]
        if i == empty_rank:
            gts, labels = gts[:0], labels[:0]
        fmap = stem(img[i:i + 1] + 0.1 * micro)
        return head.forward_train(
            fmap, [d["metas"][i]], [d["props"][i]], [gts], [labels], point_cls=d["point_cls"][i:i + 1],
            point_reg=d["point_reg"][i:i + 1], gt_points=[d["gt_points"][i][:gts.shape[0]]], gt_points_labels=[labels],
            mask_point_coords=[d["coords"][i][:gts.shape[0]]], mask_point_labels=[d["plabels"][i][:gts.shape[0]]],
            semantic_centers_split=[d["centres"][i][:gts.shape[0]]], generator=torch.Generator().manual_seed(5 + i))

    def grads():
        return [None if p.grad is None else p.grad.clone() for p in params]

    # ---- single-process reference: every rank computes BOTH images' steps and averages by hand ----
    accum = int(os.environ.get("ACCUM", "1"))
    want, logs = None, []
    for i in range(2):
        for p in params:
            p.grad = None
        for micro in range(accum):
            loss, lv = parse_losses(losses_of(i, micro))
            (loss / accum).backward()
            if micro == accum - 1:
                logs.append(lv)
        g = [torch.zeros_like(p) if p.grad is None else p.grad.clone() for p in params]
        want = g if want is None else [a + b for a, b in zip(want, g)]
    want = [w / 2 for w in want]
    # ---- the data-parallel step: this rank's image only ----
    for p in params:
        p.grad = None
    red = GradAllReducer(params, r, bucket_mb=0.02, comm_dtype=torch.float32)
    for micro in range(accum):
        loss, lv = parse_losses(losses_of(r.rank, micro), r)
        if micro < accum - 1:
            with red.no_sync():
                (loss / accum).backward()
        else:
            (loss / accum).backward()
    red.finish()
    err = max(float((p.grad - w).abs().max() / (w.abs().max() + 1e-12)) for p, w in zip(params, want))
    mean_logs = {k: 0.5 * (logs[0][k] + logs[1][k]) for k in logs[0] if k in logs[1]}
    log_err = max(abs(lv[k] - v) for k, v in mean_logs.items())
    print(json.dumps(dict(rank=r.rank, nb=len(red.buckets), err=err, log_err=log_err, keys=sorted(lv))), flush=True)
    red.close()
    r.close()
""") % (ROOT, ROOT)


def _run_train_workers(extra_env):
    import json
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="2", **extra_env)
        procs.append(subprocess.Popen([sys.executable, "-c", TRAIN_WORKER], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=300)
        assert p.returncode == 0, e[-3000:]
        outs.append(json.loads(o.strip().splitlines()[-1]))
    return outs


def test_two_rank_training_step_reproduces_single_process_gradients():
    """The TRAINING STEP, not just the reducer: each rank runs the RoI head's forward_train (point / box / mask losses on
    its own image) + parse_losses + backward with the hook-launched bucketed all-reduce; the averaged gradients of every
    parameter (stem, box head, mask head) equal the hand-averaged single-process gradients, and the logged losses equal
    the rank means (one fused all-reduce, base.py:185-218)."""
    outs = _run_train_workers({})
    for o in outs:
        assert o["nb"] >= 2 and o["err"] < 1e-5 and o["log_err"] < 1e-5, o
        assert "loss" in o["keys"] and "loss_mask" in o["keys"]


def test_two_rank_training_step_with_a_rank_that_skips_the_mask_head():
    """Rank 1's image has no object: its mask head (and box regression) get NO gradient there while rank 0's do -- the
    collectives still pair up because buckets are launched in bucket order on every rank."""
    outs = _run_train_workers({"EMPTY_RANK": "1"})
    for o in outs:
        assert o["err"] < 1e-5, o


def test_two_rank_gradient_accumulation_with_no_sync():
    """update_interval = 2 (mmdet/utils/optimizer.py:23-32): the first micro-step only accumulates (no_sync), the second
    reduces the accumulated gradients."""
    outs = _run_train_workers({"ACCUM": "2"})
    for o in outs:
        assert o["err"] < 1e-5, o


BCAST_WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, %r)
    import torch
    from attentionshift_amd.dist import Ranks, GradAllReducer
    r = Ranks(backend="gloo")
    torch.manual_seed(1000 + r.rank)                       # DIFFERENT replicas: only the broadcast can make them agree
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 1), torch.nn.BatchNorm2d(4), torch.nn.Flatten(), torch.nn.Linear(4 * 4, 3))
    net[1].running_mean.add_(r.rank + 1.0)                 # buffers differ too (incl. the int64 num_batches_tracked)
    net[1].num_batches_tracked.add_(3 * r.rank + 1)
    flag = torch.nn.Parameter(torch.tensor([bool(r.rank)] * 3), requires_grad=False)     # bool buffer-like tensor
    before = [p.detach().clone() for p in net.parameters()]
    comm = os.environ.get("COMM", "float32")
    red = GradAllReducer(net.parameters(), r, bucket_mb=0.0002, comm_dtype=getattr(torch, comm),
                         buffers=list(net.buffers()) + [flag])
    state = [t.detach().double().flatten().tolist() for t in list(net.parameters()) + list(net.buffers()) + [flag]]
    changed = any(not torch.equal(a, p.detach()) for a, p in zip(before, net.parameters()))
    # one step on rank-specific data: gradients averaged; with the 16-bit wire the values are pre-scaled by 1/world
    x = torch.randn(5, 3, 2, 2, generator=torch.Generator().manual_seed(50 + r.rank))
    net(x).square().sum().backward()
    local = [p.grad.clone() for p in net.parameters()]
    red.finish()
    print(json.dumps(dict(rank=r.rank, state=state, changed=changed, ncoll=red.broadcasts, nb=len(red.buckets),
                          avg=[p.grad.flatten().tolist() for p in net.parameters()],
                          local=[g.flatten().tolist() for g in local])), flush=True)
    red.close()
    r.close()
""") % ROOT


def _spawn2(worker, extra_env=None, timeout=180):
    import json
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, "-c", worker], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=timeout)
        assert p.returncode == 0, e[-3000:]
        outs.append(json.loads(o.strip().splitlines()[-1]))
    outs.sort(key=lambda d: d["rank"])
    return outs


def test_reducer_broadcasts_rank0_parameters_and_buffers_at_construction():
    """mmdet/apis/train.py:96-100: MMDistributedDataParallel broadcasts rank 0's state when it wraps the model.  Ranks
    seeded differently (or of which only rank 0 loaded a checkpoint) must start from identical weights and buffers."""
    a, b = _spawn2(BCAST_WORKER)
    assert a["state"] == b["state"]                      # bit-identical after construction (fp32, int64 and bool tensors)
    assert not a["changed"] and b["changed"]             # rank 0 is the source, rank 1 was overwritten
    assert 1 <= a["ncoll"] <= 4                          # packed by dtype: a few broadcasts, not one per tensor
    for i in range(len(a["avg"])):
        want = [(x + y) / 2 for x, y in zip(a["local"][i], b["local"][i])]
        for got in (a["avg"][i], b["avg"][i]):
            assert max(abs(g - w) for g, w in zip(got, want)) < 1e-5 * (1 + max(abs(w) for w in want))


def test_bf16_wire_is_prescaled_by_the_world_size():
    """comm_dtype=bfloat16: gradients are scaled by 1/world while they are packed (the running sum stays in one rank's
    range), the result is the rank mean to bf16 precision and both ranks hold the same bits."""
    a, b = _spawn2(BCAST_WORKER, {"COMM": "bfloat16"})
    assert a["avg"] == b["avg"]
    for i in range(len(a["avg"])):
        want = [(x + y) / 2 for x, y in zip(a["local"][i], b["local"][i])]
        scale = max(abs(w) for w in want) + 1e-12
        assert max(abs(g - w) for g, w in zip(a["avg"][i], want)) < 1.2e-2 * scale


def test_forced_one_rank_group_runs_the_real_reducer_path():
    """Ranks(force=True) / AS_FORCE_DIST=1: a real one-rank process group, so the bucket copy, the hook-launched
    all-reduce, finish() and the write-back all execute; gradients equal the plain step's."""
    import torch
    from attentionshift_amd.dist import GradAllReducer, Ranks, parse_losses
    env = {k: os.environ.pop(k, None) for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    grad_was = torch.is_grad_enabled()
    torch.set_grad_enabled(True)           # (the GPU test modules switch autograd off at import time)
    try:
        r = Ranks(backend="gloo", force=True)
        assert r.world == 1 and r.active and r.dist is not None
        r.barrier()
        assert r.max_over_ranks(2.5) == 2.5 and r.sum_over_ranks(3) == 3.0
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.GELU(), torch.nn.Linear(16, 4))
        x = torch.randn(6, 8)
        net(x).square().sum().backward()
        want = [p.grad.clone() for p in net.parameters()]
        for comm, tol in ((torch.float32, 0.0), (torch.bfloat16, 8e-3)):
            for p in net.parameters():
                p.grad = None
            red = GradAllReducer(net.parameters(), r, bucket_mb=0.0003, comm_dtype=comm)
            assert red.active and len(red.buckets) >= 2 and red.broadcasts >= 1
            with red.no_sync():
                (net(x).square().sum() * 0.5).backward()
            (net(x).square().sum() * 0.5).backward()         # accumulates onto the no_sync micro-step
            red.finish()
            for p, w in zip(net.parameters(), want):
                assert float((p.grad - w).abs().max()) <= tol * float(w.abs().max()) + (1e-6 if tol == 0 else 0)
            red.close()
        loss, logs = parse_losses({"loss_a": torch.tensor([1.0, 3.0]), "acc": torch.tensor(5.0)}, r)
        assert float(loss) == 2.0 and logs == {"loss_a": 2.0, "acc": 5.0, "loss": 2.0}
        # a second Ranks() finds the live group and REUSES it (mmcv's init_dist creates the group before the model is
        # built); closing the borrower leaves the owner's group alive
        r2 = Ranks()
        assert r2.dist is not None and r2.world == 1 and not r2._own_group
        r2.close()
        assert torch.distributed.is_initialized()
        r.close()
        assert not torch.distributed.is_initialized()
    finally:
        torch.set_grad_enabled(grad_was)
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
        for k, v in env.items():
            if v is not None:
                os.environ[k] = v
