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
