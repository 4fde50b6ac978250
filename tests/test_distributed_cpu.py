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
