import sys, os, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = ["kernel_bench.py", "--reps", "30"]
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import kernel_bench as KB
from attentionshift_amd import ops, _lib
orig = ops.cosine_shift
def timed(*a, **k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = orig(*a, **k)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    timed.ts.append(dt)
    return r
timed.ts = []
ops.cosine_shift = timed
KB.main()
print("per-call ms (sync'd):", [round(x, 2) for x in timed.ts[:8]], "...", [round(x, 2) for x in timed.ts[-4:]])
print(torch.cuda.memory_stats()["num_alloc_retries"], torch.cuda.memory_reserved() / 1e9)
