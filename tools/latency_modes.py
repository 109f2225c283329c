import os, sys, time
ROOT="/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
torch.cuda.init()
import support
from falcon_amd import capi
ctx = capi.Context(0)
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
for logn in (9, 10):
    pk, msg, sig, _ = support.corpus().generate(logn, 64, seed=5)
    for n in (1, 64):
        t = [torch.from_numpy(np.ascontiguousarray(a[:n])).to(dev) for a in (pk, msg, sig)]
        acc = torch.zeros(n, dtype=torch.uint8, device=dev)
        def sync_call(): ctx.verify_many_dev(logn, n, t[0].data_ptr(), t[1].data_ptr(), None, 32, t[2].data_ptr(), acc.data_ptr(), stream=st)
        def noalloc(): ctx.verify_many_dev_noalloc(logn, n, t[0].data_ptr(), t[1].data_ptr(), None, 32, t[2].data_ptr(), acc.data_ptr(), stream=st); torch.cuda.synchronize()
        def asyn(): ctx.verify_many_dev(logn, n, t[0].data_ptr(), t[1].data_ptr(), None, 32, t[2].data_ptr(), acc.data_ptr(), stream=st); ctx.finish(st)
        res = {}
        for name, fn, pre in (("sync", sync_call, 0), ("noalloc+sync", noalloc, 0), ("async+finish", asyn, 1)):
            ctx.set_option(capi.OPT_ASYNC, pre)
            for _ in range(10): fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200): fn()
            torch.cuda.synchronize()
            res[name] = (time.perf_counter() - t0) / 200 * 1e6
            ctx.set_option(capi.OPT_ASYNC, 0)
        print(f"Falcon-{1<<logn} n={n}:", {k: round(v, 1) for k, v in res.items()}, flush=True)
