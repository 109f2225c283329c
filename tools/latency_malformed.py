import os, sys, time
ROOT="/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
torch.cuda.init()
import support
from falcon_amd import capi
ctx = capi.Context(0)
dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
for logn in (9, 10):
    for kind, name in ((0, "valid"), (7, "k>=8 window, accepted"), (8, "random body"), (5, "unary run past the end")):
        pk, msg, sig, _ = support.corpus().generate(logn, 8, seed=5, kind=kind)
        t = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (pk, msg, sig)]
        acc = torch.zeros(8, dtype=torch.uint8, device=dev)
        fn = lambda: ctx.verify_many_dev(logn, 8, t[0].data_ptr(), t[1].data_ptr(), None, 32, t[2].data_ptr(), acc.data_ptr(), stream=st)
        for _ in range(10): fn()
        t0 = time.perf_counter()
        for _ in range(200): fn()
        print(f"Falcon-{1<<logn} 8 items, {name} ({support.KINDS[kind]}): {(time.perf_counter()-t0)/200*1e6:.1f} us per call, accepted {int(acc.sum())}", flush=True)
