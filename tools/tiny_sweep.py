#!/usr/bin/env python3
"""Device-resident call time against batch size with the one-wave-per-item front on and off (FA_OPT_TINY_BATCH_ITEMS)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
torch.cuda.init()
import support
from falcon_amd import capi
ctx = capi.Context(0)
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
for logn in (9, 10):
    pk, msg, sig, _ = support.corpus().generate(logn, 4096, seed=5)
    for n in (32, 64, 128, 256, 512, 1024, 2048, 4096):
        d_pk, d_msg, d_sig = (torch.from_numpy(np.ascontiguousarray(a[:n])).to(dev) for a in (pk, msg, sig))
        d_acc = torch.zeros(n, dtype=torch.uint8, device=dev)
        out = []
        for tiny in (1 << 20, 0):
            ctx.set_option(capi.OPT_TINY_BATCH_ITEMS, tiny)
            def dev_call():
                ctx.verify_many_dev(logn, n, d_pk.data_ptr(), d_msg.data_ptr(), None, 32, d_sig.data_ptr(), d_acc.data_ptr(), stream=stream)
            for _ in range(5): dev_call()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(100): dev_call()
            torch.cuda.synchronize()
            out.append((time.perf_counter() - t0) / 100 * 1e6)
            assert bool(d_acc.all())
        print(f"Falcon-{1 << logn} n={n}: device call {out[0]:.0f} us one wave per item, {out[1]:.0f} us lane per item", flush=True)
