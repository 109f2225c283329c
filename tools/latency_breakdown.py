#!/usr/bin/env python3
"""Where a small batch's time goes: host-buffer call, device-resident call, and the kernels' own HIP-event times."""
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
    pk, msg, sig, _ = support.corpus().generate(logn, 65536, seed=5)
    for n, ov in [(n, ov) for n in (1, 64, 1024, 4096, 8192, 16384, 32768) for ov in (1 << 20, 0)]:
        ctx.set_option(capi.OPT_PAIR_HASH_ITEMS, ov)
        d_pk, d_msg, d_sig = (torch.from_numpy(np.ascontiguousarray(a[:n])).to(dev) for a in (pk, msg, sig))
        d_acc = torch.zeros(n, dtype=torch.uint8, device=dev)
        reps = 50
        def dev_call():
            ctx.verify_many_dev(logn, n, d_pk.data_ptr(), d_msg.data_ptr(), None, 32, d_sig.data_ptr(), d_acc.data_ptr(), stream=stream)
        dev_call(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): dev_call()
        torch.cuda.synchronize()
        t_dev = (time.perf_counter() - t0) / reps
        ctx.verify_many(logn, pk[:n], msg[:n], sig[:n])
        t0 = time.perf_counter()
        for _ in range(reps): ctx.verify_many(logn, pk[:n], msg[:n], sig[:n])
        t_host = (time.perf_counter() - t0) / reps
        ctx.set_option(capi.OPT_PROFILE, 1); ctx.get_profile()
        for _ in range(reps): dev_call()
        p = ctx.get_profile(); ctx.set_option(capi.OPT_PROFILE, 0)
        print(f"Falcon-{1 << logn} n={n} pair_hash={ov}: host call {t_host*1e6:.0f} us, device call {t_dev*1e6:.0f} us, kernels "
              f"decode {p['ms_decode']*1e3/reps:.0f} + hash {p['ms_hash']*1e3/reps:.0f} + verify {p['ms_verify']*1e3/reps:.0f} us", flush=True)
