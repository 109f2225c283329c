#!/usr/bin/env python3
"""Driver for rocprofv3 --kernel-trace --stats: 200 device-resident calls of ONE item and 200 of 64 items for each N
(the one-wave-per-item front, tiny_kernel.hpp). Usage: python3 tools/prof_tiny.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import support  # noqa: E402
from falcon_amd import capi  # noqa: E402

dev = torch.device("cuda", 0)
ctx = capi.Context(0)
st = torch.cuda.current_stream().cuda_stream
for logn in (9, 10):
    pk, msg, sig, _ = support.corpus().generate(logn, 64, seed=78, mlen=32, threads=4)
    d_pk, d_msg, d_sig = (torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (pk, msg, sig))
    d_acc = torch.zeros(64, dtype=torch.uint8, device=dev)
    for n in (1, 64):
        for _ in range(200):
            ctx.verify_many_dev(logn, n, d_pk.data_ptr(), d_msg.data_ptr(), None, 32, d_sig.data_ptr(), d_acc.data_ptr(), stream=st)
    torch.cuda.synchronize()
    print("accepted", int(d_acc.sum().item()), "of 64")
