#!/usr/bin/env python3
"""Small driver for rocprofv3: a few passes of the verify path (and optionally polymul) on a
resident corpus. Usage: python3 tools/prof_run.py [logn] [items] [steps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import support  # noqa: E402
from falcon_amd import capi  # noqa: E402

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 9
items = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 18
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
permille = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dev = torch.device("cuda", 0)
pk, msg, sig, _ = support.corpus().generate(logn, items, seed=77, mlen=32, malformed_permille=permille, threads=16)
d_pk, d_msg, d_sig = (torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (pk, msg, sig))
d_acc = torch.zeros(items, dtype=torch.uint8, device=dev)
ctx = capi.Context(0)
for _ in range(steps):
    ctx.verify_many_dev(logn, items, d_pk.data_ptr(), d_msg.data_ptr(), None, 32, d_sig.data_ptr(), d_acc.data_ptr(),
                        stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print("accepted", int(d_acc.sum().item()), "of", items)
