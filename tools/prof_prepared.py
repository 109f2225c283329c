#!/usr/bin/env python3
"""rocprofv3 driver: the prepared-key verify path (NTT(h) resident). Usage: python3 tools/prof_prepared.py [logn] [items] [steps]"""
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
items = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda", 0)
pk, msg, sig, _ = support.corpus().generate(logn, items, seed=77, mlen=32, threads=16)
d_pk, d_msg, d_sig = (torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (pk, msg, sig))
d_acc = torch.zeros(items, dtype=torch.uint8, device=dev)
d_prep = torch.empty((items, 1 << logn), dtype=torch.int16, device=dev)
ctx = capi.Context(0)
s = torch.cuda.current_stream().cuda_stream
ctx.prepare_keys_dev(logn, items, d_pk.data_ptr(), None, d_prep.data_ptr(), stream=s)
for _ in range(steps):
    ctx.verify_many_prepared_dev(logn, items, d_prep.data_ptr(), None, d_msg.data_ptr(), None, 32, d_sig.data_ptr(), d_acc.data_ptr(), stream=s)
torch.cuda.synchronize()
print("accepted", int(d_acc.sum().item()), "of", items)
