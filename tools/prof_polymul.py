#!/usr/bin/env python3
"""rocprofv3 / tools/ab_cycles.sh driver: the batched negacyclic product (BASELINE config 2) on resident inputs.
Usage: python3 tools/prof_polymul.py [logn] [polys] [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from falcon_amd import capi  # noqa: E402

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 9
polys = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 21
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(5)
a = torch.randint(0, 12289, (polys, 1 << logn), dtype=torch.int32, device=dev, generator=g).to(torch.int16)
b = torch.randint(0, 12289, (polys, 1 << logn), dtype=torch.int32, device=dev, generator=g).to(torch.int16)
c = torch.empty_like(a)
ctx = capi.Context(0)
for _ in range(steps):
    ctx.polymul_many_dev(logn, polys, a.data_ptr(), b.data_ptr(), c.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print("checksum", int(c.to(torch.int64).sum().item()))
