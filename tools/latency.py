#!/usr/bin/env python3
"""Latency of small host-buffer batches (fa_verify_many): python tools/latency.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
torch.cuda.init()
import support
from falcon_amd import capi
ctx = capi.Context(0)
for logn in (9, 10):
    pk, msg, sig, _ = support.corpus().generate(logn, 4096, seed=5)
    for n in (1, 16, 256, 4096):
        ctx.verify_many(logn, pk[:n], msg[:n], sig[:n])
        t0 = time.perf_counter(); reps = 50
        for _ in range(reps):
            acc = ctx.verify_many(logn, pk[:n], msg[:n], sig[:n])
        dt = (time.perf_counter() - t0) / reps
        print(f"Falcon-{1 << logn} batch of {n}: {dt * 1e6:.0f} us per call, {dt * 1e6 / n:.2f} us per verify, all accepted: {bool(acc.all())}")
