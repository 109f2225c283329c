#!/usr/bin/env python3
"""Regenerates tests/golden/corpus_small.npz: a small mixed corpus (every item class of
corpus/falcon_corpus.c, both N, several message lengths) with the outputs of the CPU oracle
(oracle/falcon_oracle.c) for it. The oracle itself is pinned by readme_kat_falcon512.json, which
comes from the reference's README.md:308-312; the reference cannot be built here (absent sha3
submodule), so these vectors are oracle outputs, not reference outputs.

Usage: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import support  # noqa: E402

out = {}
for logn in (9, 10):
    pks, msgs, sigs, kinds, mlens = [], [], [], [], []
    for kind in range(len(support.KINDS)):
        for j, mlen in enumerate((32, 0, 137, 300)):
            pk, msg, sig, k = support.corpus().generate(logn, 1, seed=1000 * logn + 10 * kind + j, kind=kind, mlen=mlen)
            pks.append(pk[0]); sigs.append(sig[0]); msgs.append(msg[0]); kinds.append(kind); mlens.append(mlen)
    pk = np.stack(pks); sig = np.stack(sigs)
    off = np.concatenate([[0], np.cumsum(mlens)]).astype(np.uint64)
    flat = np.concatenate([m for m in msgs]).astype(np.uint8)
    flags, dbg = support.oracle().verify_many(logn, pk, flat, sig, msg_off=off, debug=True)
    n = 1 << logn
    out.update({f"pk{logn}": pk, f"sig{logn}": sig, f"msg{logn}": flat, f"off{logn}": off,
                f"kind{logn}": np.array(kinds, np.uint8), f"flags{logn}": flags,
                f"status{logn}": dbg["status"].astype(np.uint8), f"norm{logn}": dbg["norm_exact"],
                f"c{logn}": dbg["c"][:, :n].copy(), f"s2_{logn}": dbg["s2"][:, :n].copy(),
                f"s1_{logn}": dbg["s1"][:, :n].copy()})
np.savez_compressed(os.path.join(HERE, "corpus_small.npz"), **out)
print("wrote corpus_small.npz", {k: v.shape for k, v in out.items() if k.startswith("pk")})
