#!/usr/bin/env python3
"""Regenerates tests/golden/ref_outputs.npz: OUTPUTS OF THE REFERENCE ITSELF for the inputs of corpus_small.npz and
a few polynomials. Run in the container that holds /root/reference, after `make -C oracle ref` has built
oracle/_ref/libfalcon_ref.so (oracle/ref_harness.cpp over the reference's own headers). Per item and N:
decode_pkey (bool, h), decode_sig (bool, salt, s2, called on zeroed outputs), and -- where both succeed -- s1 and
the exact norm of verification.hpp:36-79 executed with the reference's ntt / polynomial functions on the fixture's
c (hash_to_point cannot be built from the reference: absent sha3 submodule; c is pinned by hashlib). Also ntt / intt
of four polynomials per N. The fixture is data (inputs and the reference's outputs), no reference source.

Usage: python tests/golden/make_ref_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import support  # noqa: E402

ref = support.reference()
assert ref is not None, "build oracle/_ref first: make -C oracle ref"
g = np.load(os.path.join(HERE, "corpus_small.npz"), allow_pickle=False)
out = {}
for logn in (9, 10):
    n = 1 << logn
    pk, sig, c = g[f"pk{logn}"], g[f"sig{logn}"], g[f"c{logn}"]
    count = pk.shape[0]
    pk_ok, sig_ok = np.zeros(count, np.uint8), np.zeros(count, np.uint8)
    h = np.zeros((count, n), np.uint16)
    salt = np.zeros((count, 40), np.uint8)
    s2 = np.zeros((count, n), np.int16)
    s1 = np.zeros((count, n), np.int16)
    norm = np.zeros(count, np.uint64)
    for i in range(count):
        ok, hi = ref.decode_pkey(logn, pk[i])
        ok2, sa, s2i = ref.decode_sig(logn, sig[i])
        pk_ok[i], sig_ok[i], h[i], salt[i], s2[i] = ok, ok2, hi, sa, s2i
        if ok and ok2:
            s1[i], nv = ref.verify_after_hash(logn, hi, c[i], s2i)
            norm[i] = nv
    rng = np.random.default_rng(logn)
    polys = np.stack([np.arange(n) % 12289, np.full(n, 12288), rng.integers(0, 12289, n), rng.integers(0, 12289, n)]).astype(np.uint16)
    out.update({f"pk_ok{logn}": pk_ok, f"h{logn}": h, f"sig_ok{logn}": sig_ok, f"salt{logn}": salt, f"s2_{logn}": s2,
                f"s1_{logn}": s1, f"norm{logn}": norm, f"poly{logn}": polys,
                f"ntt{logn}": np.stack([ref.ntt(p, logn) for p in polys]),
                f"intt{logn}": np.stack([ref.ntt(p, logn, inverse=True) for p in polys])})
np.savez_compressed(os.path.join(HERE, "ref_outputs.npz"), **out)
print("wrote ref_outputs.npz", {k: v.shape for k, v in out.items() if k.startswith("h")})
