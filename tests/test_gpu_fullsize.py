"""GPU, BASELINE.json's full batch sizes: properties that need no oracle run (the oracle covers the
small cases item by item in test_gpu_parity.py; here it is sampled).

  config 2  batched negacyclic products, n = 512, 2^24 polynomials: identity, multiplication by x,
            commutativity and linearity, checked element by element with torch integer ops;
  config 3  Falcon-512, 2^20 distinct valid triples: all accepted; tampering a known subset of
            messages rejects exactly that subset; the result does not depend on item order;
  config 4  Falcon-1024, 2^20 triples with the malformed mix: the accept byte equals what the class
            of each item dictates, and a sample agrees with the oracle under both norm policies.
  config 5  one GPU's share of the mixed batch: 2^23 items, 50/50 Falcon-512 / Falcon-1024 (2^22 each, about 20 GB),
            tiled on the device from 2^20-item corpora and verified in place by fa_verify_mixed_dev: the accept byte
            of every item equals what its class dictates, a sample agrees with the oracle.
"""
import numpy as np
import pytest

import support

pytestmark = pytest.mark.gpu
Q = 12289
VALID_KINDS = (0, 7, 12)  # valid, k >= 8 coded but decodable, pk coefficients >= q


def _dev(ctx, logn, pk, msg, sig, policy=0):
    import torch
    dev = torch.device("cuda", 0)
    t = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (pk, msg, sig)]
    acc = torch.full((pk.shape[0],), 7, dtype=torch.uint8, device=dev)
    ctx.verify_many_dev(logn, pk.shape[0], t[0].data_ptr(), t[1].data_ptr(), None, msg.shape[1], t[2].data_ptr(),
                        acc.data_ptr(), norm_policy=policy, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return acc.cpu().numpy()


def test_falcon512_one_million_valid_tamper_permute(ctx, corpus, oracle):
    n = 1 << 20
    pk, msg, sig, _ = corpus.generate(9, n, seed=0xC0FFEE, threads=16)
    acc = _dev(ctx, 9, pk, msg, sig)
    assert acc.all()
    bad = msg.copy()
    hit = np.arange(n) % 7 == 3
    bad[hit, 5] ^= 0x40
    acc2 = _dev(ctx, 9, pk, bad, sig)
    np.testing.assert_array_equal(acc2, (~hit).astype(np.uint8))
    perm = np.random.default_rng(1).permutation(n)
    acc3 = _dev(ctx, 9, pk[perm], bad[perm], sig[perm])
    np.testing.assert_array_equal(acc3, acc2[perm])
    idx = np.arange(0, n, 997)
    want = oracle.verify_many(9, pk[idx], bad[idx], sig[idx], threads=16)[0] & 1
    np.testing.assert_array_equal(acc2[idx], want)


def test_falcon1024_one_million_malformed_mix(ctx, corpus, oracle):
    n = 1 << 20
    pk, msg, sig, kinds = corpus.generate(10, n, seed=0xBADF00D, malformed_permille=250, threads=16)
    acc = _dev(ctx, 10, pk, msg, sig)
    want = np.isin(kinds, VALID_KINDS).astype(np.uint8)
    np.testing.assert_array_equal(acc, want)
    assert 0.2 < (kinds != 0).mean() < 0.3
    idx = np.flatnonzero(kinds != 0)[:4096]
    flags = oracle.verify_many(10, pk[idx], msg[idx], sig[idx], threads=16)[0]
    np.testing.assert_array_equal(acc[idx], flags & 1)
    accw = _dev(ctx, 10, pk[idx], msg[idx], sig[idx], policy=1)  # wrap32 == a g++-built reference
    np.testing.assert_array_equal(accw, (flags >> 1) & 1)


def test_polymul_sixteen_million(ctx):
    import torch
    dev = torch.device("cuda", 0)
    n, count = 512, 1 << 24
    stream = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device=dev).manual_seed(5)

    def rand_polys():
        out = torch.empty((count, n), dtype=torch.int16, device=dev)
        for lo in range(0, count, 1 << 21):  # bounded temporaries
            out[lo:lo + (1 << 21)] = torch.randint(0, Q, (1 << 21, n), generator=g, device=dev, dtype=torch.int32).to(torch.int16)
        return out

    def mul(a, b):
        c = torch.empty_like(a)
        ctx.polymul_many_dev(9, count, a.data_ptr(), b.data_ptr(), c.data_ptr(), stream=stream)
        return c

    a = rand_polys()
    b = torch.zeros_like(a)
    b[:, 0] = 1
    assert torch.equal(mul(a, b), a)                      # a * 1 = a
    b[:, 0] = 0
    b[:, 1] = 1
    c = mul(a, b)                                         # a * x: negacyclic shift
    assert torch.equal(c[:, 1:], a[:, :-1])
    assert torch.equal(c[:, 0].to(torch.int32), (Q - a[:, -1].to(torch.int32)) % Q)
    del c
    b = rand_polys()
    ab = mul(a, b)
    assert torch.equal(ab, mul(b, a))                     # commutative
    b2 = rand_polys()
    ab2 = mul(a, b2)
    for lo in range(0, count, 1 << 21):                   # a*(b + b2) = a*b + a*b2, sums reduced mod q in place
        s = slice(lo, lo + (1 << 21))
        b[s] = ((b[s].to(torch.int32) + b2[s].to(torch.int32)) % Q).to(torch.int16)
        ab[s] = ((ab[s].to(torch.int32) + ab2[s].to(torch.int32)) % Q).to(torch.int16)
    del b2, ab2
    assert torch.equal(mul(a, b), ab)
    # a sample against the oracle's NTT product and its schoolbook product
    orc = support.oracle()
    for i in (0, 12345, count - 1):
        ai, bi = a[i].cpu().numpy().astype(np.uint16), b[i].cpu().numpy().astype(np.uint16)
        want = orc.polymul(ai, bi, 9)
        assert ab[i].cpu().numpy().astype(np.uint16).tolist() == want.tolist()
        assert want.tolist() == orc.polymul(ai, bi, 9, schoolbook=True).tolist()


def test_mixed_one_gpu_share_of_config5(ctx, corpus, oracle):
    import torch
    import mixed_batch
    dev = torch.device("cuda", 0)
    base, tile = 1 << 20, 4
    p9 = corpus.generate(9, base, seed=0xC0FFEE, threads=16)
    p10 = corpus.generate(10, base, seed=0xBADF00D, malformed_permille=250, threads=16)
    npairs = base * tile
    bits = mixed_batch.pair_order_bits(npairs, 55)
    tags, pk_off, sig_off, msg_off, src = mixed_batch.layout(bits, 32)
    bits_t = torch.from_numpy(bits).to(dev)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    t9, t10 = [up(a) for a in p9[:3]], [up(a) for a in p10[:3]]
    d_pk = mixed_batch.pack_torch(bits_t, t9[0], t10[0], tile)
    d_msg = mixed_batch.pack_torch(bits_t, t9[1], t10[1], tile)
    d_sig = mixed_batch.pack_torch(bits_t, t9[2], t10[2], tile)
    del t9, t10
    d_tags, d_pko, d_sgo, d_mso = up(tags), up(pk_off.view(np.int64)), up(sig_off.view(np.int64)), up(msg_off.view(np.int64))
    acc = torch.full((2 * npairs,), 7, dtype=torch.uint8, device=dev)
    ctx.verify_mixed_dev(2 * npairs, d_tags.data_ptr(), d_pk.data_ptr(), d_pko.data_ptr(), d_msg.data_ptr(), d_mso.data_ptr(),
                         d_sig.data_ptr(), d_sgo.data_ptr(), acc.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = acc.cpu().numpy()
    del d_pk, d_sig, d_msg, acc
    torch.cuda.empty_cache()
    # class-dictated accept bytes: every Falcon-512 item is valid; a Falcon-1024 item is accepted iff its kind is a valid one
    want10 = np.isin(p10[3], VALID_KINDS).astype(np.uint8)
    triple = src % base                       # pair p uses triple p % base of each class
    want = np.where(tags == 9, np.uint8(1), want10[triple])
    np.testing.assert_array_equal(got, want)
    assert (tags == 9).sum() == (tags == 10).sum() == npairs
    # oracle sample: items of both classes from the last tile
    idx = np.arange(2 * npairs - 8192, 2 * npairs)
    i9, i10 = idx[tags[idx] == 9], idx[tags[idx] == 10]
    o9 = oracle.verify_many(9, p9[0][triple[i9]], p9[1][triple[i9]], p9[2][triple[i9]], threads=16)[0] & 1
    o10 = oracle.verify_many(10, p10[0][triple[i10]], p10[1][triple[i10]], p10[2][triple[i10]], threads=16)[0] & 1
    np.testing.assert_array_equal(got[i9], o9)
    np.testing.assert_array_equal(got[i10], o10)
