#!/usr/bin/env python3
"""Large differential run on the GPU box: HIP path (through the C ABI) against the CPU oracle, item by item, on seeded
corpora with half of the items malformed (all 12 classes of SURVEY.md 8d), both norm policies; the debug outputs (status,
exact and wrapped norm) are compared on every item as well. Usage: python tools/fuzz_big.py [items_512] [items_1024] [seed]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
torch.cuda.init()
import support
from falcon_amd import capi

n9 = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
n10 = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 21
seed = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0xF0323
ctx = capi.Context(0)
orc = support.oracle()
out = {}
for logn, total in ((9, n9), (10, n10)):
    bad = {"accept_exact": 0, "accept_wrap32": 0, "status": 0, "norm_exact": 0, "norm_wrap32": 0, "accept_async_dev": 0,
           "accept_noalloc": 0, "accept_prepared": 0, "accept_small_calls": 0, "status_small_calls": 0, "norm_small_calls": 0,
           "accept_keyset": 0, "accept_keyset_indexed": 0, "ntt_elements": 0, "intt_elements": 0, "polymul_elements": 0}
    done = 0
    classes = {}
    t0 = time.time()
    step = 1 << 19
    for first in range(0, total, step):
        n = min(step, total - first)
        pk, msg, sig, kinds = support.corpus().generate(logn, n, seed=seed + logn, first=first, mlen=32, malformed_permille=500, threads=16)
        flags, odbg = orc.verify_many(logn, pk, msg, sig, debug=True, threads=16)
        acc, gdbg = ctx.verify_many(logn, pk, msg, sig, debug=True)
        accw = ctx.verify_many(logn, pk, msg, sig, norm_policy=1)
        bad["accept_exact"] += int((acc != (flags & 1)).sum())
        bad["accept_wrap32"] += int((accw != ((flags >> 1) & 1)).sum())
        bad["status"] += int((gdbg["status"] != odbg["status"]).sum())
        bad["norm_exact"] += int((gdbg["norm_exact"] != odbg["norm_exact"]).sum())
        bad["norm_wrap32"] += int((gdbg["norm_wrap32"] != odbg["norm_wrap32"]).sum())
        # the same items device-resident: asynchronous call (device-side retry pass), the allocation-free entry point on
        # 4096-item slices, and the prepared-key path
        dev = torch.device("cuda", 0)
        st = torch.cuda.current_stream().cuda_stream
        t = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (pk, msg, sig)]
        d_acc = torch.full((n,), 7, dtype=torch.uint8, device=dev)
        ctx.set_option(capi.OPT_ASYNC, 1)
        ctx.verify_many_dev(logn, n, t[0].data_ptr(), t[1].data_ptr(), None, 32, t[2].data_ptr(), d_acc.data_ptr(), stream=st)
        assert ctx.finish(st) == 0
        ctx.set_option(capi.OPT_ASYNC, 0)
        bad["accept_async_dev"] += int((d_acc.cpu().numpy() != (flags & 1)).sum())
        d_acc.fill_(7)
        pkl, sgl = pk.shape[1], sig.shape[1]
        for c0 in range(0, min(n, 1 << 16), 4096):
            m = min(4096, n - c0)
            ctx.verify_many_dev_noalloc(logn, m, t[0].data_ptr() + c0 * pkl, t[1].data_ptr() + c0 * 32, None, 32,
                                        t[2].data_ptr() + c0 * sgl, d_acc.data_ptr() + c0, stream=st)
        m = min(n, 1 << 16)
        bad["accept_noalloc"] += int((d_acc[:m].cpu().numpy() != (flags[:m] & 1)).sum())
        d_prep = torch.empty((n, 1 << logn), dtype=torch.int16, device=dev)
        ctx.prepare_keys_dev(logn, n, t[0].data_ptr(), None, d_prep.data_ptr(), stream=st)
        d_acc.fill_(7)
        ctx.verify_many_prepared_dev(logn, n, d_prep.data_ptr(), None, t[1].data_ptr(), None, 32, t[2].data_ptr(), d_acc.data_ptr(), stream=st)
        torch.cuda.synchronize()
        bad["accept_prepared"] += int((d_acc.cpu().numpy() != (flags & 1)).sum())
        del t, d_acc, d_prep
        # round 3: the same items against their keys resident on the GPU (fa_keyset), host buffers; then 61 keys shared by
        # all items through an index (most then fail the norm: checked against the oracle with the keys permuted likewise)
        ks = ctx.keyset_create(logn, pk=pk)
        bad["accept_keyset"] += int((ctx.verify_many_keyset(ks, msg, sig) != (flags & 1)).sum())
        ks.close()
        m = min(n, 1 << 17)
        kidx = (np.arange(m) % 61).astype(np.uint32)
        ks = ctx.keyset_create(logn, pk=pk[:61])
        want_idx, _ = orc.verify_many(logn, pk[kidx], msg[:m], sig[:m], threads=16)
        bad["accept_keyset_indexed"] += int((ctx.verify_many_keyset(ks, msg[:m], sig[:m], key_idx=kidx) != (want_idx & 1)).sum())
        ks.close()
        # the transforms as operators (ntt.hpp:120-200) and the product built from them, element for element, on the decoded
        # keys of the first 4096 items (raw 14-bit fields, values >= q included)
        h = np.stack([orc.decode_pkey(logn, k)[1] for k in pk[:4096]])
        fwd = ctx.ntt_many(logn, h)
        want_fwd = np.stack([orc.ntt(r % 12289, logn) for r in h])
        bad["ntt_elements"] += int((fwd != want_fwd).sum())
        bad["intt_elements"] += int((ctx.ntt_many(logn, fwd, inverse=True) != h % 12289).sum())
        prod = ctx.polymul_many(logn, h[:2048], h[2048:])
        want_prod = np.stack([orc.polymul(a, b, logn) for a, b in zip(h[:256], h[2048:2304])])
        bad["polymul_elements"] += int((prod[:256] != want_prod).sum())
        # the first 2^15 items again in calls of a few items to a few thousand: the one-wave-per-item front (<= 512), its
        # copy-free host path, the lane-pair sponge (<= 2^14)
        sizes, c0, k = (1, 2, 3, 5, 17, 64, 65, 100, 333, 512, 513, 1000, 4096), 0, 0
        while c0 < min(n, 1 << 15):
            m = min(sizes[k % len(sizes)], n - c0)
            sl = slice(c0, c0 + m)
            a1 = ctx.verify_many(logn, pk[sl], msg[sl], sig[sl])
            a2, g2 = ctx.verify_many(logn, pk[sl], msg[sl], sig[sl], debug=True)
            bad["accept_small_calls"] += int((a1 != (flags[sl] & 1)).sum()) + int((a2 != (flags[sl] & 1)).sum())
            bad["status_small_calls"] += int((g2["status"] != odbg["status"][sl]).sum())
            bad["norm_small_calls"] += int((g2["norm_exact"] != odbg["norm_exact"][sl]).sum())
            c0 += m
            k += 1
        for k, c in zip(*np.unique(kinds, return_counts=True)):
            classes[int(k)] = classes.get(int(k), 0) + int(c)
        done += n
        print(f"Falcon-{1 << logn}: {done}/{total} items, mismatches so far {bad}", flush=True)
    out[f"falcon{1 << logn}"] = {"items": done, "malformed_share": 1 - classes.get(0, 0) / done, "items_per_class": classes,
                                 "mismatches": bad, "seconds": round(time.time() - t0, 1)}

# ---- ragged messages and mixed batches (the GENERAL kernel instantiations: records through offset arrays) ----
# Corpora of several message lengths around the sponge's block boundaries (0, 1, 95/96/97 = one absorb block with the
# salt exactly full, 135..137, 300 bytes), their records shuffled into one flat batch; then a Falcon-512 / Falcon-1024
# mix of the same kind through fa_verify_mixed. The oracle runs per class on the same bytes.
t0 = time.time()
rng = np.random.default_rng(seed)
parts, nrag = [], 1 << 15
for logn in (9, 10):
    for j, ml in enumerate((0, 1, 31, 95, 96, 97, 135, 136, 137, 300)):
        pk, msg, sig, _ = support.corpus().generate(logn, nrag, seed=seed + 977 * logn + j, mlen=ml, malformed_permille=400, threads=16)
        flags, _ = orc.verify_many(logn, pk, msg, sig, threads=16)
        parts.append((logn, ml, pk, msg.reshape(nrag, ml), sig, flags))
bad = {"accept_ragged_512": 0, "accept_ragged_1024": 0, "accept_ragged_keyset": 0, "accept_mixed": 0, "accept_mixed_wrap32": 0}
total_rag = 0
for logn in (9, 10):
    mine = [p for p in parts if p[0] == logn]
    order = rng.permutation(len(mine) * nrag)
    pk = np.concatenate([p[2] for p in mine])[order]
    sig = np.concatenate([p[4] for p in mine])[order]
    want = np.concatenate([p[5] for p in mine])[order]
    lens = np.concatenate([np.full(nrag, p[1], np.uint64) for p in mine])[order]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    flat = np.zeros(int(off[-1]), np.uint8)
    src = [p[3] for p in mine]
    for i, o in enumerate(order):   # scatter the messages into the flat buffer in batch order
        ml = int(lens[i])
        if ml:
            flat[int(off[i]):int(off[i]) + ml] = src[o // nrag][o % nrag]
    got = ctx.verify_many(logn, pk, flat, sig, msg_off=off)
    bad[f"accept_ragged_{1 << logn}"] += int((got != (want & 1)).sum())
    if logn == 9:
        ks = ctx.keyset_create(9, pk=pk)
        bad["accept_ragged_keyset"] += int((ctx.verify_many_keyset(ks, flat, sig, msg_off=off) != (want & 1)).sum())
        ks.close()
    total_rag += len(order)
# mixed: every record of both classes in one shuffled batch, offsets into three flat arrays
recs = [(p[0], p[2][i], p[3][i], p[4][i], p[5][i]) for p in parts for i in range(0, nrag, 8)]
order = rng.permutation(len(recs))
lg = np.array([recs[o][0] for o in order], np.uint8)
pk_off = np.concatenate([[0], np.cumsum([recs[o][1].size for o in order])]).astype(np.uint64)
sig_off = np.concatenate([[0], np.cumsum([recs[o][3].size for o in order])]).astype(np.uint64)
msg_off = np.concatenate([[0], np.cumsum([recs[o][2].size for o in order])]).astype(np.uint64)
pkf = np.concatenate([recs[o][1] for o in order])
sigf = np.concatenate([recs[o][3] for o in order])
msgf = np.concatenate([recs[o][2] for o in order]) if msg_off[-1] else np.zeros(0, np.uint8)
want = np.array([recs[o][4] for o in order], np.uint8)
bad["accept_mixed"] += int((ctx.verify_mixed(lg, pkf, pk_off[:-1], msgf, msg_off, sigf, sig_off[:-1]) != (want & 1)).sum())
bad["accept_mixed_wrap32"] += int((ctx.verify_mixed(lg, pkf, pk_off[:-1], msgf, msg_off, sigf, sig_off[:-1], norm_policy=1) != ((want >> 1) & 1)).sum())
out["ragged_and_mixed"] = {"items": total_rag + len(order), "ragged_items": total_rag, "mixed_items": int(len(order)),
                           "message_lengths": [0, 1, 31, 95, 96, 97, 135, 136, 137, 300], "mismatches": bad,
                           "seconds": round(time.time() - t0, 1)}
print(f"ragged and mixed: {out['ragged_and_mixed']}", flush=True)
out["seed"] = seed
print(json.dumps(out))
sys.exit(1 if any(v for r in out.values() if isinstance(r, dict) for v in r["mismatches"].values()) else 0)
