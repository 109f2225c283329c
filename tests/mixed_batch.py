"""Mixed Falcon-512 + Falcon-1024 batches (BASELINE.json configs[4]) for tests and bench.py.

Items come in pairs of one Falcon-512 and one Falcon-1024 triple; a seeded random bit per pair decides which of the
two comes first, so the batch is exactly 50/50 by N with an irregular tag sequence, and record j of each class keeps
its identity (expected accept bytes can be mapped back). Records are packed back to back: a pair's keys take
897 + 1793 bytes, its signatures 666 + 1280, its messages 2 * mlen."""
import numpy as np

PK = {9: 897, 10: 1793}
SIG = {9: 666, 10: 1280}


def pair_order_bits(npairs, seed):
    return np.random.default_rng(seed).integers(0, 2, npairs, dtype=np.uint8)


def layout(bits, mlen):
    """Tags and byte offsets of the 2 * len(bits) items. Returns logn (u8), pk_off, sig_off (u64, per item),
    msg_off (u64, count + 1), and src (index of the item's triple within its class = its pair number)."""
    npairs = bits.shape[0]
    first10 = bits.astype(bool)  # pair starts with the Falcon-1024 item
    logn = np.empty(2 * npairs, np.uint8)
    logn[0::2] = np.where(first10, 10, 9)
    logn[1::2] = np.where(first10, 9, 10)
    pair = np.arange(npairs, dtype=np.uint64)
    pk_off = np.empty(2 * npairs, np.uint64)
    sig_off = np.empty(2 * npairs, np.uint64)
    pk_off[0::2] = pair * (PK[9] + PK[10])
    pk_off[1::2] = pk_off[0::2] + np.where(first10, PK[10], PK[9]).astype(np.uint64)
    sig_off[0::2] = pair * (SIG[9] + SIG[10])
    sig_off[1::2] = sig_off[0::2] + np.where(first10, SIG[10], SIG[9]).astype(np.uint64)
    msg_off = np.arange(2 * npairs + 1, dtype=np.uint64) * np.uint64(mlen)
    src = np.repeat(np.arange(npairs), 2)
    return logn, pk_off, sig_off, msg_off, src


def pack_numpy(bits, a9, a10):
    """Rows of a9 (npairs, l9) and a10 (npairs, l10) packed pair by pair in the order `bits` says -> flat u8."""
    first10 = bits.astype(bool)[:, None]
    l9, l10 = a9.shape[1], a10.shape[1]
    out = np.empty((bits.shape[0], l9 + l10), np.uint8)
    ab = np.concatenate([a9, a10], axis=1)
    ba = np.concatenate([a10, a9], axis=1)
    np.copyto(out, np.where(first10, ba, ab))
    return out.reshape(-1)


def pack_torch(bits_t, a9, a10, tile=1, chunk=1 << 16):
    """The same on the GPU (torch uint8 tensors), with the class corpora repeated `tile` times: pair p uses triple
    p % npairs of each class. Built chunk by chunk so that the temporaries stay small."""
    import torch
    npairs = a9.shape[0]
    total = npairs * tile
    l9, l10 = a9.shape[1], a10.shape[1]
    out = torch.empty((total, l9 + l10), dtype=torch.uint8, device=a9.device)
    for t in range(tile):
        for c0 in range(0, npairs, chunk):
            c1 = min(c0 + chunk, npairs)
            f10 = bits_t[t * npairs + c0:t * npairs + c1].bool()[:, None]
            ab = torch.cat([a9[c0:c1], a10[c0:c1]], dim=1)
            ba = torch.cat([a10[c0:c1], a9[c0:c1]], dim=1)
            out[t * npairs + c0:t * npairs + c1] = torch.where(f10, ba, ab)
    return out.reshape(-1)
