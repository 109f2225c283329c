"""The N > 1 path on CPU: two gloo ranks shard a batch by item, verify their ranges independently
and gather the accept bytes on rank 0. No GPU here, so the per-shard
verifier is the CPU oracle standing in for Context.verify_many -- the sharding, the ragged last
range and the gather are what is under test."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, count, out_path):
    sys.path.insert(0, HERE)
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    import support
    from falcon_amd import shard

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pk, msg, sig, kinds = support.corpus().generate(9, count, seed=321, malformed_permille=400, threads=2)
        calls = []

        def verify_fn(logn, p, m, s):
            calls.append(p.shape[0])
            return support.oracle().verify_many(logn, p, m, s)[0] & 1

        acc = shard.verify_many_sharded(verify_fn, 9, pk, msg, sig)
        lo, hi = shard.shard_range(count, world, rank)
        assert calls == ([hi - lo] if hi > lo else [])
        if rank == 0:
            np.save(out_path, acc)
        else:
            assert acc is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("count", [101, 2])
def test_two_rank_sharded_verify(tmp_path, count, oracle, corpus):
    out = str(tmp_path / "acc.npy")
    mp.spawn(_rank_main, args=(2, _free_port(), count, out), nprocs=2, join=True)
    got = np.load(out)
    pk, msg, sig, kinds = corpus.generate(9, count, seed=321, malformed_permille=400, threads=2)
    want = oracle.verify_many(9, pk, msg, sig)[0] & 1
    np.testing.assert_array_equal(got, want)


def test_shard_ranges_cover_batch():
    from falcon_amd import shard
    for count in (0, 1, 7, 8, 9, 1 << 20, (1 << 26) + 3):
        for world in (1, 2, 3, 4, 8):
            r = [shard.shard_range(count, world, k) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == count
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(hi - lo for lo, hi in r) - min(hi - lo for lo, hi in r) <= -(-count // world)
    with pytest.raises(ValueError):
        shard.shard_range(10, 2, 2)


def test_split_mixed():
    from falcon_amd import shard
    i9, i10 = shard.split_mixed([9, 10, 10, 9, 9])
    assert i9.tolist() == [0, 3, 4] and i10.tolist() == [1, 2]
    with pytest.raises(ValueError):
        shard.split_mixed([9, 11])
