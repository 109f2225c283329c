"""GPU: fa_verify_many_dev_noalloc (the allocation-free entry point on the library's static scratch array)
against fa_verify_many_dev, two calls in a row, the second with a smaller input; two host threads with a context
each (all contexts of a GPU share the array: the library's per-GPU lock keeps such calls apart)."""
import threading

import numpy as np
import pytest

import support

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("logn", [9, 10])
def test_noalloc_matches_verify_many_dev(ctx, oracle, corpus, logn):
    import torch
    from falcon_amd import capi
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    for count, seed in ((capi.NOALLOC_MAX_ITEMS, 11), (1000, 12), (1, 13)):  # the second and third calls are smaller
        pk, msg, sig, kinds = corpus.generate(logn, count, seed=seed + logn, malformed_permille=300, threads=16)
        t = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (pk, msg, sig)]
        ref = torch.full((count,), 7, dtype=torch.uint8, device=dev)
        got = torch.full((count,), 9, dtype=torch.uint8, device=dev)
        ctx.verify_many_dev(logn, count, t[0].data_ptr(), t[1].data_ptr(), None, 32, t[2].data_ptr(), ref.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        # free device memory as the runtime itself reports it (hipMemGetInfo): torch's own counter only sees torch's caching
        # allocator, a hipMalloc inside the library would pass it unnoticed
        before, before_torch = torch.cuda.mem_get_info()[0], torch.cuda.memory_allocated()
        ctx.verify_many_dev_noalloc(logn, count, t[0].data_ptr(), t[1].data_ptr(), None, 32, t[2].data_ptr(), got.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == before and torch.cuda.memory_allocated() == before_torch
        np.testing.assert_array_equal(got.cpu().numpy(), ref.cpu().numpy())
        want = oracle.verify_many(logn, pk, msg, sig, threads=16)[0] & 1
        np.testing.assert_array_equal(got.cpu().numpy(), want)
        assert (kinds != 0).any() or count == 1


def test_noalloc_limits_and_retry(oracle, corpus):
    import torch
    from falcon_amd import capi
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    c = capi.Context(0)
    try:
        n = capi.NOALLOC_MAX_ITEMS + 1
        pk, msg, sig, _ = corpus.generate(9, n, seed=5, threads=16)
        t = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (pk, msg, sig)]
        acc = torch.zeros(n, dtype=torch.uint8, device=dev)
        with pytest.raises(capi.FalconAmdError) as e:
            c.verify_many_dev_noalloc(9, n, t[0].data_ptr(), t[1].data_ptr(), None, 32, t[2].data_ptr(), acc.data_ptr(), stream=stream)
        assert e.value.code == capi.E_TOO_MANY
        # too few blocks up front: about half of 3000 items go through the device-side retry pass (capacity 2048)
        c.set_option(capi.OPT_HASH_BLOCKS_512, 8)
        m = 3000
        want = oracle.verify_many(9, pk[:m], msg[:m], sig[:m], threads=16)[0] & 1
        acc.fill_(7)
        c.verify_many_dev_noalloc(9, m, t[0].data_ptr(), t[1].data_ptr(), None, 32, t[2].data_ptr(), acc.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(acc[:m].cpu().numpy(), want)
        # more queued items than the pass takes: reported, not guessed
        with pytest.raises(capi.FalconAmdError) as e:
            c.verify_many_dev_noalloc(9, 4096, t[0].data_ptr(), t[1].data_ptr(), None, 32, t[2].data_ptr(), acc.data_ptr(), stream=stream)
        assert e.value.code == capi.E_UNRESOLVED
        c.set_option(capi.OPT_WIRE_FORMAT, capi.WIRE_SPEC)
        with pytest.raises(capi.FalconAmdError):
            c.verify_many_dev_noalloc(9, 10, t[0].data_ptr(), t[1].data_ptr(), None, 32, t[2].data_ptr(), acc.data_ptr(), stream=stream)
    finally:
        c.close()


def test_noalloc_two_threads_two_contexts(oracle, corpus):
    """Every context of a GPU carves its workspace out of the same static array. Two host threads, each with its own
    context and stream, call the allocation-free entry point at the same time on different inputs: without the
    library's per-GPU lock their s2 tiles, hash blocks and counters would overwrite each other."""
    import torch
    from falcon_amd import capi
    dev = torch.device("cuda", 0)
    inputs, wants = [], []
    for t, logn in enumerate((9, 10)):
        pk, msg, sig, _ = corpus.generate(logn, capi.NOALLOC_MAX_ITEMS, seed=40 + t, malformed_permille=300, threads=16)
        inputs.append([torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (pk, msg, sig)])
        wants.append(oracle.verify_many(logn, pk, msg, sig, threads=16)[0] & 1)
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    accs = [torch.zeros(capi.NOALLOC_MAX_ITEMS, dtype=torch.uint8, device=dev) for _ in range(2)]
    torch.cuda.synchronize()
    bad = []
    go = threading.Barrier(2)

    def work(t):
        logn = 9 + t
        c = capi.Context(0)
        try:
            go.wait()
            for rep in range(12):
                n = capi.NOALLOC_MAX_ITEMS - 57 * rep
                accs[t].fill_(7)
                torch.cuda.synchronize()
                c.verify_many_dev_noalloc(logn, n, inputs[t][0].data_ptr(), inputs[t][1].data_ptr(), None, 32,
                                          inputs[t][2].data_ptr(), accs[t].data_ptr(), stream=streams[t].cuda_stream)
                if not np.array_equal(accs[t][:n].cpu().numpy(), wants[t][:n]):
                    bad.append((t, rep))
        except Exception as e:  # noqa: BLE001
            bad.append(repr(e))
        finally:
            c.close()
    th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not bad, bad
