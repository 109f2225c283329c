#!/usr/bin/env python3
"""bench.py -- Falcon-512 batched verify throughput on MI355X (BASELINE.json config 3).

A step = one pass of the verify hot path (fa_verify_many_dev: decode -> SHAKE256 -> NTT/norm
kernels) over one batch of 2^20 distinct (public key, 32-byte message, signature) triples that
are already resident in HBM. With --gpus N > 1 every rank verifies its own 2^20-item shard
(independent items, no collective on the data path: "weak" scaling); rank 0 prints ONE JSON line.

Extra legs on rank 0 at N == 1 (outside the timed region):
  * cpu_baseline: the CPU oracle ("port" of the reference's falcon::verify<N>) over a bounded
    sample of the same corpus with pthreads, plus an item-by-item comparison of the accept
    flags with the GPU's;
  * extras: Falcon-1024 with the malformed mix (config 4) and the n=512 batched polymul
    (config 2) at reduced batch sizes, one timed pass each.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# Algorithmic work per unit, SURVEY.md section 8(d) (reference op-cost model, int32 lane-ops):
#   Keccak-f[1600] = 7,440; butterfly = 11; pointwise modmul = 5; rejection sampling = 4 per draw;
#   pk unpack = 3 and norm = 3 per coefficient; signature decode = 10 per coefficient.
OPS_512 = {
    "hash_kernel": 8.593 * 7440,
    "verify_kernel": 9216 * 11 + 1024 * 5 + 546 * 4 + 512 * 3 + 512 * 3,
    "decode_kernel": 512 * 10,
}
BYTES_512 = 897 + 666 + 32 + 1   # compulsory HBM bytes per verify
PEAK_VALU_TIOPS = 256 * 4 * 32 * 2.4e9 / 1e12   # 256 CU x 4 SIMD-32 x 2.4 GHz = 78.6 T lane-ops/s
PEAK_HBM_GBS = 8000.0


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--items", type=int, default=1 << 20, help="items per GPU per step")
    ap.add_argument("--chunk", type=int, default=0, help="override FA_OPT_CHUNK_ITEMS")
    ap.add_argument("--cpu-sample", type=int, default=1 << 18)
    ap.add_argument("--logn", type=int, default=9, choices=(9, 10),
                    help="9 = Falcon-512 (the headline workload, BASELINE.json configs[2]); 10 = Falcon-1024 (A/B runs only)")
    ap.add_argument("--malformed", type=int, default=0, help="permille of malformed items in the timed corpus (A/B runs only)")
    ap.add_argument("--overlap", type=int, default=-1, help="override FA_OPT_OVERLAP (A/B runs only)")
    ap.add_argument("--grid-per-cu", type=int, default=-1, help="override FA_OPT_GRID_PER_CU (A/B runs only)")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="process group for the barrier and the max-over-ranks timing; gloo + --same-gpu rehearses the "
                         "N > 1 path on a one-GPU box")
    ap.add_argument("--same-gpu", action="store_true", help="every rank uses cuda:0 (rehearsal only)")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE")
    if args.same_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red_dev = dev if args.backend == "nccl" else torch.device("cpu")  # where the two scalar reductions live
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    import falcon_amd
    from falcon_amd import capi
    import support

    threads = max(1, min(os.cpu_count() or 1, 16))
    ctx = capi.Context(local_rank)
    if args.chunk:
        ctx.set_option(capi.OPT_CHUNK_ITEMS, args.chunk)
    if args.overlap >= 0:
        ctx.set_option(capi.OPT_OVERLAP, args.overlap)
    if args.grid_per_cu >= 0:
        ctx.set_option(capi.OPT_GRID_PER_CU, args.grid_per_cu)

    # ---- corpus: this rank's shard of the seeded stream (counter-based, corpus/falcon_corpus.c) ----
    n_items = args.items
    t0 = time.time()
    logn = args.logn
    pk, msg, sig, _ = support.corpus().generate(logn, n_items, seed=0xFA1C0512, first=rank * n_items, mlen=32,
                                                malformed_permille=args.malformed, threads=threads)
    log(f"[rank {rank}] corpus: {n_items} Falcon-{1 << logn} triples in {time.time() - t0:.1f}s ({threads} threads)")
    d_pk = torch.from_numpy(pk).to(dev)
    d_msg = torch.from_numpy(np.ascontiguousarray(msg)).to(dev)
    d_sig = torch.from_numpy(sig).to(dev)
    d_acc = torch.zeros(n_items, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        ctx.verify_many_dev(logn, n_items, d_pk.data_ptr(), d_msg.data_ptr(), None, 32, d_sig.data_ptr(),
                            d_acc.data_ptr(), stream=stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ctx.set_option(capi.OPT_PROFILE, 1)   # HIP events around every kernel, on the launch stream
    ctx.get_profile()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    prof = ctx.get_profile()
    ctx.set_option(capi.OPT_PROFILE, 0)
    accepted = int(d_acc.sum().item())
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        a = torch.tensor([accepted], dtype=torch.int64, device=red_dev)
        dist.all_reduce(a, op=dist.ReduceOp.SUM)
        accepted = int(a.item())

    total_items = n_items * world * args.steps
    value = total_items / elapsed
    result = {
        "metric": f"Falcon-{1 << logn} verifies/sec", "value": value, "unit": "verifies/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32 (mod q = 12289), u64 Keccak lanes",
        "data": "synthetic",
        "config": {"workload": f"Falcon-{1 << logn} batched verify, {n_items} distinct keys/signatures per GPU, 32-byte messages, "
                               f"device-resident (BASELINE.json configs[{logn - 7}])" + (f", {args.malformed} permille malformed" if args.malformed else ""),
                   "items_per_gpu": n_items, "parallelism": f"items sharded over {world} GPU(s), no collective"},
        "accepted_fraction": accepted / (n_items * world),
    }

    if rank == 0:
        # ---- roofline of the dominant kernel (VALU-bound integer work; HBM shown beside it) ----
        kern_ms = {"hash_kernel": prof["ms_hash"], "verify_kernel": prof["ms_verify"], "decode_kernel": prof["ms_decode"]}
        kern_n = {"hash_kernel": prof["launches_hash"], "verify_kernel": prof["launches_verify"],
                  "decode_kernel": prof["launches_decode"]}
        dom = max(kern_ms, key=kern_ms.get)
        items_per_launch = n_items * args.steps / max(kern_n[dom], 1)
        avg_ms = kern_ms[dom] / max(kern_n[dom], 1)
        achieved = OPS_512[dom] * items_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        traffic, issued = None, None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                counters = json.load(open(tpath))
                traffic = counters.get(dom)
                issued = counters.get("valu_instructions_per_launch", {}).get(dom)
            except Exception:
                traffic, issued = None, None
        result["roofline"] = {
            "bound": "valu", "kernel": dom, "achieved": achieved, "peak": PEAK_VALU_TIOPS, "unit": "Tiop/s",
            "frac": achieved / PEAK_VALU_TIOPS, "traffic": traffic,
            "avg_launch_ms": avg_ms, "items_per_launch": items_per_launch, "algorithmic_ops_per_item": OPS_512[dom],
            "note": "integer VALU work, no MFMA; peak = 256 CU x 4 SIMD-32 x 2.4 GHz lane-ops/s (SURVEY.md 8d). achieved counts the "
                    "reference's operations (11 per butterfly, 4 transforms); the kernel issues 5 per butterfly and runs 3 "
                    "transforms, so frac can pass 1 -- issue_utilisation is the share of the vector issue slots in use",
        }
        if issued and avg_ms > 0 and logn == 9 and n_items == 1 << 20 and not args.malformed:
            # wave instructions of this kernel per launch (PMC, profiles/) x 4 cycles per issue slot / (SIMDs x clock x time)
            result["roofline"]["issue_utilisation"] = issued * 4 / (1024 * 2.4e9 * avg_ms * 1e-3)
            result["roofline"]["issued_valu_instructions"] = issued
        result["kernels_ms_per_step"] = {k: v / args.steps for k, v in kern_ms.items()}
        per_gpu_rate = n_items * args.steps / elapsed
        # the same launch against the HBM roofline, in the contract's form: SURVEY.md 8(d)'s compulsory bytes per
        # verify x the items one launch processes / that kernel's average launch time; traffic = measured HBM bytes
        # of that launch (PMC, profiles/hbm_traffic.json). Far from the bound: the path is VALU-issue limited.
        hbm_achieved = BYTES_512 * items_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        result["roofline_hbm"] = {"bound": "hbm", "kernel": dom, "achieved": hbm_achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                  "frac": hbm_achieved / PEAK_HBM_GBS, "traffic": traffic,
                                  "algorithmic_bytes_per_item": BYTES_512,
                                  "whole_path_GBps": per_gpu_rate * BYTES_512 / 1e9}
        result["whole_path_valu_frac"] = per_gpu_rate * sum(OPS_512.values()) / 1e12 / PEAK_VALU_TIOPS
        result["retried_items"] = prof["retried_items"]

    if rank == 0 and world == 1 and not args.no_cpu:
        # ---- CPU baseline: the oracle over a bounded sample of the same corpus ----
        ns = min(args.cpu_sample, n_items)
        orc = support.oracle()
        t0 = time.perf_counter()
        flags, _ = orc.verify_many(logn, pk[:ns], msg[:ns], sig[:ns], threads=threads)
        dt = time.perf_counter() - t0
        t1 = time.perf_counter()
        n1 = min(8192, ns)
        orc.verify_many(logn, pk[:n1], msg[:n1], sig[:n1], threads=1)
        dt1 = time.perf_counter() - t1
        # every accept flag of the timed corpus against the CPU result: the rest of the items, untimed
        if ns < n_items:
            rest, _ = orc.verify_many(logn, pk[ns:], msg[ns:], sig[ns:], threads=threads)
            flags = np.concatenate([flags, rest])
        gpu_flags = d_acc.cpu().numpy()
        result["cpu_baseline"] = {
            "value": ns / dt, "unit": "verifies/s", "cores": threads, "kind": "port",
            "sample": f"first {ns} items of the same corpus, one verify per item, {threads} pthreads "
                      f"(~{dt * threads:.0f} core-seconds; oracle/falcon_oracle.c, gcc -O3 -std=c11); "
                      f"single thread: {1e6 * dt1 / n1:.1f} us/verify",
            "single_thread_us": 1e6 * dt1 / n1,
            "accept_flags_compared": int(flags.size),
            "accept_mismatches_vs_gpu": int((gpu_flags != (flags & 1)).sum()),
        }

    if rank == 0 and world == 1 and not args.no_extras:
        extras = {}
        # Falcon-1024 with the malformed mix (BASELINE.json configs[3]) at its full 2^20 items
        ne = min(n_items, 1 << 20)
        pk2, msg2, sig2, kinds = support.corpus().generate(10, ne, seed=0xFA1C1024, mlen=32, malformed_permille=250,
                                                           threads=threads)
        e_pk, e_msg, e_sig = (torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (pk2, msg2, sig2))
        e_acc = torch.zeros(ne, dtype=torch.uint8, device=dev)

        def step1024():
            ctx.verify_many_dev(10, ne, e_pk.data_ptr(), e_msg.data_ptr(), None, 32, e_sig.data_ptr(),
                                e_acc.data_ptr(), stream=stream)
        step1024()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            step1024()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        nchk = 1 << 14
        oflags, _ = support.oracle().verify_many(10, pk2[:nchk], msg2[:nchk], sig2[:nchk], threads=threads)
        extras["falcon1024_malformed_mix"] = {
            "verifies_per_s": ne / dt, "items": ne, "malformed_share": float((kinds != 0).mean()),
            "accepted_fraction": float(e_acc.float().mean().item()),
            "accept_mismatches_vs_oracle_first_16384": int((e_acc[:nchk].cpu().numpy() != (oflags & 1)).sum())}
        # mixed Falcon-512 + Falcon-1024 batch (BASELINE.json configs[4], one GPU's share): the batch is split by N
        # and runs as one pass per class; `ne` items of each class here
        if logn == 9:
            def step_mixed():
                ctx.verify_many_dev(9, ne, d_pk.data_ptr(), d_msg.data_ptr(), None, 32, d_sig.data_ptr(),
                                    d_acc.data_ptr(), stream=stream)
                step1024()
            step_mixed()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                step_mixed()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            extras["mixed_512_1024"] = {"verifies_per_s": 2 * ne / dt, "items": 2 * ne, "share_1024": 0.5}
        del e_pk, e_msg, e_sig, e_acc
        # host-buffer entry point (fa_verify_many: H2D staging + kernels + D2H), pageable numpy memory: the
        # PCIe-inclusive rate that DESIGN.md section 7 quotes; never the headline value
        nh = min(n_items, 1 << 19)
        ctx.verify_many(logn, pk[:nh], msg[:nh], sig[:nh])
        t0 = time.perf_counter()
        h_acc = ctx.verify_many(logn, pk[:nh], msg[:nh], sig[:nh])
        dt = time.perf_counter() - t0
        extras["host_buffers_pcie_inclusive"] = {"verifies_per_s": nh / dt, "items": nh,
                                                 "input_GBps": nh * (pk.shape[1] + sig.shape[1] + 32) / dt / 1e9,
                                                 "accepted_fraction": float(h_acc.mean())}
        # the same from pinned caller memory (fa_host_alloc): the copies are asynchronous and overlap the kernels
        pin = [capi.PinnedArray(a[:nh].shape) for a in (pk, msg, sig)] + [capi.PinnedArray((nh,))]
        for b, a in zip(pin, (pk, msg, sig)):
            b.array[...] = a[:nh]
        ctx.verify_many(logn, pin[0].array, pin[1].array, pin[2].array, accept_out=pin[3].array)
        t0 = time.perf_counter()
        ctx.verify_many(logn, pin[0].array, pin[1].array, pin[2].array, accept_out=pin[3].array)
        dt = time.perf_counter() - t0
        extras["host_buffers_pinned_pcie_inclusive"] = {"verifies_per_s": nh / dt, "items": nh,
                                                        "input_GBps": nh * (pk.shape[1] + sig.shape[1] + 32) / dt / 1e9,
                                                        "accepted_fraction": float(pin[3].array.mean())}
        for b in pin:
            b.close()
        # decoded-key fast path: keys prepared once (NTT(h) resident), same 2^20 Falcon-512 items
        if logn == 9:
            d_prep = torch.empty((n_items, 512), dtype=torch.int16, device=dev)
            t0 = time.perf_counter()
            ctx.prepare_keys_dev(9, n_items, d_pk.data_ptr(), None, d_prep.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            t_prep = time.perf_counter() - t0
            p_acc = torch.zeros(n_items, dtype=torch.uint8, device=dev)

            def step_prepared():
                ctx.verify_many_prepared_dev(9, n_items, d_prep.data_ptr(), None, d_msg.data_ptr(), None, 32,
                                             d_sig.data_ptr(), p_acc.data_ptr(), stream=stream)
            step_prepared()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                step_prepared()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            extras["falcon512_prepared_keys"] = {"verifies_per_s": n_items / dt, "items": n_items,
                                                 "prepare_keys_per_s": n_items / t_prep,
                                                 "accept_mismatches_vs_packed_path": int((p_acc != d_acc).sum().item())}
            del d_prep, p_acc
        # batched negacyclic polymul, n = 512 (BASELINE.json configs[1]) at 2^22 polynomials per operand
        npoly = 1 << 22
        a = torch.randint(0, 12289, (npoly, 512), dtype=torch.int32, device=dev).to(torch.int16)
        b = torch.randint(0, 12289, (npoly, 512), dtype=torch.int32, device=dev).to(torch.int16)
        c = torch.empty_like(a)
        ctx.polymul_many_dev(9, npoly, a.data_ptr(), b.data_ptr(), c.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            ctx.polymul_many_dev(9, npoly, a.data_ptr(), b.data_ptr(), c.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        extras["polymul_n512"] = {"polys_per_s": npoly / dt, "polys": npoly, "hbm_GBps": npoly * 3072 / dt / 1e9}
        result["extras"] = extras

    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
