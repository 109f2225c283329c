#!/usr/bin/env python3
"""bench.py -- batched Falcon verify throughput on MI355X (BASELINE.json configs[2] by default).

A step = one pass of the verify hot path (decode -> SHAKE256 -> NTT/norm kernels) over one batch of distinct
(public key, 32-byte message, signature) triples that are already resident in HBM:

  --workload falcon512   2^20 Falcon-512 triples per GPU, all valid            (configs[2]; the headline, default)
  --workload falcon1024  2^20 Falcon-1024 triples per GPU, 25 % malformed mix  (configs[3])
  --workload mixed       2^23 items per GPU, 50/50 Falcon-512 / Falcon-1024, tiled on the device from the two
                         corpora above and verified in place by fa_verify_mixed_dev  (configs[4]: one GPU's share of
                         the 64M-item batch; with --gpus 8 the whole of it)

With --gpus N > 1 every rank verifies its own shard (independent items, no collective on the data path: "weak"
scaling); rank 0 prints ONE JSON line.

Extra legs on rank 0 at N == 1 (outside the timed region): cpu_baseline (the CPU oracle, a port of the reference's
falcon::verify<N>, built on this machine with -O3 -march=native, over a bounded sample of the same corpus with pthreads,
plus an item-by-item comparison of the accept flags with the GPU's) and `extras` (the other BASELINE configs, the
prepared-key path against the packed path in interleaved rounds, the host-buffer entry points, resident key sets).
Before anything touches the GPU, rank 0 at N == 1 also runs ONE rocprofv3 counter pass over the same launch shape in a
child process (GRBM_GUI_ACTIVE, SQ_INSTS_VALU): the clock the chip holds under these kernels and the instructions they
issue, quoted in `roofline.pmc` as of this run (--no-pmc skips it; profiles/hbm_traffic.json is the fallback).
At N > 1 every rank additionally times BASELINE config 5's shape (mixed Falcon-512 + Falcon-1024, 2^23 items per rank)
after the headline region: `extras.mixed_512_1024_config5`.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# ---- work per unit -------------------------------------------------------------------------------------------------
# ALG_OPS: the minimum number of 32-bit vector lane-operations one item needs in each kernel, counted at the cost the
# kernels actually issue: a butterfly is 5 (3-instruction Shoup product, add, subtract), three
# transforms per verify (two forward, one inverse); a Keccak-f[1600] round on 32-bit halves is 180 (theta 80, rho/pi 48,
# chi 50, iota 2) and an item needs 8.593 permutations on average (the kernel runs 9 for every item);
# signature decoding 10 per coefficient. By construction these are <= what the kernels issue, so achieved / peak <= 1.
# REF_OPS is SURVEY.md 8(d)'s model of the REFERENCE's work (11 per butterfly, 4 transforms), reported beside it only.
# Round 3: a butterfly is 4 (two-instruction Plantard product, add, subtract); the inverse transform's first stage
# multiplies nothing (2 per butterfly); its final scaling is part of the epilogue's one reduction (4 per coefficient:
# 64-bit multiply-add, multiply-add, 64-bit multiply-add, multiply-add); pointwise 4 (64-bit product, low product, high
# product, subtract), key unpack 2, 3 per draw of the rejection sampling.
def _verify_ops(n, logn, draws):
    bf = n // 2 * logn                     # butterflies per transform
    return 2 * bf * 4 + ((bf - n // 2) * 4 + n // 2 * 2) + n * 4 + n * 2 + n * 4 + draws * 3


ALG_OPS = {
    9: {"verify_kernel": _verify_ops(512, 9, 612),
        "hash_kernel": 8.593 * 24 * 180,
        "decode_kernel": 512 * 10},
    10: {"verify_kernel": _verify_ops(1024, 10, 1224),
         "hash_kernel": 16.7 * 24 * 180,
         "decode_kernel": 1024 * 10},
}
ALG_OPS_R02_VERIFY = {9: 3 * 2304 * 5 + 512 * 5 + 512 * 2 + 512 * 6 + 612 * 3,   # the work model of round 2 (5 per butterfly), kept
                      10: 3 * 5120 * 5 + 1024 * 5 + 1024 * 2 + 1024 * 6 + 1224 * 3}   # so that the rounds' fractions can be compared
REF_OPS_512 = {"hash_kernel": 8.593 * 7440, "verify_kernel": 9216 * 11 + 1024 * 5 + 546 * 4 + 512 * 3 + 512 * 3,
               "decode_kernel": 512 * 10}
POLYMUL_OPS_512 = 2 * 2304 * 4 + (2048 * 4 + 256 * 2 + 512 * 2) + 512 * 4   # transforms (final scaling 2), pointwise (4)
BYTES = {9: 897 + 666 + 32 + 1, 10: 1793 + 1280 + 32 + 1}   # compulsory HBM bytes per verify
PEAK_VALU_TIOPS = 256 * 4 * 32 * 2.4e9 / 1e12   # 256 CU x 4 SIMD-32 x 2.4 GHz = 78.6 T lane-ops/s (full-rate instructions)
PEAK_HBM_GBS = 8000.0
PMC_SOURCE = os.path.join("profiles", "hbm_traffic.json")   # counters of the same launch shape, collected by tools/profile_all.sh


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def host_cores():
    """Host cores this process may use: the smaller of the machine's count, the affinity mask and the job's share
    (the GPU boxes export it as OMP_NUM_THREADS; os.cpu_count() there is the whole machine, many jobs' worth)."""
    n = os.cpu_count() or 1
    src = "os.cpu_count()"
    try:
        a = len(os.sched_getaffinity(0))
        if a < n:
            n, src = a, "sched_getaffinity"
    except (AttributeError, OSError):
        pass
    share = os.environ.get("OMP_NUM_THREADS", "")
    if share.isdigit() and 0 < int(share) < n:
        n, src = int(share), "the job's CPU share (OMP_NUM_THREADS)"
    return n, src


def pmc_child_pass(items, steps=3):
    """One rocprofv3 counter pass over the headline launch shape (tools/prof_run.py: `steps` device-resident calls of
    2^20 Falcon-512 items) in a child process, started before this process touches the GPU. Returns per kernel the
    clock (GRBM_GUI_ACTIVE, summed over the 8 XCDs, / 8 / the dispatch's own duration) and the issued vector
    instructions (SQ_INSTS_VALU), medians over the dispatches; None when the profiler is not there or the pass fails."""
    exe = shutil.which("rocprofv3")
    if not exe:
        return None
    out = tempfile.mkdtemp(prefix="fa_pmc_")
    try:
        t0 = time.time()
        r = subprocess.run([exe, "--pmc", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU", "--output-format", "csv", "-d", out, "-o", "pmc", "--",
                            sys.executable, os.path.join(ROOT, "tools", "prof_run.py"), "9", str(items), str(steps)],
                           cwd=ROOT, env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=240)
        if r.returncode != 0:
            log("pmc pass failed:", r.stderr[-400:])
            return None
        rows = {}
        for f in glob.glob(os.path.join(out, "**", "*_counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                name = row["Kernel_Name"]
                if "fa::" not in name:
                    continue
                short = ("decode_kernel" if "decode_fast_kernel" in name else "hash_kernel" if "hash_kernel" in name
                         else "verify_kernel" if "verify_kernel" in name else None)
                if short is None:
                    continue
                d = rows.setdefault((short, row["Dispatch_Id"]), {})
                d[row["Counter_Name"]] = d.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                d["ns"] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
        res = {}
        for (short, _), d in rows.items():
            if d.get("ns", 0) > 0 and "GRBM_GUI_ACTIVE" in d:
                e = res.setdefault(short, {"clock": [], "valu": [], "ns": []})
                e["clock"].append(d["GRBM_GUI_ACTIVE"] / 8 / d["ns"])
                e["valu"].append(d.get("SQ_INSTS_VALU", 0.0))
                e["ns"].append(d["ns"])
        med = lambda v: sorted(v)[len(v) // 2]
        return {k: {"clock_ghz": med(v["clock"]), "valu_instructions": med(v["valu"]), "dispatch_ms_under_profiler": med(v["ns"]) / 1e6,
                    "dispatches": len(v["clock"])} for k, v in res.items()} | {"_seconds": time.time() - t0}
    except (subprocess.TimeoutExpired, OSError, KeyError, ValueError) as e:
        log("pmc pass failed:", repr(e))
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="", choices=("", "falcon512", "falcon1024", "mixed"))
    ap.add_argument("--items", type=int, default=0, help="items per GPU per step (default 2^20; mixed: 2^23)")
    ap.add_argument("--chunk", type=int, default=0, help="override FA_OPT_CHUNK_ITEMS")
    ap.add_argument("--cpu-sample", type=int, default=1 << 18)
    ap.add_argument("--logn", type=int, default=9, choices=(9, 10), help="older spelling of --workload falcon512 / falcon1024")
    ap.add_argument("--malformed", type=int, default=-1, help="permille of malformed items in the timed corpus (A/B runs only)")
    ap.add_argument("--overlap", type=int, default=-1, help="override FA_OPT_OVERLAP (A/B runs only)")
    ap.add_argument("--grid-per-cu", type=int, default=-1, help="override FA_OPT_GRID_PER_CU (A/B runs only)")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="process group for the barrier and the max-over-ranks timing; gloo + --same-gpu rehearses the "
                         "N > 1 path on a one-GPU box")
    ap.add_argument("--same-gpu", action="store_true", help="every rank uses cuda:0 (rehearsal only)")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 counter pass that precedes the run at N == 1")
    ap.add_argument("--mixed-steps", type=int, default=3, help="timed steps of the config-5 leg at N > 1")
    ap.add_argument("--mixed-items", type=int, default=1 << 23, help="items per rank of the config-5 leg at N > 1")
    args = ap.parse_args()
    workload = args.workload or ("falcon512" if args.logn == 9 else "falcon1024")
    mixed = workload == "mixed"
    logn = 10 if workload == "falcon1024" else 9
    malformed = args.malformed if args.malformed >= 0 else (250 if workload == "falcon1024" and args.workload else 0)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    pmc_now = None
    if world == 1 and not args.no_pmc and workload == "falcon512" and (args.items or (1 << 20)) == 1 << 20 and not os.environ.get("ROCPROFILER_CONFIGURED") \
            and "rocprof" not in os.environ.get("LD_PRELOAD", ""):
        pmc_now = pmc_child_pass(1 << 20)   # before `import torch`: nothing in this process has touched the GPU yet
        if pmc_now:
            log(f"[rank 0] counter pass in {pmc_now['_seconds']:.0f}s:", {k: v for k, v in pmc_now.items() if k != "_seconds"})

    import torch
    import torch.distributed as dist

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

    import falcon_amd  # noqa: F401
    from falcon_amd import capi
    import mixed_batch
    import support

    threads, threads_src = host_cores()
    ctx = capi.Context(local_rank)
    if args.chunk:
        ctx.set_option(capi.OPT_CHUNK_ITEMS, args.chunk)
    if args.overlap >= 0:
        ctx.set_option(capi.OPT_OVERLAP, args.overlap)
    if args.grid_per_cu >= 0:
        ctx.set_option(capi.OPT_GRID_PER_CU, args.grid_per_cu)
    stream = torch.cuda.current_stream().cuda_stream

    def upload(*arrs):
        return [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in arrs]

    def build_mixed(c9, c10, n_items, seed):
        """2 * npairs items on the device: the two class corpora (torch tensors pk, msg, sig) tiled and interleaved
        pair by pair (tests/mixed_batch.py). Returns the tensors fa_verify_mixed_dev takes."""
        base = c9[0].shape[0]
        npairs = n_items // 2
        tile = max(1, npairs // base)
        npairs = base * tile if npairs >= base else npairs
        if npairs < base:
            c9 = [t[:npairs] for t in c9]
            c10 = [t[:npairs] for t in c10]
        bits = mixed_batch.pair_order_bits(npairs, seed)
        tags, pk_off, sig_off, msg_off, _ = mixed_batch.layout(bits, 32)
        bits_t = torch.from_numpy(bits).to(dev)
        d = {"count": 2 * npairs,
             "logn": torch.from_numpy(tags).to(dev),
             "pk": mixed_batch.pack_torch(bits_t, c9[0], c10[0], tile if npairs >= base else 1),
             "msg": mixed_batch.pack_torch(bits_t, c9[1], c10[1], tile if npairs >= base else 1),
             "sig": mixed_batch.pack_torch(bits_t, c9[2], c10[2], tile if npairs >= base else 1),
             "pk_off": torch.from_numpy(pk_off.view(np.int64)).to(dev),
             "sig_off": torch.from_numpy(sig_off.view(np.int64)).to(dev),
             "msg_off": torch.from_numpy(msg_off.view(np.int64)).to(dev)}
        d["accept"] = torch.zeros(d["count"], dtype=torch.uint8, device=dev)
        return d, bits

    def run_mixed(d):
        ctx.verify_mixed_dev(d["count"], d["logn"].data_ptr(), d["pk"].data_ptr(), d["pk_off"].data_ptr(), d["msg"].data_ptr(),
                             d["msg_off"].data_ptr(), d["sig"].data_ptr(), d["sig_off"].data_ptr(), d["accept"].data_ptr(),
                             stream=stream)

    # ---- corpus: this rank's shard of the seeded stream (counter-based, corpus/falcon_corpus.c) ----
    t0 = time.time()
    if mixed:
        n_items = args.items or (1 << 23)
        base = min(1 << 20, n_items // 2)
        c9h = support.corpus().generate(9, base, seed=0xFA1C0512, first=rank * base, mlen=32, threads=threads)
        c10h = support.corpus().generate(10, base, seed=0xFA1C1024, first=rank * base, mlen=32, malformed_permille=250, threads=threads)
        c9, c10 = upload(*c9h[:3]), upload(*c10h[:3])
        mx, mx_bits = build_mixed(c9, c10, n_items, 0x5EED + rank)
        n_items = mx["count"]
        del c9, c10
        torch.cuda.empty_cache()
        log(f"[rank {rank}] corpus: {n_items} mixed items ({base} distinct triples per class, tiled on the device) in {time.time() - t0:.1f}s")

        def step():
            run_mixed(mx)
        d_acc = mx["accept"]
    else:
        n_items = args.items or (1 << 20)
        pk, msg, sig, kinds = support.corpus().generate(logn, n_items, seed=0xFA1C0512 if logn == 9 else 0xFA1C1024, first=rank * n_items,
                                                        mlen=32, malformed_permille=malformed, threads=threads)
        log(f"[rank {rank}] corpus: {n_items} Falcon-{1 << logn} triples in {time.time() - t0:.1f}s ({threads} threads)")
        d_pk, d_msg, d_sig = upload(pk, msg, sig)
        d_acc = torch.zeros(n_items, dtype=torch.uint8, device=dev)

        def step():
            ctx.verify_many_dev(logn, n_items, d_pk.data_ptr(), d_msg.data_ptr(), None, 32, d_sig.data_ptr(),
                                d_acc.data_ptr(), stream=stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    # HIP events around every kernel, on the launch stream, taken from a pool that the warm-up pass below fills: the
    # timed loop creates no events, it records two per kernel launch
    ctx.set_option(capi.OPT_PROFILE, 1)
    step()
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
    names = {"falcon512": ("Falcon-512 verifies/sec", f"Falcon-512 batched verify, {n_items} distinct keys/signatures per GPU, 32-byte messages, "
                           "device-resident (BASELINE.json configs[2])"),
             "falcon1024": ("Falcon-1024 verifies/sec", f"Falcon-1024 batched verify, {n_items} distinct keys/signatures per GPU, 32-byte messages, "
                            "device-resident (BASELINE.json configs[3])"),
             "mixed": ("Falcon-512/1024 verifies/sec", f"mixed Falcon-512 + Falcon-1024 batch, {n_items} items per GPU, 50/50 by N, the Falcon-1024 half "
                       "with the 25 % malformed mix, tiled on the device from 2^20-item corpora, verified in place by fa_verify_mixed_dev "
                       f"(BASELINE.json configs[4]: {n_items * world} items over {world} GPU(s))")}
    result = {
        "metric": names[workload][0], "value": value, "unit": "verifies/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32 (mod q = 12289), u64 Keccak lanes",
        "data": "synthetic",
        "config": {"workload": names[workload][1] + (f", {malformed} permille malformed" if malformed and not mixed else ""),
                   "items_per_gpu": n_items, "parallelism": f"items sharded over {world} GPU(s), no collective"},
        "accepted_fraction": accepted / (n_items * world),
    }

    if rank == 0:
        # ---- roofline of the dominant kernel (VALU-bound integer work; HBM shown beside it) ----
        kern_ms = {"hash_kernel": prof["ms_hash"], "verify_kernel": prof["ms_verify"], "decode_kernel": prof["ms_decode"]}
        kern_n = {"hash_kernel": prof["launches_hash"], "verify_kernel": prof["launches_verify"],
                  "decode_kernel": prof["launches_decode"]}
        dom = max(kern_ms, key=kern_ms.get)
        avg_ms = kern_ms[dom] / max(kern_n[dom], 1)
        if mixed:   # two launches per kernel and step (one per class): ops and bytes of an average launch
            items_per_launch = n_items * args.steps / max(kern_n[dom], 1)
            ops_item = 0.5 * (ALG_OPS[9][dom] + ALG_OPS[10][dom])
            bytes_item = 0.5 * (BYTES[9] + BYTES[10])
        else:
            items_per_launch = n_items * args.steps / max(kern_n[dom], 1)
            ops_item, bytes_item = ALG_OPS[logn][dom], BYTES[logn]
        achieved = ops_item * items_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        roof = {
            "bound": "valu", "kernel": dom, "achieved": achieved, "peak": PEAK_VALU_TIOPS, "unit": "Tiop/s",
            "frac": achieved / PEAK_VALU_TIOPS, "traffic": None,
            "avg_launch_ms": avg_ms, "launches_timed": kern_n[dom], "items_per_launch": items_per_launch,
            "algorithmic_ops_per_item": ops_item,
            "note": "integer VALU work, no MFMA. achieved = minimum lane-operations of this kernel's work at the issued cost (bench.py "
                    "ALG_OPS: 4 per butterfly, 3 transforms; 180 per Keccak round) x items per launch / average launch time measured with "
                    "HIP events in this run; peak = 256 CU x 4 SIMD-32 x 2.4 GHz lane-ops/s, which only full-rate instructions reach "
                    "(every instruction of these streams takes a four-cycle issue slot: DESIGN.md section 6). Round 3 removed a sixth "
                    "of the verify kernel's work (4 instead of 5 instructions per butterfly), so this fraction is not comparable with "
                    "round 2's: frac_round2_work_model is",
        }
        if dom == "verify_kernel" and not mixed:
            r2 = ALG_OPS_R02_VERIFY[logn] * items_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
            roof["frac_round2_work_model"] = r2 / PEAK_VALU_TIOPS
        if workload == "falcon512":
            roof["reference_op_model"] = {"ops_per_item": REF_OPS_512[dom],
                                          "rate_Tiops": REF_OPS_512[dom] * items_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0,
                                          "note": "SURVEY.md 8(d) cost model of the REFERENCE's operations (11 per butterfly, 4 transforms); "
                                                  "not a roofline fraction: the kernel does the same work in fewer instructions"}
        ppath = os.path.join(ROOT, "profiles", "valu_peak_r01.json")   # tools/valu_peak.hip on an MI355X: single-instruction streams
        if os.path.exists(ppath):
            try:
                rates = json.load(open(ppath))["rates_Tops"]
                roof["peak_measured"] = {"source": "profiles/valu_peak_r01.json (tools/valu_peak.hip, SURVEY.md 8d: nominal and measured peak)",
                                         "v_add_u32_Tiops": rates["v_add_u32"], "v_mad_u32_u24_Tiops": rates["v_mad_u32_u24"],
                                         "v_bitop3_b32_Tiops": rates["v_bitop3_b32"], "v_alignbit_b32_Tiops": rates["v_alignbit_b32"],
                                         "frac_of_measured_full_rate_peak": achieved / rates["v_add_u32"]}
            except Exception as e:
                log("no measured peak:", e)
        tpath = os.path.join(ROOT, PMC_SOURCE)
        if workload == "falcon512" and n_items == 1 << 20 and os.path.exists(tpath):
            try:   # counters of the same launch (2^20 Falcon-512 items), collected in separate rocprofv3 --pmc passes
                counters = json.load(open(tpath))
                roof["traffic"] = counters["hbm_bytes_per_launch"][dom]
                roof["traffic_source"] = PMC_SOURCE + " (" + counters.get("_from", "?") + "): HBM bytes do not depend on the clock"
                issued = counters["valu_instructions_per_launch"][dom]
                clock = counters.get("clock_ghz", {}).get(dom)
                source = PMC_SOURCE + " (" + counters.get("_from", "?") + ")"
                if pmc_now and dom in pmc_now:   # this run's own counter pass (pmc_child_pass) takes precedence
                    issued, clock = pmc_now[dom]["valu_instructions"] or issued, pmc_now[dom]["clock_ghz"]
                    source = ("this run: rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU over tools/prof_run.py (3 launches of the same "
                              "2^20 items) in a child process on this GPU, before the timed region")
                roof["pmc"] = {"source": source,
                               "issued_valu_wave_instructions": issued,
                               "issued_lane_ops_frac_of_peak": issued * 64 / (avg_ms * 1e-3) / 1e12 / PEAK_VALU_TIOPS,
                               "clock_ghz_from_GRBM_GUI_ACTIVE": clock,
                               "cycles_per_valu_instruction_per_simd": (clock * 1e9 * avg_ms * 1e-3 * 1024 / issued) if clock else None,
                               "all_kernels": {k: v for k, v in pmc_now.items() if k != "_seconds"} if pmc_now else None}
            except Exception as e:   # the profile is optional
                log("no PMC summary:", e)
        result["roofline"] = roof
        result["kernels_ms_per_step"] = {k: v / args.steps for k, v in kern_ms.items()}
        per_gpu_rate = n_items * args.steps / elapsed
        hbm_achieved = bytes_item * items_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        result["roofline_hbm"] = {"bound": "hbm", "kernel": dom, "achieved": hbm_achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                  "frac": hbm_achieved / PEAK_HBM_GBS, "traffic": roof["traffic"],
                                  "algorithmic_bytes_per_item": bytes_item,
                                  "whole_path_GBps": per_gpu_rate * bytes_item / 1e9}
        ops_all = sum(ALG_OPS[logn].values()) if not mixed else 0.5 * (sum(ALG_OPS[9].values()) + sum(ALG_OPS[10].values()))
        result["whole_path_valu_frac"] = per_gpu_rate * ops_all / 1e12 / PEAK_VALU_TIOPS
        result["retried_items"] = prof["retried_items"]

    def timeit(fn, reps=5, warm=2):
        """Median and minimum time of `reps` calls of fn (each synchronised), after `warm` untimed calls."""
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        return ts[len(ts) // 2], ts[0]

    def leg_roofline(fn, items, ops_by_kernel, steps=3):
        """Per-kernel HIP-event times of `steps` calls of fn -> the dominant kernel's roofline entry (as for the headline)."""
        ctx.set_option(capi.OPT_PROFILE, 1)
        fn()
        ctx.get_profile()
        for _ in range(steps):
            fn()
        pr = ctx.get_profile()
        ctx.set_option(capi.OPT_PROFILE, 0)
        ms = {"hash_kernel": pr["ms_hash"], "verify_kernel": pr["ms_verify"], "decode_kernel": pr["ms_decode"]}
        nl = {"hash_kernel": pr["launches_hash"], "verify_kernel": pr["launches_verify"], "decode_kernel": pr["launches_decode"]}
        k = max(ms, key=ms.get)
        avg = ms[k] / max(nl[k], 1)
        per_launch = items * steps / max(nl[k], 1)
        ach = ops_by_kernel[k] * per_launch / (avg * 1e-3) / 1e12 if avg > 0 else 0.0
        return {"bound": "valu", "kernel": k, "achieved": ach, "peak": PEAK_VALU_TIOPS, "unit": "Tiop/s", "frac": ach / PEAK_VALU_TIOPS,
                "avg_launch_ms": avg, "items_per_launch": per_launch, "algorithmic_ops_per_item": ops_by_kernel[k],
                "kernels_ms_per_step": {a: b / steps for a, b in ms.items()}}

    if rank == 0 and world == 1 and not args.no_cpu and not mixed:
        # ---- CPU baseline: the oracle over a bounded sample of the same corpus ----
        ns = min(args.cpu_sample, n_items)
        orc = support.oracle()   # the checker: portable build
        fast = support.oracle_native()   # the timed build: this machine's -O3 -march=native (oracle/Makefile `native`), the reference's own flags
        timed, build = (fast, fast.built_with + ", built on this machine") if fast else (orc, "gcc -O3 -std=c11 (no native build here)")
        timed.verify_many(logn, pk[:4096], msg[:4096], sig[:4096], threads=threads)   # page in
        t0 = time.perf_counter()
        tflags, _ = timed.verify_many(logn, pk[:ns], msg[:ns], sig[:ns], threads=threads)
        dt = time.perf_counter() - t0
        t1 = time.perf_counter()
        n1 = min(16384, ns)
        timed.verify_many(logn, pk[:n1], msg[:n1], sig[:n1], threads=1)
        dt1 = time.perf_counter() - t1
        # where one verify's time goes on this host (the timed build's own entry points, 3,000 calls each through ctypes,
        # about a microsecond of call overhead included): the reference runs 3 ntt + 1 intt per verify (verification.hpp:47-57)
        import ctypes
        prof_us = {}
        u8p, u16p, i32p = (ctypes.POINTER(t) for t in (ctypes.c_uint8, ctypes.c_uint16, ctypes.c_int32))
        poly = np.arange(1 << logn, dtype=np.uint16)
        c_out = np.zeros(1 << logn, np.uint16)
        s2_out = np.zeros(1 << logn, np.int32)
        sig0, pk0, salt0, msg0 = (np.ascontiguousarray(a) for a in (sig[0], pk[0], sig[0][1:41], msg[0]))
        rng = np.random.default_rng(1)
        poly[:] = rng.integers(0, 12289, 1 << logn)
        pp, cp, sp2 = poly.ctypes.data_as(u16p), c_out.ctypes.data_as(u16p), s2_out.ctypes.data_as(i32p)   # made once: not part of the timing
        sigp, pkp, saltp, msgp = (a.ctypes.data_as(u8p) for a in (sig0, pk0, salt0, msg0))
        legs = {"ntt": lambda: timed.lib.fo_ntt(pp, logn),
                "intt": lambda: timed.lib.fo_intt(pp, logn),
                "hash_to_point": lambda: timed.lib.fo_hash_to_point(logn, saltp, 40, msgp, 32, cp),
                "decompress_sig": lambda: timed.lib.fo_decompress_sig(logn, sigp, sp2),
                "decode_pkey": lambda: timed.lib.fo_decode_pkey(logn, pkp, cp),
                "empty_call": lambda: timed.lib.fo_debug_size()}
        for name, fn in legs.items():
            fn()
            tl = time.perf_counter()
            for _ in range(3000):
                fn()
            prof_us[name] = (time.perf_counter() - tl) / 3000 * 1e6
        prof_us["note"] = ("exported entry points called one at a time (empty_call = the ctypes overhead in each figure); inside "
                           "fo_verify the same functions are inlined for a constant logn, so the parts need not add up to the whole")
        # every accept flag of the timed corpus against the CPU checker, untimed; the timed build's flags
        # against the checker's on the sample
        flags, _ = orc.verify_many(logn, pk, msg, sig, threads=threads)
        gpu_flags = d_acc.cpu().numpy()
        result["cpu_baseline"] = {
            "value": ns / dt, "unit": "verifies/s", "cores": threads, "kind": "port",
            "sample": f"first {ns} items of the same corpus, one verify per item, {threads} pthreads = {threads_src} "
                      f"(machine: {os.cpu_count()} CPUs; ~{dt * threads:.0f} core-seconds; oracle/falcon_oracle.c, {build}); "
                      f"single thread: {1e6 * dt1 / n1:.1f} us/verify (the reference publishes 19.8 us on an i7-1260P, README.md:150-155)",
            "host_cpus": os.cpu_count(), "single_thread_us": 1e6 * dt1 / n1, "build": build,
            "single_thread_profile_us": prof_us,
            "accept_flags_compared": int(flags.size),
            "accept_mismatches_vs_gpu": int((gpu_flags != (flags & 1)).sum()),
            "timed_build_mismatches_vs_checker": int((tflags != flags[:ns]).sum()),
        }
    elif rank == 0 and world == 1 and not args.no_cpu and mixed:
        # mixed workload: the oracle over a bounded sample of the first pairs (both classes), flags compared item by item
        ns = min(args.cpu_sample // 4, c9h[0].shape[0])
        orc = support.oracle()
        t0 = time.perf_counter()
        f9, _ = orc.verify_many(9, c9h[0][:ns], c9h[1][:ns], c9h[2][:ns], threads=threads)
        f10, _ = orc.verify_many(10, c10h[0][:ns], c10h[1][:ns], c10h[2][:ns], threads=threads)
        dt = time.perf_counter() - t0
        tags = mx["logn"][:2 * ns].cpu().numpy()
        src = np.repeat(np.arange(ns), 2)
        want = np.where(tags == 9, f9[src], f10[src]) & 1
        got = d_acc[:2 * ns].cpu().numpy()
        result["cpu_baseline"] = {
            "value": 2 * ns / dt, "unit": "verifies/s", "cores": threads, "kind": "port",
            "sample": f"the first {ns} pairs of the batch (one Falcon-512 and one Falcon-1024 verify each), {threads} pthreads = {threads_src} "
                      f"(machine: {os.cpu_count()} CPUs; oracle/falcon_oracle.c, gcc -O3 -std=c11)",
            "host_cpus": os.cpu_count(), "accept_flags_compared": int(want.size),
            "accept_mismatches_vs_gpu": int((got != want).sum()),
        }

    if rank == 0 and world == 1 and not args.no_extras and workload == "falcon512":
        extras = {}
        # Falcon-1024 with the malformed mix (BASELINE.json configs[3]) at its full 2^20 items
        ne = min(n_items, 1 << 20)
        pk2, msg2, sig2, kinds2 = support.corpus().generate(10, ne, seed=0xFA1C1024, mlen=32, malformed_permille=250, threads=threads)
        e_pk, e_msg, e_sig = upload(pk2, msg2, sig2)
        e_acc = torch.zeros(ne, dtype=torch.uint8, device=dev)

        def step1024():
            ctx.verify_many_dev(10, ne, e_pk.data_ptr(), e_msg.data_ptr(), None, 32, e_sig.data_ptr(),
                                e_acc.data_ptr(), stream=stream)
        med, best = timeit(step1024, reps=7, warm=3)
        nchk = 1 << 16
        t0 = time.perf_counter()
        oflags, _ = support.oracle().verify_many(10, pk2[:nchk], msg2[:nchk], sig2[:nchk], threads=threads)
        dt_cpu = time.perf_counter() - t0
        extras["falcon1024_malformed_mix"] = {
            "verifies_per_s": ne / med, "best": ne / best, "items": ne, "malformed_share": float((kinds2 != 0).mean()),
            "accepted_fraction": float(e_acc.float().mean().item()),
            "accept_mismatches_vs_oracle_first_65536": int((e_acc[:nchk].cpu().numpy() != (oflags & 1)).sum()),
            "roofline": leg_roofline(step1024, ne, ALG_OPS[10]),
            "cpu_baseline": {"value": nchk / dt_cpu, "unit": "verifies/s", "cores": threads, "kind": "port",
                             "sample": f"first {nchk} items of this corpus, {threads} pthreads"}}
        rate1024 = ne / med
        # mixed Falcon-512 + Falcon-1024 batch (BASELINE.json configs[4]): ONE GPU's share of the 64M-item batch, 2^23 items,
        # tiled on the device from the two 2^20-item corpora and verified in place by fa_verify_mixed_dev
        if n_items == 1 << 20:
            mx, _ = build_mixed([d_pk, d_msg, d_sig], [e_pk, e_msg, e_sig], 1 << 23, 0x5EED)
            med, best = timeit(lambda: run_mixed(mx), reps=5, warm=2)
            rate512 = value
            harm = 2.0 / (1.0 / rate512 + 1.0 / rate1024)
            got = mx["accept"][:1 << 15].cpu().numpy()
            tags = mx["logn"][:1 << 15].cpu().numpy()
            f9 = d_acc[:1 << 14].cpu().numpy()
            f10 = e_acc[:1 << 14].cpu().numpy()
            src = np.repeat(np.arange(1 << 14), 2)
            extras["mixed_512_1024_config5_share"] = {
                "verifies_per_s": mx["count"] / med, "best": mx["count"] / best, "items": mx["count"], "share_1024": 0.5,
                "harmonic_mean_of_single_class_rates": harm, "ratio_to_harmonic_mean": mx["count"] / med / harm,
                "accept_mismatches_vs_single_class_paths_first_32768": int((got != np.where(tags == 9, f9[src], f10[src])).sum()),
                "accepted_fraction": float(mx["accept"].float().mean().item()),
                "roofline": leg_roofline(lambda: run_mixed(mx), mx["count"],
                                         {k: 0.5 * (ALG_OPS[9][k] + ALG_OPS[10][k]) for k in ALG_OPS[9]}, steps=2)}
            del mx
            torch.cuda.empty_cache()
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
        # resident key set (fa_keyset_create / fa_verify_many_keyset): the keys stay on the GPU in prepared form, a call
        # sends signatures, messages and 4-byte key indices only -- 702 instead of 1,595 bytes per item over the host link
        ks = ctx.keyset_create(logn, pk=pk[:nh])
        kidx_pin = capi.PinnedArray((nh,), dtype=np.uint32)
        kidx_pin.array[...] = np.arange(nh, dtype=np.uint32)
        for name, (m_a, s_a, k_a, a_out) in (("pageable", (msg[:nh], sig[:nh], np.arange(nh, dtype=np.uint32), None)),
                                            ("pinned", (pin[1].array, pin[2].array, kidx_pin.array, pin[3].array))):
            ctx.verify_many_keyset(ks, m_a, s_a, key_idx=k_a, accept_out=a_out)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                k_acc = ctx.verify_many_keyset(ks, m_a, s_a, key_idx=k_a, accept_out=a_out)
                ts.append(time.perf_counter() - t0)
            dt = sorted(ts)[1]
            extras["host_buffers_resident_keys" + ("" if name == "pinned" else "_pageable")] = {
                "verifies_per_s": nh / dt, "items": nh, "caller_memory": name, "input_bytes_per_item": sig.shape[1] + 32 + 4,
                "input_GBps": nh * (sig.shape[1] + 32 + 4) / dt / 1e9, "keys_resident": nh,
                "accept_mismatches_vs_device_path": int((k_acc != d_acc[:nh].cpu().numpy()).sum())}
        ks.close()
        kidx_pin.close()
        for b in pin:
            b.close()
        # decoded-key fast path: keys prepared once (NTT(h) resident), same 2^20 Falcon-512 items, against the
        # packed-key path in interleaved rounds of one process (the GPU's clock drifts with load: back-to-back legs are not
        # comparable, alternating rounds are)
        d_prep = torch.empty((n_items, 512), dtype=torch.int16, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.prepare_keys_dev(9, n_items, d_pk.data_ptr(), None, d_prep.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        t_prep = time.perf_counter() - t0
        p_acc = torch.zeros(n_items, dtype=torch.uint8, device=dev)

        def step_prepared():
            ctx.verify_many_prepared_dev(9, n_items, d_prep.data_ptr(), None, d_msg.data_ptr(), None, 32,
                                         d_sig.data_ptr(), p_acc.data_ptr(), stream=stream)
        for _ in range(3):
            step()
            step_prepared()
        torch.cuda.synchronize()
        tp, tk = [], []
        for _ in range(8):
            for fn, acc_t in ((step, tk), (step_prepared, tp)):
                t0 = time.perf_counter()
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                acc_t.append((time.perf_counter() - t0) / 3)
        tp.sort()
        tk.sort()
        extras["falcon512_prepared_keys"] = {"verifies_per_s": n_items / tp[len(tp) // 2], "items": n_items,
                                             "packed_path_same_rounds_verifies_per_s": n_items / tk[len(tk) // 2],
                                             "gain_over_packed_path": tk[len(tk) // 2] / tp[len(tp) // 2],
                                             "rounds": "8 interleaved rounds of 3 steps each, medians",
                                             "prepare_keys_per_s": n_items / t_prep,
                                             "accept_mismatches_vs_packed_path": int((p_acc != d_acc).sum().item())}
        del d_prep, p_acc
        # batched negacyclic polymul, n = 512 (BASELINE.json configs[1]) at its full 2^24 polynomials per operand
        # (3 x 16 GiB resident), with its own roofline entries
        npoly = 1 << 24
        a = torch.empty((npoly, 512), dtype=torch.int16, device=dev)
        b = torch.empty((npoly, 512), dtype=torch.int16, device=dev)
        for c0 in range(0, npoly, 1 << 21):
            a[c0:c0 + (1 << 21)] = torch.randint(0, 12289, (1 << 21, 512), dtype=torch.int16, device=dev)
            b[c0:c0 + (1 << 21)] = torch.randint(0, 12289, (1 << 21, 512), dtype=torch.int16, device=dev)
        c = torch.empty_like(a)
        med, best = timeit(lambda: ctx.polymul_many_dev(9, npoly, a.data_ptr(), b.data_ptr(), c.data_ptr(), stream=stream), reps=5, warm=2)
        extras["polymul_n512"] = {"polys_per_s": npoly / med, "best": npoly / best, "polys": npoly,
                                  "roofline": {"bound": "valu", "kernel": "polymul_kernel", "achieved": npoly / med * POLYMUL_OPS_512 / 1e12,
                                               "peak": PEAK_VALU_TIOPS, "unit": "Tiop/s", "frac": npoly / med * POLYMUL_OPS_512 / 1e12 / PEAK_VALU_TIOPS,
                                               "algorithmic_ops_per_item": POLYMUL_OPS_512},
                                  "roofline_hbm": {"bound": "hbm", "achieved": npoly / med * 3072 / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                                   "frac": npoly / med * 3072 / 1e9 / PEAK_HBM_GBS, "algorithmic_bytes_per_item": 3072}}
        del a, b, c
        # one signature per call (falcon::verify<N>, falcon.hpp:230-249, the reference's own per-call benchmark shape,
        # README.md:150-167: 20-44 us on a CPU core): wall time of a device-resident call and of a host-buffer call
        lat = {}
        for ln, (lpk, lmsg, lsig) in ((9, (pk, msg, sig)), (10, (pk2, msg2, sig2))):
            one = [torch.from_numpy(np.ascontiguousarray(x[:1])).to(dev) for x in (lpk, lmsg, lsig)]
            one_acc = torch.zeros(1, dtype=torch.uint8, device=dev)

            def dev_one():
                ctx.verify_many_dev(ln, 1, one[0].data_ptr(), one[1].data_ptr(), None, 32, one[2].data_ptr(), one_acc.data_ptr(), stream=stream)
            for _ in range(20):
                dev_one()
            t0 = time.perf_counter()
            for _ in range(200):
                dev_one()  # synchronises its stream itself (no FA_OPT_ASYNC): nothing is left in flight
            med_d = (time.perf_counter() - t0) / 200
            t_h = []
            for _ in range(220):
                t0 = time.perf_counter()
                ctx.verify_many(ln, lpk[:1], lmsg[:1], lsig[:1])
                t_h.append(time.perf_counter() - t0)
            t_h = sorted(t_h[20:])
            lat[f"falcon{1 << ln}"] = {"device_resident_call_us": med_d * 1e6, "host_buffer_call_us": t_h[len(t_h) // 2] * 1e6,
                                           "note": "mean of 200 device-resident calls, median of 200 host-buffer calls"}
        extras["one_item_per_call_latency"] = lat
        result["extras"] = extras

    if world > 1 and not args.no_extras and workload == "falcon512" and n_items >= 1 << 15:
        # BASELINE config 5 on the driver's plain multi-GPU command: after the headline region every rank builds ITS share
        # of the mixed batch -- 2^23 items, 50/50 Falcon-512 / Falcon-1024 (the latter with the 25 % malformed mix), tiled
        # on the device from this rank's two 2^20-item corpora -- and all ranks time the same number of steps between two
        # barriers. Items are independent: no collective on the data path, the only way this fails to scale is a
        # straggler, so the per-rank rates are reported too.
        t0 = time.time()
        pk2, msg2, sig2, _ = support.corpus().generate(10, n_items, seed=0xFA1C1024, first=rank * n_items, mlen=32, malformed_permille=250,
                                                        threads=threads)
        e_dev = upload(pk2, msg2, sig2)
        mx, _ = build_mixed([d_pk, d_msg, d_sig], e_dev, args.mixed_items, 0x5EED + rank)
        del e_dev
        torch.cuda.empty_cache()
        t_build = time.time() - t0
        run_mixed(mx)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.mixed_steps):
            run_mixed(mx)
        torch.cuda.synchronize()
        t_own = time.perf_counter() - t0
        dist.barrier()
        t_all = time.perf_counter() - t0
        # this rank's accept bytes against what each item's class dictates: the single-class result of the headline corpus
        # for the Falcon-512 items (all valid), a sample of the Falcon-1024 items against the oracle
        tags = mx["logn"][:1 << 15].cpu().numpy()
        got = mx["accept"][:1 << 15].cpu().numpy()
        src = np.repeat(np.arange(1 << 14), 2)
        o10, _ = support.oracle().verify_many(10, pk2[:1 << 14], msg2[:1 << 14], sig2[:1 << 14], threads=threads)
        want = np.where(tags == 9, d_acc[:1 << 14].cpu().numpy()[src], (o10 & 1)[src])
        vals = torch.tensor([t_all, t_own, t_build, float((got != want).sum()), float(mx["count"])], dtype=torch.float64, device=red_dev)
        gathered = [torch.zeros_like(vals) for _ in range(world)]
        dist.all_gather(gathered, vals)
        if rank == 0:
            g = np.stack([x.cpu().numpy() for x in gathered])
            items_total = int(g[:, 4].sum())
            per_rank = g[:, 4] * args.mixed_steps / g[:, 1]
            result.setdefault("extras", {})["mixed_512_1024_config5"] = {
                "verifies_per_s": items_total * args.mixed_steps / float(g[:, 0].max()), "items_total": items_total, "items_per_rank": int(g[0, 4]),
                "steps": args.mixed_steps, "share_1024": 0.5, "per_rank_verifies_per_s_min": float(per_rank.min()),
                "per_rank_verifies_per_s_max": float(per_rank.max()), "per_rank_corpus_build_s_max": float(g[:, 2].max()),
                "accept_mismatches_first_32768_per_rank_sum": int(g[:, 3].sum()),
                "note": "BASELINE.json configs[4]: mixed Falcon-512 + Falcon-1024 batch sharded by item over the ranks, no collective on the "
                        "data path; timed between two barriers, aggregate = all items / the slowest rank's time"}
        del mx
        torch.cuda.empty_cache()

    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
