"""GPU: the N > 1 path with the real verifier. Two gloo ranks on one GPU (--same-gpu style: this pool has one-GPU boxes)
shard a batch with falcon_amd.shard.verify_many_sharded and Context.verify_many, and rank 0 gathers the accept bytes;
and a two-rank rehearsal of bench.py itself. No scaling curve is measured here -- only that the path is correct."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

RANK_MAIN = r"""
import os, sys
sys.path.insert(0, {here!r}); sys.path.insert(0, {root!r})
import numpy as np
import torch, torch.distributed as dist
torch.cuda.init()
import support
from falcon_amd import capi, shard
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
ctx = capi.Context(0)            # every rank on GPU 0
count = int(sys.argv[1])
pk, msg, sig, kinds = support.corpus().generate(9, count, seed=321, malformed_permille=400, threads=4)
acc = shard.verify_many_sharded(ctx.verify_many, 9, pk, msg, sig)
if rank == 0:
    np.save(sys.argv[2], acc)
else:
    assert acc is None
dist.barrier()
dist.destroy_process_group()
ctx.close()
"""


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(script_args, nproc=2, timeout=300):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + script_args
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


@pytest.mark.parametrize("count", [4001, 3])
def test_two_ranks_one_gpu_sharded_verify(tmp_path, oracle, corpus, count):
    script = tmp_path / "rank_main.py"
    script.write_text(RANK_MAIN.format(here=HERE, root=ROOT))
    out_path = str(tmp_path / "acc.npy")
    r = _launch([str(script), str(count), out_path])
    assert r.returncode == 0, r.stderr[-3000:]
    got = np.load(out_path)
    pk, msg, sig, _ = corpus.generate(9, count, seed=321, malformed_permille=400, threads=4)
    want = oracle.verify_many(9, pk, msg, sig, threads=8)[0] & 1
    np.testing.assert_array_equal(got, want)


def test_bench_two_rank_rehearsal():
    """bench.py as the driver launches it for N = 2, with both ranks on GPU 0 and gloo for the barrier."""
    r = _launch(["bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--items", "65536", "--backend", "gloo", "--same-gpu",
                 "--mixed-items", "262144", "--mixed-steps", "2"])
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["accepted_fraction"] == 1.0
    assert d["value"] > 0 and d["roofline"]["frac"] <= 1.0
    # the plain multi-GPU command also measures BASELINE config 5's shape (mixed 512 + 1024) on every rank
    m = d["extras"]["mixed_512_1024_config5"]
    assert m["items_total"] == 2 * 262144 and m["items_per_rank"] == 262144 and m["accept_mismatches_first_32768_per_rank_sum"] == 0
    assert 0 < m["per_rank_verifies_per_s_min"] <= m["per_rank_verifies_per_s_max"] and m["verifies_per_s"] > 0
