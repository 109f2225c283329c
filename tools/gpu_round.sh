#!/bin/bash
# GPU-box routine: parity tests, default bench line, profiles. usage: tools/gpu_round.sh <tag>
set -uo pipefail
tag="${1:-r01}"
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_gpu_$tag.log 2>&1 \
  || { echo "GPU TESTS FAILED"; tail -30 gpurun_out/pytest_gpu_$tag.log; exit 1; }
tail -3 gpurun_out/pytest_gpu_$tag.log
timeout -k 10 600 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || { echo "BENCH FAILED"; tail -20 gpurun_out/bench_$tag.err; exit 1; }
cat gpurun_out/bench_$tag.json
bash tools/profile_all.sh "$tag"
