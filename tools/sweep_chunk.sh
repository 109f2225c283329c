#!/bin/bash
# usage: tools/sweep_chunk.sh chunk...   (prints value and per-kernel ms for each chunk size)
for c in "$@"; do
  python bench.py --steps 5 --warmup 1 --no-extras --no-cpu --chunk "$c" 2>>gpurun_out/bench_err.log \
    | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['value']/1e6,1), {k: round(v,3) for k,v in d['kernels_ms_per_step'].items()})" "$c"
done
