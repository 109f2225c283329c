#!/bin/bash
# usage: tools/ab.sh "<bench args A>" "<bench args B>" ...   (prints value and per-kernel ms for each)
for a in "$@"; do
  python bench.py --steps 5 --warmup 1 --no-extras --no-cpu --no-pmc $a 2>>gpurun_out/bench_err.log \
    | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], '|', round(d['value']/1e6,1), 'M/s', round(d['ms_per_step'],3), 'ms', {k: round(v,3) for k,v in d['kernels_ms_per_step'].items()})" "$a"
done
