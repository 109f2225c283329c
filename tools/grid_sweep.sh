#!/bin/bash
# A/B: workgroups per CU of the verify kernel
set -e
for g in 0 5 10 20 30 40 80 120; do
  timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu --grid-per-cu $g 2>>gpurun_out/bench_err.log \
    | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('grid_per_cu', sys.argv[1], '|', round(d['value']/1e6,1), 'M/s', round(d['ms_per_step'],3), 'ms', {k: round(v,3) for k,v in d['kernels_ms_per_step'].items()}, 'acc', d['accepted_fraction'])" $g
done
for g in 0 3 6 12 24 48; do
  timeout -k 10 200 python bench.py --logn 10 --steps 5 --warmup 2 --no-extras --no-cpu --grid-per-cu $g 2>>gpurun_out/bench_err.log \
    | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('F1024 grid_per_cu', sys.argv[1], '|', round(d['value']/1e6,1), 'M/s', round(d['ms_per_step'],3), 'ms', {k: round(v,3) for k,v in d['kernels_ms_per_step'].items()}, 'acc', d['accepted_fraction'])" $g
done
