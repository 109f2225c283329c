#!/bin/bash
# usage: tools/ab_lib.sh "<bench args>" <lib.so> [<lib.so> ...]   (A/B of alternative builds of the library, same bench args)
args="$1"; shift
mkdir -p gpurun_out
for lib in "$@"; do
  FALCON_AMD_LIB="$PWD/$lib" timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-extras --no-cpu --no-pmc $args 2>>gpurun_out/bench_err.log \
    | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], '|', round(d['value']/1e6,1), 'M/s', round(d['ms_per_step'],3), 'ms', {k: round(v,3) for k,v in d['kernels_ms_per_step'].items()}, 'acc', d['accepted_fraction'])" "$lib" || exit 1
done
