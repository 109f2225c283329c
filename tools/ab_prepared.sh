#!/bin/bash
# usage: tools/ab_prepared.sh <lib.so> ...   (prepared-key path and polymul of alternative builds: bench extras)
for lib in "$@"; do
  FALCON_AMD_LIB="$PWD/$lib" timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu --no-pmc 2>>gpurun_out/bench_err.log \
    | python -c "import json,sys; d=json.loads(sys.stdin.read()); e=d['extras']; print(sys.argv[1], '| packed', round(d['value']/1e6,1), 'prepared', round(e['falcon512_prepared_keys']['verifies_per_s']/1e6,1), 'M/s, mismatches', e['falcon512_prepared_keys']['accept_mismatches_vs_packed_path'], 'polymul', round(e['polymul_n512']['polys_per_s']/1e6,1), 'M/s, F1024 mix', round(e['falcon1024_malformed_mix']['verifies_per_s']/1e6,1))" "$lib" || exit 1
done
